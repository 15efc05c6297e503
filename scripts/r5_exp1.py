"""Round-5 experiments on the 256x320 kernel's developer instantiation (GemmArgs::dev through knob 40; run on the GPU box):
  (1) phase stagger of the blocks (bits 16-18 = phase groups, bits 20-27 = delay per group in ~us) on the short-K row GEMMs whose
      launches alternate a read-only K-loop phase with a store-bound epilogue phase on ALL CUs at once;
  (2) conv K order (channel chunk, tap) instead of (tap, channel chunk) (bit 0x800): the nine taps of a 64-channel chunk in
      consecutive K tiles, so that a tile's re-reads of neighbouring pixels hit the L2."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops
from flash_diffusion_amd._lib import lib
from scripts.rowbench import bench

BF = torch.bfloat16
TILE = (256 << 16) | 320
L = lib()


def stagger(reps):
    cases = [(131072, 2560, 320, False, True), (131072, 960, 320, False, False), (131072, 320, 320, True, False),
             (32768, 5120, 640, False, True), (131072, 320, 1280, True, False), (65536, 2560, 320, False, True)]
    variants = [("base", 0)] + [(f"p{ph}d{dl}", (ph << 16) | (dl << 20)) for ph in (2, 3) for dl in (3, 6, 10, 16)]
    for (M, N, K, res, geglu) in cases:
        Nout = N // 2 if geglu else N
        sets = []
        for _ in range(3):
            A = torch.randn(M, K, device="cuda").to(BF)
            W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
            out = torch.empty(M, Nout, dtype=BF, device="cuda")
            R = torch.randn(M, Nout, device="cuda").to(BF) if res else None
            sets.append((A, W, out, R))
        bias = torch.randn(N, device="cuda")
        fns = [(lambda A=A, W=W, out=out, R=R: ops.gemm(A, W, bias=bias, residual=R, out=out, force_tile=TILE,
                                                         act=ops.ACT_GEGLU if geglu else ops.ACT_NONE)) for (A, W, out, R) in sets]
        L.fdmi_tune_set(40, 0)
        fns[0]()
        ref = sets[0][2].clone()
        line = f"M={M:6d} N={N:5d} K={K:5d} res={int(res)} geglu={int(geglu)} |"
        for name, d in variants:
            L.fdmi_tune_set(40, d)
            us = bench(fns, reps)
            sets[0][2].zero_()
            fns[0]()
            torch.cuda.synchronize()
            ok = torch.equal(sets[0][2], ref)
            line += f" {name}: {us:6.1f}{'' if ok else ' MISMATCH'} |"
        L.fdmi_tune_set(40, 0)
        print(line, flush=True)


def conv_order(reps, B=32):
    for (hw, ci, co) in [(64, 320, 320), (64, 640, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280), (64, 960, 320)]:
        M = B * hw * hw
        sets = []
        for _ in range(2):
            x = torch.randn(B, hw, hw, ci, device="cuda").to(BF)
            w = (torch.randn(co, 9 * ci, device="cuda") * (9 * ci) ** -0.5).to(BF)
            out = torch.empty(M, co, dtype=BF, device="cuda")
            sets.append((x, w, out))
        bias = torch.zeros(co, device="cuda")
        conv = dict(Hin=hw, Win=hw, Cin=ci, Hout=hw, Wout=hw, KH=3, KW=3, stride=1, pad=1)
        fl = 2.0 * M * co * 9 * ci
        fns = [(lambda x=x, w=w, out=out: ops.gemm(x, w, M=M, bias=bias, out=out, conv=conv, force_tile=TILE)) for (x, w, out) in sets]
        line = f"conv3x3 B={B} {hw}x{hw} {ci}->{co} |"
        ref = None
        # 16: every tile gathers the first 256 output pixels' windows (A served by the L2: WRONG results) -- the conv kernel's own
        # "A from L2" ablation (is its 5.2x fabric traffic free?); 32: no epilogue; 48: both
        for name, d in (("(tap,c)", 0), ("(c,tap)", 0x800), ("A from L2", 16), ("no epilogue", 32), ("both", 48), ("(tap,c) again", 0)):
            L.fdmi_tune_set(40, d)
            us = bench(fns, reps)
            fns[0]()
            torch.cuda.synchronize()
            o = sets[0][2].float()
            if ref is None:
                ref = o.clone()
            err = float((o - ref).norm() / ref.norm())
            line += f" {name}: {us:7.1f} us {fl / us / 1e6:6.0f} TF" + (f" rel {err:.1e}" if d in (0, 0x800) else "") + " |"
        L.fdmi_tune_set(40, 0)
        print(line, flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    if what in ("stagger", "all"):
        stagger(reps)
    if what in ("conv", "all"):
        conv_order(reps)
