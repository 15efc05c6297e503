"""Short-K row GEMMs of the C2 step under each kernel geometry (dev tool; run on the GPU box):
  python scripts/rowbench.py [reps]
per problem: the 256x320 one-block-per-CU kernel, the 128x320 two-blocks-per-CU kernel (round 6, gemm5.hip; its output must equal the 256x320 kernel's bit for bit), the 256x160 ring kernel and
the 128x128 tile -- us per launch, algorithmic TB/s (every operand once) and TFLOP/s.  Operands rotate over several buffer sets
so that a launch does not find its inputs in the 256 MB Infinity Cache."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops

BF = torch.bfloat16
TILES = {"g4 256x320": (256 << 16) | 320, "g5 128x320": (128 << 16) | 320, "g3 256x160": (256 << 16) | 160, "t 128x128": (128 << 16) | 128}


def bench(fns, reps):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


# gemm4 row-kernel variants (GemmArgs::dev through developer knob 40): L2 prefetch distance 1..3, residual touches (4), burst issue
# (8); 16 / 32 / 48 are timing ablations (A re-read from its first tile / no epilogue / both: WRONG results, never a product path)
DEV = [0, 256, 64, 32]          # 256: lean epilogue, every pair single (16 rows x 64 B per instruction); 64: the general epilogue; 32: the K loop alone (wrong results)


def dev_sweep(reps):
    """round 4: the 256x320 kernel of every row case under each GemmArgs::dev variant; the non-ablation variants must reproduce
    the default's output bit for bit"""
    from flash_diffusion_amd._lib import lib
    L = lib()
    cases = [(131072, 320, 320, True), (131072, 320, 320, False), (131072, 960, 320, False), (65536, 320, 320, True),
             (131072, 320, 1280, True), (32768, 640, 640, True), (32768, 1920, 640, False), (32768, 640, 2560, True)]
    tile = TILES["g4 256x320"]
    for (M, N, K, res) in cases:
        nset = min(6, max(2, int(600e6 // (2 * M * (K + N * (2 if res else 1)))) + 1))
        sets = []
        for s_ in range(nset):
            A = torch.randn(M, K, device="cuda").to(BF)
            W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
            out = torch.empty(M, N, dtype=BF, device="cuda")
            R = torch.randn(M, N, device="cuda").to(BF) if res else None
            sets.append((A, W, out, R))
        bias = torch.randn(N, device="cuda")
        by = 2.0 * M * (K + N * (2 if res else 1)) + 2.0 * N * K
        fns = [(lambda A=A, W=W, out=out, R=R: ops.gemm(A, W, bias=bias, residual=R, out=out, force_tile=tile)) for (A, W, out, R) in sets]
        L.fdmi_tune_set(40, 64)
        fns[0]()
        ref = sets[0][2].clone()
        line = f"M={M:6d} N={N:5d} K={K:5d} res={int(res)} |"
        for d in DEV:
            L.fdmi_tune_set(40, d)
            us = bench(fns, reps)
            tag = ""
            if d in (0, 256):
                sets[0][2].zero_()
                fns[0]()
                torch.cuda.synchronize()
                if not torch.equal(sets[0][2], ref):
                    tag = f" MISMATCH({float((sets[0][2].float() - ref.float()).abs().max()):.3g})"
            line += f" dev{d}: {us:6.1f} us {by / us / 1e6:4.2f} TB/s{tag} |"
        L.fdmi_tune_set(40, 0)
        print(line, flush=True)
    for (M, N, K) in [(131072, 2560, 320), (65536, 2560, 320), (32768, 5120, 640)]:   # GEGLU: K loop vs epilogue (ablations only)
        sets = []
        for s_ in range(3):
            A = torch.randn(M, K, device="cuda").to(BF)
            W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
            sets.append((A, W, torch.empty(M, N // 2, dtype=BF, device="cuda")))
        bias = torch.randn(N, device="cuda")
        fns = [(lambda A=A, W=W, out=out: ops.gemm(A, W, bias=bias, out=out, force_tile=tile, act=ops.ACT_GEGLU)) for (A, W, out) in sets]
        line = f"M={M:6d} N={N:5d} K={K:5d} GEGLU |"
        L.fdmi_tune_set(40, 64)
        fns[0]()
        ref = sets[0][2].clone()
        for d in (0, 64, 32):
            L.fdmi_tune_set(40, d)
            us = bench(fns, reps)
            tag = ""
            if d == 0:
                sets[0][2].zero_()
                fns[0]()
                torch.cuda.synchronize()
                if not torch.equal(sets[0][2], ref):
                    tag = f" MISMATCH({float((sets[0][2].float() - ref.float()).abs().max()):.3g})"
            line += f" dev{d}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF{tag} |"
        L.fdmi_tune_set(40, 0)
        print(line, flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "dev":
        return dev_sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 30)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    cases = [  # (M, N, K, residual, geglu)
        (131072, 320, 320, True, False), (131072, 320, 320, False, False), (131072, 960, 320, False, False),
        (65536, 320, 320, True, False), (65536, 320, 128, True, False), (131072, 2560, 320, False, True),
        (32768, 640, 640, True, False), (32768, 1920, 640, False, False), (32768, 5120, 640, False, True),
        (131072, 320, 1280, True, False), (32768, 640, 2560, True, False), (8192, 1280, 1280, True, False),
        (65536, 2560, 320, False, True), (32768, 640, 640, False, False), (65536, 320, 320, False, False), (8192, 10240, 1280, False, True),
        (8192, 3840, 1280, False, False), (16384, 640, 640, True, False)]
    for (M, N, K, res, geglu) in cases:
        Nout = N // 2 if geglu else N
        nset = max(2, int(600e6 // (2 * M * (K + Nout * (2 if res else 1)))) + 1)
        sets = []
        for s in range(min(nset, 6)):
            A = torch.randn(M, K, device="cuda").to(BF)
            W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
            out = torch.empty(M, Nout, dtype=BF, device="cuda")
            R = torch.randn(M, Nout, device="cuda").to(BF) if res else None
            sets.append((A, W, out, R))
        bias = torch.zeros(N, device="cuda")
        by = 2.0 * M * (K + Nout * (2 if res else 1)) + 2.0 * N * K
        fl = 2.0 * M * N * K
        line = f"M={M:6d} N={N:5d} K={K:5d} res={int(res)} geglu={int(geglu)} |"
        for name, tile in TILES.items():
            if geglu and name in ("g3 256x160", "t 128x128"):
                continue
            try:
                fns = [(lambda A=A, W=W, out=out, R=R: ops.gemm(A, W, bias=bias, residual=R, out=out, force_tile=tile,
                                                                 act=ops.ACT_GEGLU if geglu else ops.ACT_NONE)) for (A, W, out, R) in sets]
                us = bench(fns, reps)
                tag = ""
                sets[0][2].zero_()
                fns[0]()
                torch.cuda.synchronize()
                if name == "g4 256x320":
                    ref = sets[0][2].clone()
                elif name == "g5 128x320" and not torch.equal(sets[0][2], ref):
                    tag = f" MISMATCH({float((sets[0][2].float() - ref.float()).abs().max()):.3g})"
                line += f" {name}: {us:7.1f} us {by / us / 1e6:5.2f} TB/s {fl / us / 1e6:6.0f} TF{tag} |"
            except RuntimeError as e:
                line += f" {name}: n/a |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
