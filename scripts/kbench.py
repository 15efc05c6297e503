"""Kernel micro-benchmarks on the SD1.5 B=16 shapes (dev tool; run on the GPU box).
usage: python scripts/kbench.py [gemm|attn|norm|all] [reps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops

BF = torch.bfloat16


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def gemm_cases():
    cs = []
    # (name, kind, B, H, W, Cin, Cout) convs 3x3 ; linear (M,N,K)
    for (hw, ci, co) in [(64, 320, 320), (64, 640, 320), (32, 640, 640), (32, 320, 640), (16, 1280, 1280), (8, 1280, 1280),
                         (8, 2560, 1280), (16, 2560, 1280), (32, 1920, 640)]:
        cs.append((f"conv3x3 {hw}x{hw} {ci}->{co}", "conv", hw, ci, co))
    for (m, n, k) in [(65536, 2560, 320), (65536, 320, 1280), (65536, 320, 320), (16384, 5120, 640), (16384, 640, 640),
                      (4096, 10240, 1280), (4096, 1280, 1280), (4096, 1280, 5120), (1232, 320, 768), (65536, 128, 320)]:
        cs.append((f"linear {m}x{n}x{k}", "lin", m, n, k))
    return cs


def run_gemm(reps, **kw):
    B = 16
    for c in gemm_cases():
        if c[1] == "conv":
            _, _, hw, ci, co = c
            x = torch.randn(B, hw, hw, ci, device="cuda").to(BF)
            w = (torch.randn(co, 9 * ci, device="cuda") * (9 * ci) ** -0.5).to(BF)
            bias = torch.zeros(co, device="cuda")
            M = B * hw * hw
            out = torch.empty(M, co, dtype=BF, device="cuda")
            ws = torch.empty(16 * M * co, dtype=torch.float32, device="cuda")
            conv = dict(Hin=hw, Win=hw, Cin=ci, Hout=hw, Wout=hw, KH=3, KW=3, stride=1, pad=1)
            fl = 2.0 * M * co * 9 * ci
            us = timeit(lambda: ops.gemm(x, w, M=M, bias=bias, out=out, conv=conv, ws=ws, splitk=0, **kw), reps)
        else:
            _, _, m, n, k = c
            x = torch.randn(m, k, device="cuda").to(BF)
            w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(BF)
            out = torch.empty(m, n, dtype=BF, device="cuda")
            ws = torch.empty(16 * m * n, dtype=torch.float32, device="cuda")
            fl = 2.0 * m * n * k
            us = timeit(lambda: ops.gemm(x, w, out=out, ws=ws, splitk=0, **kw), reps)
        print(f"{c[0]:34s} {us:9.1f} us  {fl / us / 1e6:8.1f} TF/s", flush=True)


def run_attn(reps):
    from flash_diffusion_amd._lib import lib
    for qf in (0, 1):
        lib().fdmi_tune_set(0, qf)
        print("fwd QF knob", qf)
        _run_attn(reps)
    lib().fdmi_tune_set(0, 0)


def _run_attn(reps):
    B = 16
    for (S, Skv, H, d) in [(4096, 4096, 8, 40), (4096, 77, 8, 40), (1024, 1024, 8, 80), (256, 256, 8, 160), (64, 64, 8, 160)]:
        q = torch.randn(B, S, H * d, device="cuda").to(BF)
        k = torch.randn(B, Skv, H * d, device="cuda").to(BF)
        v = torch.randn(B, Skv, H * d, device="cuda").to(BF)
        us = timeit(lambda: ops.attn_fwd(q, k, v, H, d ** -0.5), reps)
        fl = 4.0 * B * H * S * Skv * d
        print(f"attn fwd S={S} Skv={Skv} d={d}: {us:9.1f} us {fl / us / 1e6:8.1f} TF/s", flush=True)
        o, lse = ops.attn_fwd(q, k, v, H, d ** -0.5, need_lse=True)
        do = torch.randn_like(o)
        us = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, H, d ** -0.5), max(reps // 4, 2))
        print(f"attn bwd S={S} Skv={Skv} d={d}: {us:9.1f} us {2.5 * fl / us / 1e6:8.1f} TF/s (2.5x fwd flops)", flush=True)


def run_norm(reps):
    B = 16
    for (hw, c) in [(64, 320), (64, 640), (32, 640), (16, 1280), (8, 2560)]:
        x = torch.randn(B, hw * hw, c, device="cuda").to(BF)
        g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        us = timeit(lambda: ops.groupnorm_fwd(x, g, b, 32, 1e-5, 1), reps)
        by = x.numel() * 2 * 3
        print(f"groupnorm+silu {hw}x{hw}x{c}: {us:8.1f} us  {by / us / 1e6:6.2f} TB/s (3 passes)", flush=True)
        us = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5), reps)
        print(f"layernorm      {hw}x{hw}x{c}: {us:8.1f} us  {x.numel() * 4 / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    if what in ("gemm", "all"):
        run_gemm(reps)
    if what in ("attn", "all"):
        run_attn(reps)
    if what in ("norm", "all"):
        run_norm(reps)
