"""Short-K row GEMMs of the C2 step under each kernel geometry (dev tool; run on the GPU box):
  python scripts/rowbench.py [reps]
per problem: the 256x320 one-block-per-CU kernel, the 128x160 two-blocks-per-CU geometry (round 3), the 256x160 ring kernel and
the 128x128 tile -- us per launch, algorithmic TB/s (every operand once) and TFLOP/s.  Operands rotate over several buffer sets
so that a launch does not find its inputs in the 256 MB Infinity Cache."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_diffusion_amd import ops

BF = torch.bfloat16
TILES = {"g4 256x320": (256 << 16) | 320, "g5 128x160": (128 << 16) | 160, "g3 256x160": (256 << 16) | 160, "t 128x128": (128 << 16) | 128}


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


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    cases = [  # (M, N, K, residual, geglu)
        (131072, 320, 320, True, False), (131072, 320, 320, False, False), (131072, 960, 320, False, False),
        (65536, 320, 320, True, False), (65536, 320, 128, True, False), (131072, 2560, 320, False, True),
        (32768, 640, 640, True, False), (32768, 1920, 640, False, False), (32768, 5120, 640, False, True),
        (131072, 320, 1280, True, False), (32768, 640, 2560, True, False), (8192, 1280, 1280, True, False)]
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
                line += f" {name}: {us:7.1f} us {by / us / 1e6:5.2f} TB/s {fl / us / 1e6:6.0f} TF |"
            except RuntimeError as e:
                line += f" {name}: n/a |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
