#!/usr/bin/env python
"""GPU dev tool: us per launch of `wgrad_tn` (C[n1][n2] += sum_m X[m][n1] Y[m][n2]) at the LoRA weight-gradient shapes of C2 / C4 / C5,
for each value of the developer knobs given on the command line (43: operand swap of the old kernel, 45: blocks the row split aims
for, 46: 1 = round 4's register-transposing kernel, 0 = round 5's streaming kernel):

  python scripts/wgrad_rates.py                 # defaults
  python scripts/wgrad_rates.py 45=512,2048,4096 43=0,1
  FDMI_LIB=/path/to/other/libfdmi.so python scripts/wgrad_rates.py     # another build of the library, same box

(round 4's table: profiles/r4_wgrad_tn_rates.txt)"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flash_diffusion_amd import _lib, ops  # noqa: E402

SHAPES = [(32768, 1152, 64), (32768, 64, 1152), (32768, 4608, 64), (32768, 64, 4608), (65536, 320, 128), (65536, 128, 320),
          (16384, 640, 128), (16384, 128, 640), (16384, 1536, 64), (4096, 1280, 128), (4096, 128, 1280), (1232, 128, 768)]


def rate(M, N1, N2, reps=20):
    x = torch.randn(M, N1, device="cuda").bfloat16()
    y = torch.randn(M, N2, device="cuda").bfloat16()
    c = torch.zeros(N1, N2, device="cuda")
    for _ in range(3):
        ops.wgrad_tn(x, y, c)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.wgrad_tn(x, y, c)
    b.record()
    torch.cuda.synchronize()
    ref = x.float().t() @ y.float()
    err = float(((c / (reps + 3)) - ref).norm() / ref.norm())
    return a.elapsed_time(b) / reps * 1e3, err


def main():
    knobs = {}
    for a in sys.argv[1:]:
        k, vs = a.split("=")
        knobs[int(k)] = [int(v) for v in vs.split(",")]
    keys = sorted(knobs)
    print("us per launch (relative error of the accumulated result) at (M, N1, N2) =", " ".join(str(s) for s in SHAPES))
    for combo in itertools.product(*[knobs[k] for k in keys]) if keys else [()]:
        for k, v in zip(keys, combo):
            _lib.lib().fdmi_tune_set(k, v)
        res = [rate(*s) for s in SHAPES]
        print(" ".join(f"{k}={v}" for k, v in zip(keys, combo)) or "defaults", " ".join(f"{u:7.1f}" for u, _ in res),
              f"(max rel err {max(e for _, e in res):.1e})" + ("  WRONG RESULT at " + str([s for s, (_, e) in zip(SHAPES, res) if e >= 2e-2])
                                                              if any(e >= 2e-2 for _, e in res) else ""), flush=True)


if __name__ == "__main__":
    main()
