#!/usr/bin/env python
"""GPU dev tool: us per launch of `wgrad_tn` (C[n1][n2] += sum_m X[m][n1] Y[m][n2]) at the LoRA weight-gradient shapes of C2 / C4 / C5,
for each value of the developer knobs given on the command line (43: operand swap of the old kernel, 45: blocks the row split aims
for, 46: 1 = round 4's register-transposing kernel, 0 = round 5's streaming kernel):

  python scripts/wgrad_rates.py                 # defaults
  python scripts/wgrad_rates.py 45=512,2048,4096 43=0,1
  FDMI_LIB=/path/to/other/libfdmi.so python scripts/wgrad_rates.py     # another build of the library, same box
  python scripts/wgrad_rates.py groups          # fdmi_wgrad_tn_group: us per GROUP, one launch vs one launch per product (switch 47 = 1)

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


# (name, [(X operand, Y operand) as (buffer, first column, columns)]): the groups csrc/unet.hip launches
def group_cases():
    def qkv(M, C, r):
        return [p for s in range(3) for p in ((("dy", s * C, C), ("t3", s * r, r)), (("dt3", s * r, r), ("x", 0, C)))], \
            {"dy": 3 * C, "t3": 3 * r, "dt3": 3 * r, "x": C}

    def pair(M, cin, cout, r):
        return [(("dy", 0, cout), ("t", 0, r)), (("dt", 0, r), ("x", 0, cin))], {"dy": cout, "t": r, "dt": r, "x": cin}
    return [("C2 qkv 64x64 (M 65536, C 320, r 128)", 65536, *qkv(65536, 320, 128)),
            ("C2 qkv 32x32 (M 16384, C 640, r 128)", 16384, *qkv(16384, 640, 128)),
            ("C2 qkv 16x16 (M 4096, C 1280, r 128)", 4096, *qkv(4096, 1280, 128)),
            ("C2 to_out 64x64 (M 65536, 320 -> 320, r 128)", 65536, *pair(65536, 320, 320, 128)),
            ("C2 to_k 77 tokens (M 1232, 768 -> 320, r 128)", 1232, *pair(1232, 768, 320, 128)),
            ("C4 qkv (M 32768, C 1152, r 64)", 32768, *qkv(32768, 1152, 64)),
            ("C4 ff.net.0 (M 32768, 1152 -> 4608, r 64)", 32768, *pair(32768, 1152, 4608, 64)),
            ("C4 ff.net.2 (M 32768, 4608 -> 1152, r 64)", 32768, *pair(32768, 4608, 1152, 64))]


def groups(reps=20):
    print("us per group: one launch | one launch per product (switch 47 = 1) | products")
    for name, M, probs, bufs in group_cases():
        t = {k: torch.randn(M, w, device="cuda").bfloat16() for k, w in bufs.items()}
        view = lambda o: t[o[0]][:, o[1]:o[1] + o[2]]
        work = [(view(x), view(y), torch.zeros(x[2], y[2], device="cuda")) for x, y in probs]
        res = []
        for single in (0, 1):
            _lib.lib().fdmi_tune_set(47, single)
            for _, _, c in work:
                c.zero_()
            for _ in range(3):
                ops.wgrad_tn_group(work)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                ops.wgrad_tn_group(work)
            b.record()
            torch.cuda.synchronize()
            err = max(float(((c / (reps + 3)) - x.float().t() @ y.float()).norm() / (x.float().t() @ y.float()).norm()) for x, y, c in work)
            res.append((a.elapsed_time(b) / reps * 1e3, err))
        _lib.lib().fdmi_tune_set(47, 0)
        print(f"{name:52s} {res[0][0]:8.1f} | {res[1][0]:8.1f} | {len(work)}   (max rel err {max(e for _, e in res):.1e})", flush=True)


def main():
    if sys.argv[1:] == ["groups"]:
        return groups()
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
