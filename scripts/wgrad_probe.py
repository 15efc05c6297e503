#!/usr/bin/env python
"""GPU dev tool: which (row, column) of the operands meets which in `wgrad_tn` -- one-hot probes through the product path.
X = one-hot at (m*, a), Y[m][:] = m + 1: a consistent kernel returns C[a][:] = m* + 1 and zeros elsewhere; anything else names the
column the one-hot landed in and the row of Y it was paired with (the lane / granule / reduction-slot mapping that is off)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from flash_diffusion_amd import ops  # noqa: E402


def probe(M, N1, N2, swap=False):
    bad = 0
    for ms in list(range(0, min(M, 64))) + [M - 1]:
        for a in (0, 1, 5, 17, 33, N1 - 1):
            x = torch.zeros(M, N1, device="cuda", dtype=torch.bfloat16)
            x[ms, a] = 1
            y = (torch.arange(M, device="cuda", dtype=torch.float32) + 1)[:, None].expand(M, N2).contiguous().bfloat16()
            X, Y = (y, x) if swap else (x, y)
            c = torch.zeros(X.shape[1], Y.shape[1], device="cuda")
            ops.wgrad_tn(X, Y, c)
            c = c.t() if swap else c
            want = torch.zeros_like(c)
            want[a] = float(torch.tensor(ms + 1.0).bfloat16())
            if not torch.equal(c, want):
                bad += 1
                if bad <= 12:
                    nz = c.nonzero()
                    rows = sorted(set(nz[:, 0].tolist()))[:6]
                    vals = sorted(set(c[nz[:, 0], nz[:, 1]].tolist()))[:6]
                    print(f"  M={M} N1={N1} N2={N2} swap={int(swap)} one-hot (m={ms}, col={a}): nonzero rows {rows} values {vals} "
                          f"(want row {a} value {ms + 1})")
    print(f"probe M={M} N1={N1} N2={N2} swap={int(swap)}: {'OK' if not bad else str(bad) + ' probes wrong'}", flush=True)


if __name__ == "__main__":
    for (M, N1, N2) in [(64, 128, 128), (64, 64, 128), (200, 128, 320), (200, 64, 200)]:
        probe(M, N1, N2)
        probe(M, N1, N2, swap=True)
