#!/usr/bin/env python
"""How busy is the GPU during a step?  Reads a rocprofv3 --kernel-trace CSV and reports, over the steady-state tail of the run: the
wall span, the union of the kernels' busy intervals (any queue), the idle time between kernels (no kernel of any queue running),
the time with two or more kernels in flight, the per-queue busy time, and the idle gaps by size -- i.e. what launch gaps cost after
the two-stream overlap, and the budget a graph capture / fewer launches could win back at most.

  rocprofv3 --kernel-trace -f csv -d gpurun_out/tg -o tg -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary
  python scripts/trace_gaps.py gpurun_out/tg [fraction of the run to keep, default 0.4]"""
import collections
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:48]


def main(d, frac=0.4):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no *kernel_trace.csv under {d}"
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows)
    t_end = max(e[1] for e in ev)
    t_beg = ev[0][0]
    cut = t_end - (t_end - t_beg) * frac
    ev = [e for e in ev if e[0] >= cut]
    span = (max(e[1] for e in ev) - ev[0][0]) / 1e6
    # sweep line over start / end events
    pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
    depth, last, busy, multi, idle = 0, pts[0][0], 0, 0, 0
    gaps = []
    for t, dlt in pts:
        dt = t - last
        if depth == 0 and dt > 0:
            idle += dt
            gaps.append(dt)
        elif depth >= 1:
            busy += dt
            if depth >= 2:
                multi += dt
        depth += dlt
        last = t
    perq = collections.defaultdict(float)
    for s, e, q, _ in ev:
        perq[q] += (e - s) / 1e6
    print(f"tail of the run: {len(ev)} dispatches over {span:.2f} ms")
    print(f"  some kernel running   {busy / 1e6:8.2f} ms ({busy / 1e4 / span:5.1f} %)")
    print(f"  two or more in flight {multi / 1e6:8.2f} ms ({multi / 1e4 / span:5.1f} %)")
    print(f"  idle (no kernel)      {idle / 1e6:8.2f} ms ({idle / 1e4 / span:5.1f} %) in {len(gaps)} gaps")
    print("  sum of kernel durations per queue:", {q: round(v, 2) for q, v in sorted(perq.items(), key=lambda kv: -kv[1])})
    for lo, hi in ((0, 1e3), (1e3, 3e3), (3e3, 1e4), (1e4, 1e5), (1e5, 1e12)):
        g = [x for x in gaps if lo <= x < hi]
        print(f"  idle gaps {lo / 1e3:6.0f} - {hi / 1e3:9.0f} us: {len(g):6d}  total {sum(g) / 1e6:7.3f} ms")
    # which kernels precede the long idle gaps
    ends = sorted((e, short(n)) for _, e, _, n in ev)
    print("  (per-queue serial time minus busy union = what the overlap hides)")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
