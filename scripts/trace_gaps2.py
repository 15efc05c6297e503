#!/usr/bin/env python
"""Where does the GPU idle inside the TIMED steps?  (VERDICT r5 item 2 ii: attribute the gaps.)

Reads a rocprofv3 --kernel-trace CSV of `python bench.py --steps K --warmup W --no-cpu-baseline --no-secondary`, windows on whole
steps between `adamw_kernel` launches (one per step; the window runs from the end of the (W+1)-th to the end of the (W+K)-th one, so
the serial profiled step that follows the timed ones and the plan build of the first step are out), and reports
  * the union-busy / idle / overlap shares and the per-queue busy time;
  * idle gaps (no kernel of ANY queue running) by size;
  * per queue: the gaps between consecutive kernels of that queue, by size -- a queue's own launch cadence;
  * the (previous kernel -> next kernel) pairs that own the most idle time, and the kernels that most often run ALONE
    (nothing of the other queue beside them) for short durations -- the launch-bound stretches.

  rocprofv3 --kernel-trace -f csv -d gpurun_out/tg -o tg -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary
  python scripts/trace_gaps2.py gpurun_out/tg [W K]"""
import collections
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    return n[:56]


def main(d, W=2, K=3):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no *kernel_trace.csv under {d}"
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])) for r in rows)
    marks = [e[1] for e in ev if e[3].startswith("adamw_kernel")]
    print(f"{len(ev)} dispatches, {len(marks)} adamw_kernel launches")
    if len(marks) >= W + K + 1:
        t0, t1 = marks[W], marks[W + K]
    else:
        t0, t1 = marks[0], marks[-1]
        K = len(marks) - 1
    ev = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    span = (t1 - t0) / 1e6
    print(f"window: {K} steps, {span:.2f} ms = {span / K:.2f} ms per step, {len(ev)} dispatches = {len(ev) / K:.0f} per step")
    pts = sorted([(s, 1, i) for i, (s, _, _, _) in enumerate(ev)] + [(e, -1, i) for i, (_, e, _, _) in enumerate(ev)])
    depth, last, busy, multi, idle = 0, t0, 0, 0, 0
    gaps = []          # (length, index of the kernel that ended last before it, index of the kernel that starts after it)
    last_end_idx = None
    for t, dlt, i in pts:
        dt = t - last
        if depth == 0 and dt > 0:
            idle += dt
            if dlt == 1:
                gaps.append((dt, last_end_idx, i))
        elif depth >= 1:
            busy += dt
            if depth >= 2:
                multi += dt
        depth += dlt
        if dlt == -1:
            last_end_idx = i
        last = t
    perq = collections.defaultdict(float)
    for s, e, q, _ in ev:
        perq[q] += (e - s) / 1e6
    print(f"  some kernel running   {busy / 1e6 / K:8.2f} ms/step ({busy / 1e4 / span:5.1f} %)")
    print(f"  two or more in flight {multi / 1e6 / K:8.2f} ms/step ({multi / 1e4 / span:5.1f} %)")
    print(f"  idle (no kernel)      {idle / 1e6 / K:8.2f} ms/step ({idle / 1e4 / span:5.1f} %) in {len(gaps) / K:.0f} gaps per step")
    print("  kernel time per queue, ms/step:", {q: round(v / K, 2) for q, v in sorted(perq.items(), key=lambda kv: -kv[1])})
    bins = ((0, 1e3), (1e3, 3e3), (3e3, 1e4), (1e4, 3e4), (3e4, 1e5), (1e5, 1e12))
    for lo, hi in bins:
        g = [x[0] for x in gaps if lo <= x[0] < hi]
        print(f"  idle gaps {lo / 1e3:6.0f} - {hi / 1e3:9.0f} us: {len(g) / K:8.1f} per step, {sum(g) / 1e6 / K:7.3f} ms/step")
    # per-queue cadence
    byq = collections.defaultdict(list)
    for s, e, q, n in ev:
        byq[q].append((s, e, n))
    for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:3]:
        lst.sort()
        gq = [(lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)]
        print(f"  queue {q}: {len(lst) / K:.0f} kernels per step; gaps between ITS consecutive kernels:")
        for lo, hi in bins:
            g = [x for x in gq if lo <= x < hi]
            print(f"      {lo / 1e3:6.0f} - {hi / 1e3:9.0f} us: {len(g) / K:8.1f} per step, {sum(g) / 1e6 / K:7.3f} ms/step")
        neg = [x for x in gq if x < 0]
        print(f"      overlapping (next starts before previous ends): {len(neg) / K:.1f} per step")
    pair = collections.defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        if a is None:
            continue
        k = (ev[a][3][:40], ev[b][3][:40])
        pair[k][0] += 1
        pair[k][1] += g
    print("  idle time by (kernel before the gap -> kernel after it), top 25:")
    for k, (n, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"      {t / 1e6 / K:7.3f} ms/step  {n / K:7.1f} gaps/step  avg {t / n / 1e3:6.1f} us   {k[0]}  ->  {k[1]}")
    after = collections.defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        after[ev[b][3][:48]][0] += 1
        after[ev[b][3][:48]][1] += g
    print("  idle time by the kernel that ENDS the gap (= the launch that arrived late), top 15:")
    for k, (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:15]:
        print(f"      {t / 1e6 / K:7.3f} ms/step  {n / K:7.1f} gaps/step  avg {t / n / 1e3:6.1f} us   {k}")
    # the long gaps one by one
    print("  the 12 longest gaps:")
    for g, a, b in sorted(gaps, key=lambda x: -x[0])[:12]:
        print(f"      {g / 1e3:9.1f} us at +{(ev[b][0] - t0) / 1e6:8.2f} ms   {ev[a][3][:40] if a is not None else '-'} (q{ev[a][2] if a is not None else '-'})  ->  {ev[b][3][:40]} (q{ev[b][2]})")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 2, int(a[3]) if len(a) > 3 else 3)
