#!/usr/bin/env python
"""Who issues the runtime's fill / copy kernels?  Reads a rocprofv3 --kernel-trace CSV (one step is enough) and prints, for every
__amd_rocclr_fillBufferAligned / __amd_rocclr_copyBuffer dispatch, a histogram of (previous kernel, next kernel) on the same queue,
plus the distribution of grid sizes (the byte count of the memset).

  rocprofv3 --kernel-trace -f csv -d gpurun_out/ft -o ft -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary
  python scripts/fill_sources.py gpurun_out/ft"""
import collections
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:60]


def main(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no *kernel_trace.csv under {d}"
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last third of the run = the steady-state step(s), past weight packing
    rows = rows[len(rows) * 2 // 3:]
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r.get("Queue_Id", "0")].append(r)
    for kind in ("fillBufferAligned", "copyBuffer"):
        hist = collections.Counter()
        sizes = collections.Counter()
        for q, rs in byq.items():
            for i, r in enumerate(rs):
                if kind in r["Kernel_Name"]:
                    p = short(rs[i - 1]["Kernel_Name"]) if i else "-"
                    n = short(rs[i + 1]["Kernel_Name"]) if i + 1 < len(rs) else "-"
                    hist[(p, n)] += 1
                    sizes[(r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))] += 1
        print(f"== {kind}: {sum(hist.values())} dispatches in the sampled tail ({len(rows)} dispatches)")
        for (p, n), c in hist.most_common(25):
            print(f"  {c:5d}  after {p:60s} before {n}")
        print("  grid sizes:", sizes.most_common(12))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ft")
