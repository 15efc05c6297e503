#!/usr/bin/env python
"""Per-kernel sums of the counters of one rocprofv3 --pmc pass (dev tool):  python scripts/pmc_by_kernel.py DIR [top]
Prints, per kernel name (template arguments kept, argument lists dropped): launches, total ms, each counter's total, and -- when the
LDS counters are present -- bank-conflict cycles as a share of the LDS-active cycles."""
import collections
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)[:70]


def main(d, top=40):
    dur = {}
    name = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            i = int(r["Dispatch_Id"])
            dur[i] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            name[i] = short(r["Kernel_Name"])
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            i = int(r["Dispatch_Id"])
            k = short(r["Kernel_Name"])
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(i)
    ms = collections.defaultdict(float)
    for i, k in name.items():
        ms[k] += dur[i]
    counters = sorted({c for k in cnt for c in cnt[k]})
    print(f"{'kernel':70s} {'n':>6s} {'ms':>9s} " + " ".join(f"{c[-18:]:>18s}" for c in counters) + "  conflict/LDS-active")
    for k in sorted(cnt, key=lambda k: -ms[k])[:top]:
        c = cnt[k]
        share = ""
        if c.get("SQ_LDS_IDX_ACTIVE"):
            share = f"{100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:6.1f} %"
        print(f"{k:70s} {len(seen[k]):6d} {ms[k]:9.3f} " + " ".join(f"{c.get(x, 0):18.0f}" for x in counters) + "  " + share)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
