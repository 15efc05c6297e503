#!/usr/bin/env python
"""Turn the raw rocprofv3 output of scripts/profile_c2.sh into the small, committed artefacts under profiles/:

  python scripts/rocprof_to_profiles.py --round 2 --steps 4 --stats-dir gpurun_out/prof/stats \\
         [--fetch-dir gpurun_out/prof/pmc_fetch --write-dir gpurun_out/prof/pmc_write --pmc-steps 3]

  profiles/rN_kernel_stats.csv   per kernel: launches per step, ms per step, average us   (from *_kernel_stats.csv)
  profiles/rN_pmc_hbm_traffic.csv, profiles/rN_traffic.json   HBM bytes per launch per kernel (from *_counter_collection.csv
      of the FETCH_SIZE and WRITE_SIZE passes; FETCH_SIZE doubled -- the gfx950 correction of MI355X_MICROARCH.md's HBM section)

`--steps` is the number of training steps the profiled command ran in total (warm-up + timed + the bench's profiled leg).
The column names of rocprofv3's CSV files differ a little between ROCm releases; they are looked up by candidates."""
import argparse
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _col(row, *cands):
    for c in cands:
        if c in row:
            return c
    low = {k.lower(): k for k in row}
    for c in cands:
        if c.lower() in low:
            return low[c.lower()]
    raise KeyError(f"none of {cands} in {list(row)}")


def _files(d, suffix):
    fs = sorted(glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True))
    if not fs:
        sys.exit(f"no *{suffix} under {d}")
    return fs


def short(name):
    """'void gemm4_kernel<1, false>(GemmArgs)' -> 'void gemm4_kernel<1, false>' (argument lists and the anonymous namespace
    carry no information here)"""
    name = name.replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip().replace(" [clone .kd]", "").replace(".kd", "")


# bench.py's bucket names for the MFMA kernel families (profiles/README.md "Name mapping")
def bucket(name):
    m = re.search(r"gemm4_kernel<(\d), (?:false|true)(?:, (\d+))?", name)
    if m:
        return f"gemm4_kernel<256x{m.group(2) or '320'},{'conv' if m.group(1) == '1' else 'row'}>"
    m = re.search(r"gemm3_kernel<(\d+), (\d)", name)
    if m:
        return f"gemm3_kernel<256x{m.group(1)},{'conv' if m.group(2) == '1' else 'row'}>"
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d)", name)
    if m:
        return f"gemm_kernel<{m.group(1)},{m.group(2)},{'conv' if m.group(3) == '1' else 'row'}>"
    for k, b in (("attn_fwd32_kernel", "attn_fwd_kernel"), ("attn_bwd_dq32_kernel", "attn_bwd_dq_kernel"),
                 ("attn_bwd_dkv32_kernel", "attn_bwd_dkv_kernel"),       # the 32x32x16 family shares the bench's attention buckets
                 ("wgrad_tn2_kernel", "wgrad_tn_kernel")):               # round 5's streaming kernel: the same bench bucket
        if k in name:
            return b
    for k in ("attn_fwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel", "wgrad_tn_kernel"):
        if k in name:
            return k
    return None


def kernel_stats(stats_dir, steps, out_csv, header):
    agg = defaultdict(lambda: [0, 0.0])
    for f in _files(stats_dir, "kernel_stats.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                n = short(row[_col(row, "Name", "Kernel_Name", "KernelName")])
                agg[n][0] += int(float(row[_col(row, "Calls", "Count")]))
                agg[n][1] += float(row[_col(row, "TotalDurationNs", "TotalDuration", "Total_Duration_Ns")])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out_csv, "w") as fh:
        fh.write(f"# {header}\n")
        fh.write("kernel,launches_per_step,ms_per_step,avg_us\n")
        for n, (calls, ns) in rows:
            fh.write(f"\"{n}\",{calls / steps:.1f},{ns / steps / 1e6:.3f},{ns / calls / 1e3:.1f}\n")
        fh.write(f"# sum of all kernels per step: {sum(v[1] for v in agg.values()) / steps / 1e6:.1f} ms\n")
    return rows


def counter(d, want):
    """{kernel: (launches, mean counter value per launch)} -- the per-dispatch rows of one --pmc pass"""
    agg = defaultdict(lambda: [0, 0.0])
    for f in _files(d, "counter_collection.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row[_col(row, "Counter_Name", "CounterName")] != want:
                    continue
                n = short(row[_col(row, "Kernel_Name", "KernelName", "Name")])
                agg[n][0] += 1
                agg[n][1] += float(row[_col(row, "Counter_Value", "CounterValue", "Value")])
    return {n: (c, v / c) for n, (c, v) in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True, help="training steps the --stats run executed in total")
    ap.add_argument("--stats-dir", required=True)
    ap.add_argument("--fetch-dir")
    ap.add_argument("--write-dir")
    ap.add_argument("--tcc-dir", help="a --pmc TCC_HIT_sum TCC_MISS_sum pass: per-kernel L2 hit rate -> rN_l2_hit_rate.csv")
    ap.add_argument("--grbm-dir", help="a --pmc GRBM_GUI_ACTIVE pass: effective shader clock per kernel -> rN_clock.csv")
    ap.add_argument("--tag", default="", help="file-name suffix, e.g. _gn for a knob run")
    ap.add_argument("--command", default="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary")
    a = ap.parse_args()
    pre = os.path.join(ROOT, "profiles", f"r{a.round}")
    rows = kernel_stats(a.stats_dir, a.steps, f"{pre}_kernel_stats{a.tag}.csv",
                        f"rocprofv3 --kernel-trace --stats of `{a.command}` ({a.steps} steps incl. warm-up + profiled leg; "
                        f"per-step = total/{a.steps}; pack_* = one-time weight packing of the first step)")
    print("top kernels (ms per step):")
    for n, (calls, ns) in rows[:12]:
        print(f"  {ns / a.steps / 1e6:8.3f}  {calls / a.steps:7.1f} x {ns / calls / 1e3:8.1f} us  {n}")
    if a.tcc_dir:   # L2 (TCC) hit rate per kernel: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)  (MI355X_MICROARCH.md, L2 section)
        hit, miss = counter(a.tcc_dir, "TCC_HIT_sum"), counter(a.tcc_dir, "TCC_MISS_sum")
        with open(f"{pre}_l2_hit_rate{a.tag}.csv", "w") as fh:
            fh.write("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace (own pass); requests per launch, "
                     "hit rate = hit / (hit + miss).  FETCH_SIZE counts the misses' traffic to the fabric, Infinity-Cache hits "
                     "included -- a low rate here with FETCH far above the algorithmic bytes is re-reads the 4 MiB per-XCD L2 "
                     "did not hold\n")
            fh.write("kernel,launches,tcc_hit_per_launch,tcc_miss_per_launch,l2_hit_rate\n")
            for n, (c, h) in sorted(hit.items(), key=lambda kv: -kv[1][0] * (kv[1][1] + miss.get(kv[0], (0, 0.0))[1])):
                m = miss.get(n, (0, 0.0))[1]
                if h + m > 0:
                    fh.write(f"\"{n}\",{c},{h:.0f},{m:.0f},{h / (h + m):.4f}\n")
        print("wrote", f"{pre}_l2_hit_rate{a.tag}.csv")
    if a.grbm_dir:   # effective clock = GRBM_GUI_ACTIVE / kernel wall time (MI355X_MICROARCH.md, "DVFS give-back")
        agg = defaultdict(lambda: [0, 0.0, 0.0])
        for f in _files(a.grbm_dir, "counter_collection.csv"):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if row[_col(row, "Counter_Name", "CounterName")] != "GRBM_GUI_ACTIVE":
                        continue
                    try:
                        dur = float(row[_col(row, "End_Timestamp", "EndTimestamp")]) - float(row[_col(row, "Start_Timestamp", "StartTimestamp")])
                    except KeyError:
                        dur = 0.0
                    n = short(row[_col(row, "Kernel_Name", "KernelName", "Name")])
                    agg[n][0] += 1
                    agg[n][1] += float(row[_col(row, "Counter_Value", "CounterValue", "Value")])
                    agg[n][2] += dur
        with open(f"{pre}_clock{a.tag}.csv", "w") as fh:
            fh.write("# rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace (own pass): busy cycles per launch as reported and the dispatch's "
                     "duration from the same rows; GHz = cycles / ns.  The counter is summed over the chip's 8 XCDs when the value "
                     "is ~8x a plausible clock: ghz_per_xcd = GHz / 8 is then the shader clock the kernel ran at\n")
            fh.write("# ONLY kernels whose launches average >= 200 us are listed (VERDICT r4 weak 10): GRBM_GUI_ACTIVE also counts the "
                     "dispatch's ramp on either side of the timestamps, so cycles / duration of a short launch exceeds the part's "
                     "2.4 GHz maximum (round 4's file showed 2.5 - 3.3 GHz for launches under 100 us) and is not a clock\n")
            fh.write("kernel,launches,avg_us,grbm_gui_active_per_launch,ghz_raw,ghz_per_xcd\n")
            for n, (c, cyc, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
                if ns > 0 and ns / c >= 200e3:
                    fh.write(f"\"{n}\",{c},{ns / c / 1e3:.1f},{cyc / c:.0f},{cyc / ns:.3f},{cyc / ns / 8:.3f}\n")
        print("wrote", f"{pre}_clock{a.tag}.csv")
    if not (a.fetch_dir and a.write_dir):
        return
    # FETCH_SIZE / WRITE_SIZE are reported in KB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE x 2 on gfx950
    fetch, write = counter(a.fetch_dir, "FETCH_SIZE"), counter(a.write_dir, "WRITE_SIZE")
    if not fetch:   # the raw counters FETCH_SIZE is derived from on gfx950 (counter_defs.yaml), collected when the derived pass aborts:
        # FETCH_SIZE [KB] = (BUBBLE * 128 + (RDREQ - BUBBLE - RDREQ_32B) * 64 + RDREQ_32B * 32) / 1024
        rd, r32, bub = (counter(a.fetch_dir, n) for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"))
        fetch = {n: (c, (bub.get(n, (0, 0.0))[1] * 128 + (v - bub.get(n, (0, 0.0))[1] - r32.get(n, (0, 0.0))[1]) * 64
                         + r32.get(n, (0, 0.0))[1] * 32) / 1024) for n, (c, v) in rd.items()}
    out = {"source": f"profiles/r{a.round}_pmc_hbm_traffic{a.tag}.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                     "fetch doubled per the gfx950 correction)", "kernels": {}}
    with open(f"{pre}_pmc_hbm_traffic{a.tag}.csv", "w") as fh:
        fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); KB per launch as reported; "
                 "corrected_fetch = 2 x reported (gfx950), WRITE_SIZE as reported\n")
        fh.write("kernel,launches,fetch_KB_per_launch_reported,fetch_MB_per_launch_corrected,write_MB_per_launch\n")
        per_bucket = defaultdict(lambda: [0, 0.0, 0.0])
        for n, (c, kb) in sorted(fetch.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            wkb = write.get(n, (0, 0.0))[1]
            fmb, wmb = 2 * kb * 1024 / 1e6, wkb * 1024 / 1e6
            fh.write(f"{n},{c},{kb:.1f},{fmb:.2f},{wmb:.2f}\n")
            b = bucket(n)
            if b:
                per_bucket[b][0] += c
                per_bucket[b][1] += c * fmb
                per_bucket[b][2] += c * wmb
    for b, (c, f, w) in per_bucket.items():
        out["kernels"][b] = {"hbm_bytes_per_launch": (f + w) / c * 1e6, "fetch_MB_corrected": round(f / c, 2),
                             "write_MB": round(w / c, 2), "launches_sampled": c}
    sys.path.insert(0, ROOT)
    from flash_diffusion_amd import _lib
    out["csrc_sha"] = _lib.source_hash()   # the whole build, for the record
    for b in out["kernels"]:               # bench.py reports a family's traffic only while ITS sources are the measured ones
        out["kernels"][b]["src_sha"] = _lib.kernel_source_hash(b)
    with open(f"{pre}_traffic{a.tag}.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", f"{pre}_traffic{a.tag}.json")


if __name__ == "__main__":
    main()
