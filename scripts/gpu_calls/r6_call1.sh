#!/bin/bash
# Round-6 first GPU call: baseline C2 line of the round-5 tree on this box, SQ / LDS / GRBM counter passes over the GEMM and conv
# kernels (VERDICT r5 item 1a / missing 4), the MFMA-rate micro-benchmark on random and zero operands with its clock (weak 12).
set -u
out=gpurun_out/r6c1
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_bench timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
run 02_gemm_pmc bash scripts/gemm_pmc.sh "$out/gemm_pmc"
cp "$out/gemm_pmc/table.txt" "$out/gemm_sq_table.txt"
run 03_mfma_rate scripts/ubench/mfma_rate
cat "$out/03_mfma_rate.log"
run 04_mfma_clock timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$out/mfma_clock" -o a -- scripts/ubench/mfma_rate
python - "$out/mfma_clock" <<'PY' > "$out/mfma_clock.txt" 2>&1
import csv, glob, os, sys
d = sys.argv[1]
tr, ct = {}, {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        tr[int(r["Dispatch_Id"])] = (r["Kernel_Name"][:40], (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            ct[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
for i in sorted(tr):
    n, us = tr[i]
    print(f"dispatch {i:3d} {n:40s} {us:10.1f} us  GRBM_GUI_ACTIVE {ct.get(i, 0):14.0f}  clock {ct.get(i, 0) / us / 1e3:5.2f} GHz")
PY
cat "$out/mfma_clock.txt"
head -150 "$out/gemm_sq_table.txt"
