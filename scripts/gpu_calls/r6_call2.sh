#!/bin/bash
# Round-6 call 2: kernel trace of the timed C2 steps for the gap attribution (scripts/trace_gaps2.py), the HBM-bound kernels alone
# on the chip (scripts/kbench.py norm), the row kernels' rates on this box (scripts/rowbench.py).
set -u
out=gpurun_out/r6c2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_trace timeout -s KILL 500 rocprofv3 --kernel-trace -f csv -d "$out/tg" -o tg -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary
python scripts/trace_gaps2.py "$out/tg" 2 3 > "$out/trace_gaps2.txt" 2>&1
cat "$out/trace_gaps2.txt"
# keep a compact copy of the timed window's trace for offline analysis (start, end, queue, kernel)
python - "$out/tg" "$out/trace_compact.csv" <<'PY'
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
names = {}
with open(sys.argv[2], "w") as fh:
    for r in rows:
        n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:60]
        i = names.setdefault(n, len(names))
        fh.write(f"{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - t0},{r.get('Queue_Id', 0)},{i},{r.get('Grid_Size', 0)}\n")
with open(sys.argv[2] + ".names", "w") as fh:
    for n, i in names.items():
        fh.write(f"{i},{n}\n")
PY
gzip -f "$out/trace_compact.csv"
find "$out/tg" -name "*.csv" -size +1M -delete
run 02_kbench_norm timeout 300 python scripts/kbench.py norm 30
cat "$out/02_kbench_norm.log"
run 03_rowbench timeout 400 python scripts/rowbench.py 20
cat "$out/03_rowbench.log" | cut -c1-330
