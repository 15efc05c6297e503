#!/bin/bash
# Round-5 sixth GPU call: the WHOLE GPU suite under xdist (per-test durations -> gpurun_out/test_durations.txt), then the C4 line with
# the vectorised fp32-stream epilogue (both sides of knob 49), and the C2 / C3 lines.
set -u
out=gpurun_out/r5c6
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
rm -f gpurun_out/*_parity.txt gpurun_out/batch_invariance.txt gpurun_out/test_durations.txt
SECONDS=0
run 01_pytest_all timeout 1500 python -m pytest tests -q -m gpu -rxXsf
echo "   suite wall: $SECONDS s"
tail -12 "$out/01_pytest_all.log" | cut -c1-400
sort -rn gpurun_out/test_durations.txt | head -30
awk '{w[$4]+=$1} END {for (k in w) print k, w[k]}' gpurun_out/test_durations.txt
for v in "" "49=1"; do
  run 02_bench_pixart_$v env FDMI_TUNE=$v timeout 900 python bench.py --arch pixart --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
done
run 03_bench_c2 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
run 04_bench_sdxl timeout 900 python bench.py --arch sdxl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5c6/0[234]_bench*.log")):
    try:
        d=json.loads([x for x in open(f) if x.startswith('{')][-1]); print(f, round(d['ms_per_step'],1),'ms', round(d['value'],2), d['unit'])
    except Exception as e: print(f,'failed',e)
PY
