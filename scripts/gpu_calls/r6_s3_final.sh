#!/bin/bash
# Round-6 closing validation: the whole GPU suite as the driver runs it, the default C2 bench line, and the C3 / C4 / C5 lines on ONE box.
set -u
out=gpurun_out/r6s3final
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/*_parity.txt gpurun_out/fp32_gate.txt gpurun_out/batch_invariance.txt gpurun_out/test_durations.txt gpurun_out/kernel_diag.txt gpurun_out/deterministic.txt
SECONDS=0
timeout 1700 python -m pytest tests/ -x -q -m gpu > "$out/01_pytest.log" 2>&1
echo "   pytest exit $? ; suite wall: $SECONDS s"; tail -4 "$out/01_pytest.log" | cut -c1-300
for arch in sdxl pixart sd3; do
  timeout 900 python bench.py --arch $arch --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2> "$out/$arch.err" | grep '^{"metric"' > "$out/r6_bench_final_$arch.json"
  python -c "
import json
d=json.loads(open('$out/r6_bench_final_$arch.json').readline())
print('$arch', round(d['ms_per_step'],1), 'ms', round(d['value'],2), 'images/s', round(d['roofline']['whole_step']['frac_of_peak'],3))
"
done
timeout 900 python bench.py --steps 20 --warmup 5 > "$out/02_bench.log" 2> "$out/02_bench.err"
grep '^{"metric"' "$out/02_bench.log" | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('C2', d['ms_per_step'], d['value'], d['roofline']['frac'])"
cp gpurun_out/*_parity.txt gpurun_out/fp32_gate.txt gpurun_out/batch_invariance.txt gpurun_out/deterministic.txt "$out/" 2>/dev/null
