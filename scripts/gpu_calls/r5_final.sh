#!/bin/bash
# Round-5 closing GPU call: the whole GPU suite as the driver runs it, smoke(), then the C3 / C4 / C5 bench lines of the closing tree on
# ONE box (the C2 line and its profile: scripts/profile_c2.sh 5, the call before this one).
set -u
out=gpurun_out/r5final2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/*_parity.txt gpurun_out/batch_invariance.txt gpurun_out/test_durations.txt gpurun_out/kernel_diag.txt
SECONDS=0
timeout 1500 python -m pytest tests/ -x -q -m gpu > "$out/01_pytest.log" 2>&1
echo "   pytest exit $? ; suite wall: $SECONDS s"; tail -4 "$out/01_pytest.log" | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" > "$out/02_smoke.log" 2>&1; tail -3 "$out/02_smoke.log" | cut -c1-300
for arch in sdxl pixart sd3; do
  timeout 900 python bench.py --arch $arch --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_${arch}.json" 2> "$out/bench_${arch}.err"
  python - <<PY
import json
try:
    l=[x for x in open("$out/bench_${arch}.json") if x.startswith('{')][-1]
    d=json.loads(l); print("$arch", round(d['ms_per_step'],1), 'ms', round(d['value'],3), d['unit'], 'whole-step frac', round(d['roofline']['whole_step']['frac_of_peak'],4))
except Exception as e:
    print("$arch failed", e)
PY
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$out/bench_sd15.json" 2> "$out/bench_sd15.err"
python -c "
import json
d=json.loads([x for x in open('$out/bench_sd15.json') if x.startswith('{')][-1]); print('sd15', round(d['ms_per_step'],1), 'ms', round(d['value'],2), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_over_algorithmic'])"
