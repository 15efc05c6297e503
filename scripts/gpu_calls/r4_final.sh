#!/bin/bash
# Round-4 closing GPU call: the whole GPU suite, smoke(), the C3 / C4 / C5 bench lines, then every measured artefact of C2
# (scripts/profile_c2.sh 4: bench line, kernel statistics, FETCH / WRITE / TCC / GRBM counter passes)
set -u
out=gpurun_out/r4final
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/*_parity.txt
timeout 2700 python -m pytest tests -q -m gpu > "$out/01_pytest.log" 2>&1; tail -6 "$out/01_pytest.log" | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" > "$out/02_smoke.log" 2>&1; tail -3 "$out/02_smoke.log" | cut -c1-300
for arch in sdxl pixart sd3; do
  timeout 600 python bench.py --arch $arch --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_${arch}.json" 2> "$out/bench_${arch}.err"
  python - <<PY
import json
try:
    l=[x for x in open("$out/bench_${arch}.json") if x.startswith('{')][-1]
    d=json.loads(l); print("$arch", round(d['ms_per_step'],1), 'ms', round(d['value'],3), d['unit'])
except Exception as e:
    print("$arch failed", e)
PY
done
bash scripts/profile_c2.sh 4 2>&1 | grep -v "warning\|hipEvent\|\^~\|^ *[0-9]* |" | tail -40
