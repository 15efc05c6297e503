#!/bin/bash
# Round 4, call 14: plan vs op-by-op equivalence at mid size, the full-width step fixtures through the plans, C4 / C5 bench A/B.
mkdir -p gpurun_out/c14
timeout 600 python -m pytest tests/test_zz_dit_gpu.py -q -k "plan_matches" > gpurun_out/c14/equiv.txt 2>&1; echo "equiv rc=$?"
tail -12 gpurun_out/c14/equiv.txt | cut -c1-700
timeout 1200 python -m pytest tests/test_fullsize_parity_gpu.py -q -k "pixart or sd3" > gpurun_out/c14/full.txt 2>&1; echo "full rc=$?"
tail -8 gpurun_out/c14/full.txt | cut -c1-600
for arch in pixart sd3; do
  for plan in 1 0; do
    FDMI_DIT_PLAN=$plan timeout 600 python bench.py --arch $arch --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c14/bench_${arch}_plan$plan.json 2> gpurun_out/c14/bench_${arch}_plan$plan.err
    echo "bench $arch plan=$plan rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c14/bench_${arch}_plan$plan.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "unit", "ms_per_step")}, d.get("roofline", {}).get("whole_step"))
except Exception as e:
    print("no line", e); print(open("gpurun_out/c14/bench_${arch}_plan$plan.err").read()[-1500:])
PY
  done
done
