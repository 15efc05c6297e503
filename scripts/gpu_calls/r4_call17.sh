#!/bin/bash
# Round 4, call 17: the whole GPU suite on the closing tree (transformer plans, rank padding, wgrad operand swap) + the four bench lines.
mkdir -p gpurun_out/c17
timeout 1300 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/c17/suite.txt 2>&1; echo "suite rc=$?"
tail -15 gpurun_out/c17/suite.txt | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c17/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/c17/smoke.txt
for arch in sd15 pixart sd3 sdxl; do
  timeout 600 python bench.py --arch $arch --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c17/bench_${arch}.json 2> gpurun_out/c17/bench_${arch}.err
  echo "bench $arch rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c17/bench_${arch}.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "unit", "ms_per_step")}, d.get("roofline", {}).get("whole_step"), "traffic", d.get("roofline", {}).get("traffic"))
except Exception as e:
    print("no line", e); print(open("gpurun_out/c17/bench_${arch}.err").read()[-1500:])
PY
done
