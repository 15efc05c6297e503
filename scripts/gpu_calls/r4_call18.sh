#!/bin/bash
# Round 4, call 18: same-box A/B of the rank padding (44) and the wgrad operand swap (43) on C4.
mkdir -p gpurun_out/c18
for t in "" "44=1" "43=1"; do
  FDMI_TUNE=$t timeout 70 python bench.py --arch pixart --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/c18/b_$t.json 2>/dev/null
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c18/b_$t.json").read().strip().splitlines()[-1]); print("tune '$t'", round(d["ms_per_step"], 1), "ms")
except Exception as e:
    print("tune '$t' no line", e)
PY
done
