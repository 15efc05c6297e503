#!/bin/bash
# Round-3 tenth GPU call: split-K of the transformer denoisers' projections only for M <= 4096 rows: A/B against "never" and "always"
set -u
out=gpurun_out/r3c10
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for arch in pixart sd3; do
  for rows in 4096 0 100000000; do
    FDMI_DIT_SPLITK_ROWS=$rows timeout 600 python bench.py --arch $arch --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_${arch}_$rows.json" 2> "$out/bench_${arch}_$rows.err"
    python - <<PY
import json
l=[x for x in open("$out/bench_${arch}_$rows.json") if x.startswith('{')][-1]
d=json.loads(l); print("$arch auto split-K for M <= $rows:", round(d['ms_per_step'],1), 'ms', round(d['value'],3), d['unit'])
PY
  done
done
