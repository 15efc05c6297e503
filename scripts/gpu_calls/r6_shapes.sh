#!/bin/bash
# Round-6: per-shape table of the profiled C2 step (fdmi_prof_dump -> scripts/shape_table.py) + the C2 line after ops.upload
set -u
out=gpurun_out/r6shapes
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
FDMI_BENCH_SHAPES=$out/shapes_c2.csv timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > "$out/bench.log" 2>&1; echo "exit $?"
tail -1 "$out/bench.log" | cut -c1-200
python scripts/shape_table.py "$out/shapes_c2.csv" > "$out/shape_table_c2.txt"; head -70 "$out/shape_table_c2.txt"
