#!/bin/bash
# Round-6 closing profile of the FINAL tree: scripts/profile_c2.sh 6 (bench line, rocprofv3 kernel stats, FETCH / WRITE / TCC / GRBM counter
# passes keyed by source hash), then the per-shape tables of the C4 / C5 profiled steps.
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
bash scripts/profile_c2.sh 6 2>&1 | tail -30
mkdir -p gpurun_out/r6s3shapes
for arch in pixart sd3; do
  FDMI_BENCH_SHAPES=gpurun_out/r6s3shapes/$arch.csv timeout 900 python bench.py --arch $arch --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r6s3shapes/$arch.json 2> gpurun_out/r6s3shapes/$arch.err
  python scripts/shape_table.py gpurun_out/r6s3shapes/$arch.csv > gpurun_out/r6s3shapes/shape_table_$arch.txt 2>&1
  head -30 gpurun_out/r6s3shapes/shape_table_$arch.txt | cut -c1-150
done
