#!/bin/bash
# Round-3 seventh GPU call: bench line + rocprofv3 kernel stats of the current build (to pick the next targets), PixArt / SD3 legs.
set -u
out=gpurun_out/r3c7
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
python bench.py --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; tail -c 1500 "$out/bench.json"
rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -o r3 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/stats_bench.json" 2> "$out/stats.err"
python scripts/rocprof_to_profiles.py --round 93 --steps 4 --stats-dir "$out/stats" > "$out/summary.txt" 2>&1
cp profiles/r93_kernel_stats.csv "$out/" ; rm -f profiles/r93_*
find "$out" -name '*kernel_trace.csv' -delete
head -45 "$out/r93_kernel_stats.csv" | cut -c1-150
for arch in pixart sd3 sdxl; do
  timeout 600 python bench.py --arch $arch --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_$arch.json" 2> "$out/bench_$arch.err"; tail -c 600 "$out/bench_$arch.json"; echo
done
