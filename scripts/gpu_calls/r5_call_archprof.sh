#!/bin/bash
# Round-5: rocprofv3 kernel statistics of the C4 (PixArt) and C5 (SD3) steps of the closing tree (stats pass only; the bench lines are
# profiles/r5_bench_final_*.json).
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for arch in pixart sd3; do
  out=gpurun_out/prof_$arch
  rm -rf "$out"; mkdir -p "$out/profiles"
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -o r5 -- python bench.py --arch "$arch" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
    > "$out/stats_bench.json" 2> "$out/stats.err"
  python scripts/rocprof_to_profiles.py --round 5 --steps 3 --stats-dir "$out/stats" --tag "_${arch}" \
    --command "python bench.py --arch $arch --steps 1 --warmup 1 --no-cpu-baseline --no-secondary" > "$out/summary.txt" 2>&1
  find "$out" -name '*kernel_trace.csv' -delete
  cp profiles/r5_kernel_stats_${arch}.csv "$out/profiles/" 2>/dev/null
  tail -16 "$out/summary.txt" | cut -c1-200
done
