#!/bin/bash
# Bench line + rocprofv3 kernel statistics of one of the other BASELINE.json configurations at its own per-GPU shape:
#   gpurun --timeout 1500 -- 'bash scripts/profile_arch.sh 2 sdxl'      (sdxl = C3, pixart = C4, sd3 = C5)
set -u
round=${1:-2}
arch=${2:-sdxl}
out=gpurun_out/prof_$arch
mkdir -p "$out/profiles"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
echo "== bench $arch"
python bench.py --arch "$arch" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$out/profiles/r${round}_bench_${arch}.json" 2> "$out/bench.err"
tail -c 600 "$out/profiles/r${round}_bench_${arch}.json"
echo "== rocprofv3 --kernel-trace --stats (3 steps: 1 warm-up + 1 timed + the profiled leg)"
rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -o r${round} -- python bench.py --arch "$arch" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/stats_bench.json" 2> "$out/stats.err"
python scripts/rocprof_to_profiles.py --round "$round" --steps 3 --stats-dir "$out/stats" --tag "_${arch}" \
  --command "python bench.py --arch $arch --steps 1 --warmup 1 --no-cpu-baseline --no-secondary" > "$out/summary.txt" 2>&1
find "$out" -name '*kernel_trace.csv' -delete
cp profiles/r${round}_kernel_stats_${arch}.csv "$out/profiles/" 2>/dev/null
tail -14 "$out/summary.txt"
