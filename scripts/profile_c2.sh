#!/bin/bash
# One gpurun call that produces every measured artefact of a round for the headline workload C2 (about 6 GPU-minutes):
#   gpurun --timeout 1200 -- 'bash scripts/profile_c2.sh 2'          (argument: round number; optional 2nd: FDMI_TUNE value)
# then, back in the authoring container (gpurun_out/ is merged back):
#   python scripts/rocprof_to_profiles.py --round 2 --steps 4 --stats-dir gpurun_out/prof/stats \
#          --fetch-dir gpurun_out/prof/pmc_fetch --write-dir gpurun_out/prof/pmc_write
#   cp gpurun_out/prof/bench.json profiles/r2_bench.json      and point bench.py's traffic file at profiles/r2_traffic.json
# Counter passes run with --kernel-trace only (gpurun refuses --pmc together with the hip/hsa/sys trace domains).
set -u
round=${1:-2}
[ -n "${2:-}" ] && export FDMI_TUNE=$2
out=gpurun_out/prof
mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
echo "== bench (defaults: the judged line)"
python bench.py > "$out/bench.json" 2> "$out/bench.err"; tail -c 600 "$out/bench.json"
echo "== rocprofv3 --kernel-trace --stats (4 steps: 1 warm-up + 2 timed + the profiled leg)"
rocprofv3 --kernel-trace --stats -d "$out/stats" -o r${round} -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/stats_bench.json" 2> "$out/stats.err"
echo "== rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 3 steps each)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/pmc_fetch" -o r${round} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/pmc_fetch_bench.json" 2> "$out/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/pmc_write" -o r${round} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/pmc_write_bench.json" 2> "$out/pmc_write.err"
# the per-dispatch traces are large: keep only what rocprof_to_profiles.py reads (gpurun_out/ merges back <= 64 MiB)
find "$out" -name '*kernel_trace.csv' -delete
python scripts/rocprof_to_profiles.py --round "$round" --steps 4 --stats-dir "$out/stats" --fetch-dir "$out/pmc_fetch" \
  --write-dir "$out/pmc_write" > "$out/summary.txt" 2>&1
mkdir -p "$out/profiles" && cp profiles/r${round}_* "$out/profiles/" 2>/dev/null
du -sh "$out"; tail -20 "$out/summary.txt"
