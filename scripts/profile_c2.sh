#!/bin/bash
# One gpurun call that produces every measured artefact of a round for the headline workload C2 (about 7 GPU-minutes):
#   gpurun --timeout 1500 -- 'bash scripts/profile_c2.sh 2'          (argument: round number; optional 2nd: FDMI_TUNE value)
# The summaries land in gpurun_out/prof/profiles/ (already converted); back in the authoring container:
#   cp gpurun_out/prof/profiles/* profiles/ && cp gpurun_out/prof/bench.json profiles/r2_bench.json
# Counter passes run with --kernel-trace only (gpurun refuses --pmc together with the hip/hsa/sys trace domains); every rocprofv3
# pass sits under its own `timeout` (round 3: a counter pass aborted and then hung in its signal handler for 20 minutes).
set -u
round=${1:-2}
[ -n "${2:-}" ] && export FDMI_TUNE=$2
out=gpurun_out/prof
mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
echo "== bench (defaults: the judged line)"
python bench.py > "$out/bench.json" 2> "$out/bench.err"; tail -c 900 "$out/bench.json"
echo "== rocprofv3 --kernel-trace --stats (4 steps: 1 warm-up + 2 timed + the profiled leg)"
timeout -s KILL 420 rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -o r${round} -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/stats_bench.json" 2> "$out/stats.err"
echo "== rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 3 steps each)"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/pmc_fetch" -o r${round} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/pmc_fetch_bench.json" 2> "$out/pmc_fetch.err"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d "$out/pmc_write" -o r${round} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/pmc_write_bench.json" 2> "$out/pmc_write.err"
echo "== rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum (own pass: L2 hit rate per kernel)"
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -f csv -d "$out/pmc_tcc" -o r${round} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/pmc_tcc_bench.json" 2> "$out/pmc_tcc.err"
echo "== rocprofv3 --pmc GRBM_GUI_ACTIVE (own pass: effective clock per kernel)"
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$out/pmc_grbm" -o r${round} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/pmc_grbm_bench.json" 2> "$out/pmc_grbm.err"
python scripts/rocprof_to_profiles.py --round "$round" --steps 4 --stats-dir "$out/stats" --fetch-dir "$out/pmc_fetch" \
  --write-dir "$out/pmc_write" --tcc-dir "$out/pmc_tcc" --grbm-dir "$out/pmc_grbm" > "$out/summary.txt" 2>&1
# the per-dispatch traces are large: keep only the summaries (gpurun_out/ merges back <= 64 MiB)
find "$out" -name '*kernel_trace.csv' -delete; find "$out" -name '*counter_collection.csv' -delete
mkdir -p "$out/profiles" && cp profiles/r${round}_kernel_stats.csv profiles/r${round}_pmc_hbm_traffic.csv profiles/r${round}_traffic.json profiles/r${round}_l2_hit_rate.csv profiles/r${round}_clock.csv "$out/profiles/" 2>/dev/null
echo "== register-only MFMA rate under full-chip load (scripts/ubench/mfma_rate.hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate scripts/ubench/mfma_rate.hip && /tmp/mfma_rate > "$out/profiles/r${round}_mfma_rate.txt" 2>&1
cat "$out/profiles/r${round}_mfma_rate.txt"
du -sh "$out"; tail -20 "$out/summary.txt"
