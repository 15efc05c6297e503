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
rm -rf "$out"
mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
echo "== bench (defaults: the judged line)"
python bench.py > "$out/bench.json" 2> "$out/bench.err"; tail -c 900 "$out/bench.json"
echo "== rocprofv3 --kernel-trace --stats (4 steps: 1 warm-up + 2 timed + the profiled leg)"
timeout -s KILL 420 rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -o r${round} -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary \
  > "$out/stats_bench.json" 2> "$out/stats.err"
echo "== rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 3 steps each)"
# The FETCH_SIZE pass aborts now and then with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT ~15 s in (round 3: every time on one box;
# round 4: call 5 passed, call 10 failed, the WRITE / TCC / GRBM passes of the same call passed): retry, then fall back to the
# three raw counters FETCH_SIZE is derived from on gfx950 (counter_defs.yaml) in one pass
fetch_ok=0
for try in 1 2 3; do
  rm -rf "$out/pmc_fetch"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/pmc_fetch" -o r${round} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
    > "$out/pmc_fetch_bench.json" 2> "$out/pmc_fetch.err"
  if ls "$out"/pmc_fetch/*counter_collection.csv > /dev/null 2>&1 && ! grep -q INVALID_PACKET "$out/pmc_fetch.err"; then fetch_ok=1; echo "FETCH_SIZE pass: ok (try $try)"; break; fi
  echo "FETCH_SIZE pass: failed (try $try): $(grep -c INVALID_PACKET "$out/pmc_fetch.err") malformed-packet lines"
done
if [ "$fetch_ok" = 0 ]; then
  rm -rf "$out/pmc_fetch"
  timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace -f csv -d "$out/pmc_fetch" -o r${round} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary \
    > "$out/pmc_fetch_bench.json" 2> "$out/pmc_fetch.err"
  echo "raw TCC_EA0_RDREQ / RDREQ_32B / BUBBLE pass: $(ls "$out"/pmc_fetch/*counter_collection.csv 2>/dev/null | wc -l) counter file(s)"
fi
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
