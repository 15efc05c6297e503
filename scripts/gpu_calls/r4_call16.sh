#!/bin/bash
# Round 4, call 16: rocprofv3 kernel stats of the C4 / C5 steps through the plans.
mkdir -p gpurun_out/c16
cd /tmp && export TMPDIR=/tmp
for arch in sd3 pixart; do
  rm -rf /tmp/prof_$arch
  cd $GRAFT_REPO_ROOT
  FDMI_BENCH_NO_PROFILE_LEG=1 timeout -s KILL 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$arch -o st -- python bench.py --arch $arch --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/c16/prof_$arch.log 2>&1
  echo "rocprof $arch rc=$?"; tail -3 gpurun_out/c16/prof_$arch.log | cut -c1-300
  f=$(find /tmp/prof_$arch -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/c16/kernel_stats_${arch}_raw.csv; head -14 "$f" | cut -c1-170; fi
done
