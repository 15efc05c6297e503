#!/bin/bash
# Round 4, call 15: teacher loop / KV join / LayerNorm loads: the DiT tests, C4 / C5 bench lines, kernel stats of both with the plans.
mkdir -p gpurun_out/c15
timeout 900 python -m pytest tests/test_zz_dit_gpu.py tests/test_pixart_step_gpu.py tests/test_sd3_mmdit_step_gpu.py tests/test_flash_sd3_gpu.py -q -k "dit or pixart or sd3 or mmdit or layernorm or plan" > gpurun_out/c15/dit.txt 2>&1; echo "dit rc=$?"
tail -12 gpurun_out/c15/dit.txt | cut -c1-700
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "layernorm or ln" > gpurun_out/c15/ln.txt 2>&1; echo "ln rc=$?"; tail -3 gpurun_out/c15/ln.txt
for arch in pixart sd3; do
  timeout 600 python bench.py --arch $arch --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c15/bench_${arch}.json 2> gpurun_out/c15/bench_${arch}.err
  echo "bench $arch rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c15/bench_${arch}.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "unit", "ms_per_step")}, d.get("roofline", {}).get("whole_step"))
except Exception as e:
    print("no line", e); print(open("gpurun_out/c15/bench_${arch}.err").read()[-1500:])
PY
done
cd /tmp && export TMPDIR=/tmp
for arch in pixart sd3; do
  rm -rf /tmp/prof_$arch
  (cd $GRAFT_REPO_ROOT && FDMI_BENCH_NO_PROFILE_LEG=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$arch -o st -- python bench.py --arch $arch --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/prof_$arch.log 2>&1)
  echo "rocprof $arch rc=$?"
  f=$(find /tmp/prof_$arch -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/c15/kernel_stats_${arch}_raw.csv && head -12 "$f" | cut -c1-160
done
