#!/bin/bash
# Round 4, call 13: every transformer-denoiser test through the plans + C4 / C5 bench lines, plan vs op-by-op on the same box.
mkdir -p gpurun_out/c13
timeout 900 python -m pytest tests/test_zz_dit_gpu.py tests/test_pixart_step_gpu.py tests/test_sd3_mmdit_step_gpu.py tests/test_flash_sd3_gpu.py -q -k "dit or pixart or sd3 or mmdit" > gpurun_out/c13/dit.txt 2>&1; echo "dit rc=$?"
tail -15 gpurun_out/c13/dit.txt | cut -c1-600
timeout 1200 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_fullsize_gpu.py -q -k "pixart or sd3 or dit or mmdit" > gpurun_out/c13/full.txt 2>&1; echo "full rc=$?"
tail -15 gpurun_out/c13/full.txt | cut -c1-600
for arch in pixart sd3; do
  for plan in 1 0; do
    FDMI_DIT_PLAN=$plan timeout 600 python bench.py --arch $arch --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/c13/bench_${arch}_plan$plan.json 2> gpurun_out/c13/bench_${arch}_plan$plan.err
    echo "bench $arch plan=$plan rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c13/bench_${arch}_plan$plan.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "unit", "ms_per_step")}, d.get("roofline", {}).get("whole_step"))
except Exception as e:
    print("no line", e); print(open("gpurun_out/c13/bench_${arch}_plan$plan.err").read()[-1500:])
PY
  done
done
rocm-smi --showmeminfo vram 2>/dev/null | head -5
