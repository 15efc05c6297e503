#!/bin/bash
# Round-5 fifth GPU call: the fp32 residual stream of the transformer denoisers -- every DiT / MMDiT test, the anchored four-step
# fixtures, batch invariance, and its cost on the C4 / C5 lines (knob 49 = 1: the bf16 stream of round 4); the re-run of the unet test
# that failed with the in-launch split-K reduction on.
set -u
out=gpurun_out/r5c5
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
rm -f gpurun_out/fullsize_parity.txt gpurun_out/batch_invariance.txt gpurun_out/test_durations.txt
run 01_pytest_dit timeout 1700 python -m pytest tests/test_zz_dit_gpu.py tests/test_pixart_step_gpu.py tests/test_sd3_mmdit_step_gpu.py tests/test_flash_sd3_gpu.py tests/test_step4_parity_gpu.py tests/test_batch_invariance_gpu.py tests/test_unet_gpu.py tests/test_fullsize_parity_gpu.py -q -rxXsf -k "not sdxl-bf16 or batch"
tail -30 "$out/01_pytest_dit.log" | cut -c1-500
cat gpurun_out/fullsize_parity.txt gpurun_out/batch_invariance.txt 2>/dev/null | grep -v "gradient tensors" | cut -c1-420
for arch in pixart sd3; do
  run 02_bench_${arch}_f32stream timeout 900 python bench.py --arch $arch --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
  run 03_bench_${arch}_bf16stream env FDMI_TUNE=49=1 timeout 900 python bench.py --arch $arch --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5c5/0[23]_bench*.log")):
    try:
        d=json.loads([x for x in open(f) if x.startswith('{')][-1]); print(f, round(d['ms_per_step'],1),'ms', round(d['value'],2), d['unit'])
    except Exception as e: print(f,'failed',e)
PY
sort -rn gpurun_out/test_durations.txt | head -25
