#!/bin/bash
# Round-4 GPU call (re-used for every epilogue iteration): the lean / line-wide epilogue (gemm_tile.h) -- per-shape timings against
# general epilogue (knob 40 = 64) with bit-equality of the outputs, the kernel / UNet / step parity tests, the step A/B
set -u
out=gpurun_out/r4c2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 400 python scripts/rowbench.py dev 30 > "$out/rowbench_dev.txt" 2>&1
cut -c1-700 "$out/rowbench_dev.txt"
echo "== kernel + UNet + step parity"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_flash_gpu.py -x -q > "$out/pytest.log" 2>&1; tail -4 "$out/pytest.log"
echo "== knob A/B"
timeout 700 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "general_epilogue:40=64;singles:40=256" > "$out/knob_ab.log" 2>&1
grep -E "^(base|general|variant|leg|\{)" "$out/knob_ab.log" | cut -c1-400
tail -5 "$out/knob_ab.log" | cut -c1-600
