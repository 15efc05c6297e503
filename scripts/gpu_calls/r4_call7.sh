#!/bin/bash
# Round-4 GPU call 7: the new parity gates -- C2 at B = 16, the full-width SDXL step (bf16 + fp32), per-fixture loss bars,
# the gradient-level two-rank check
set -u
out=gpurun_out/r4c7
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -k "own_batch or full_width_step" > "$out/pytest_fullsize.log" 2>&1; tail -15 "$out/pytest_fullsize.log" | cut -c1-300
timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_multiproc_gpu.py -x -q > "$out/pytest_flash_mp.log" 2>&1; tail -6 "$out/pytest_flash_mp.log" | cut -c1-300
tail -12 gpurun_out/fullsize_parity.txt | cut -c1-400
tail -3 gpurun_out/multiproc_parity.txt | cut -c1-400
