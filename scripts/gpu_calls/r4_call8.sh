#!/bin/bash
# Round-4 GPU call 8: lean epilogue of the GroupNorm-sum kernels (A/B against the general epilogue), tiled VAE decode, parity tests
set -u
out=gpurun_out/r4c8
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_flash_gpu.py -x -q > "$out/pytest.log" 2>&1; tail -4 "$out/pytest.log" | cut -c1-300
timeout 900 python -m pytest tests/test_nets_gpu.py -x -q -k "tiled or vae" > "$out/pytest_nets.log" 2>&1; tail -4 "$out/pytest_nets.log" | cut -c1-300
timeout 700 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "general_epilogue:40=64;singles:40=256" > "$out/knob_ab.log" 2>&1
grep -E "^(base|general|singles|variant)" "$out/knob_ab.log" | cut -c1-200
tail -4 gpurun_out/nets_parity.txt | cut -c1-300
