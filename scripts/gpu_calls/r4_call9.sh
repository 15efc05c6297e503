#!/bin/bash
# Round-4 GPU call 9: the 2^-(polynomial) GELU (GEGLU timings + parity tests), the PixArt full-width step, the nets tests
set -u
out=gpurun_out/r4c9
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 400 python scripts/rowbench.py dev 30 > "$out/rowbench_dev.txt" 2>&1
grep GEGLU "$out/rowbench_dev.txt" | cut -c1-300
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_flash_gpu.py -x -q > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log" | cut -c1-300
timeout 900 python -m pytest tests/test_nets_gpu.py -x -q > "$out/pytest_nets.log" 2>&1; tail -3 "$out/pytest_nets.log" | cut -c1-300
timeout 2400 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -k "full_width_step and pixart" > "$out/pytest_pixart.log" 2>&1; tail -5 "$out/pytest_pixart.log" | cut -c1-300
timeout 500 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base > "$out/knob_ab.log" 2>&1
grep -E "^(base|variant)" "$out/knob_ab.log" | cut -c1-200
grep "step_pixart" gpurun_out/fullsize_parity.txt | tail -4 | cut -c1-400
