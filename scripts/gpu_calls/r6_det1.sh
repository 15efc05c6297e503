#!/bin/bash
# Round-6: first run of the deterministic mode (knob 50) and the tests that use it
set -u
out=gpurun_out/r6det1
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_deterministic_gpu.py tests/test_flash_gpu.py tests/test_unet_gpu.py tests/test_fullsize_parity_gpu.py -q -x -rxXsf -k "deterministic or c2_shaped or tiny_backward or folded or step_matches" -p no:cacheprovider > "$out/pytest.log" 2>&1
echo "exit $?"
tail -40 "$out/pytest.log" | cut -c1-400
cat gpurun_out/deterministic.txt 2>/dev/null | cut -c1-300
grep "deterministic" gpurun_out/flash_parity.txt gpurun_out/unet_parity.txt gpurun_out/fullsize_parity.txt 2>/dev/null | cut -c1-400 | tail -40
