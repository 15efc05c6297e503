#!/bin/bash
set -u
out=gpurun_out/r6det3
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_deterministic_gpu.py tests/test_flash_gpu.py -q -rxXsf -k "transformer_step or (step_matches and deterministic)" -p no:cacheprovider > "$out/pytest.log" 2>&1
echo "exit $?"
grep "^FAILED\|^E  \|passed\|failed" "$out/pytest.log" | cut -c1-400 | tail -30
tail -5 gpurun_out/deterministic.txt
