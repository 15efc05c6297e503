#!/bin/bash
set -u
out=gpurun_out/r6det2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/flash_parity.txt
timeout 1500 python -m pytest tests/test_flash_gpu.py -q -rxXsf -k "step_matches and deterministic" -p no:cacheprovider > "$out/pytest.log" 2>&1
echo "exit $?"
grep "^FAILED\|^E  \|passed\|failed" "$out/pytest.log" | cut -c1-300 | tail -30
grep "deterministic" gpurun_out/flash_parity.txt | cut -c1-330
