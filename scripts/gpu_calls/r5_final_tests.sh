#!/bin/bash
# Round-5 last GPU call: the whole GPU suite as the driver runs it (after the last test edits), parity logs for profiles/.
set -u
out=gpurun_out/r5final3
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/*_parity.txt gpurun_out/fp32_gate.txt gpurun_out/batch_invariance.txt gpurun_out/test_durations.txt gpurun_out/kernel_diag.txt
SECONDS=0
timeout 1500 python -m pytest tests/ ${FDMI_SUITE_X--x} -q -m gpu > "$out/01_pytest.log" 2>&1
echo "   pytest exit $? ; suite wall: $SECONDS s"; tail -4 "$out/01_pytest.log" | cut -c1-300
