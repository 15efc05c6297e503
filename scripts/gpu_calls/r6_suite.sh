#!/bin/bash
# Round-6: the whole GPU suite as the driver runs it, then the default bench line (as the driver runs it: no flags)
set -u
out=gpurun_out/r6suite${1:-}
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/*_parity.txt gpurun_out/fp32_gate.txt gpurun_out/batch_invariance.txt gpurun_out/test_durations.txt gpurun_out/kernel_diag.txt
SECONDS=0
timeout 1500 python -m pytest tests/ -x -q -m gpu > "$out/01_pytest.log" 2>&1
echo "   pytest exit $? ; suite wall: $SECONDS s"; tail -6 "$out/01_pytest.log" | cut -c1-300
if [ "${2:-bench}" = "bench" ]; then
SECONDS=0
timeout 900 python bench.py > "$out/02_bench.log" 2> "$out/02_bench.err"
echo "   bench exit $? ; wall: $SECONDS s"; tail -1 "$out/02_bench.log" | cut -c1-400
fi
