#!/bin/bash
# Round-5 eighth GPU call: the WHOLE GPU suite as the driver runs it (pytest.ini: xdist, bounded host threads), with durations; smoke().
set -u
out=gpurun_out/r5c8
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/*_parity.txt gpurun_out/batch_invariance.txt gpurun_out/test_durations.txt gpurun_out/kernel_diag.txt
SECONDS=0
timeout 1500 python -m pytest tests/ -x -q -m gpu > "$out/01_pytest_all.log" 2>&1
echo "   pytest exit $? ; suite wall: $SECONDS s"
tail -8 "$out/01_pytest_all.log" | cut -c1-400
sort -rn gpurun_out/test_durations.txt | head -14
awk '{w[$4]+=$1; n[$4]+=1} END {for (k in w) print k, n[k], w[k]}' gpurun_out/test_durations.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')" > "$out/02_smoke.log" 2>&1; tail -3 "$out/02_smoke.log" | cut -c1-300
