#!/bin/bash
# Round-5 seventh GPU call: xdist with bounded host threads on the files that thrashed in call 6, the anchored C2 bars, SDXL batch invariance.
set -u
out=gpurun_out/r5c7
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
python -c "import os; print('cpu_count', os.cpu_count())"
rm -f gpurun_out/test_durations.txt gpurun_out/fullsize_parity.txt gpurun_out/batch_invariance.txt
SECONDS=0
run 01_pytest_subset timeout 1200 python -m pytest tests/test_flash_gpu.py tests/test_nets_gpu.py tests/test_fp32_gate_gpu.py tests/test_batch_invariance_gpu.py tests/test_fullsize_parity_gpu.py -q -m gpu -rxXsf -k "not fp32] and not full_size_forward and not step_pixart and not step_sd3 or test_fp32_gate or test_nets"
echo "   subset wall: $SECONDS s"
tail -8 "$out/01_pytest_subset.log" | cut -c1-400
sort -rn gpurun_out/test_durations.txt | head -12
awk '{w[$4]+=$1; n[$4]+=1} END {for (k in w) print k, n[k], w[k]}' gpurun_out/test_durations.txt
cat gpurun_out/fullsize_parity.txt gpurun_out/batch_invariance.txt 2>/dev/null | grep -v "gradient tensors" | cut -c1-420
