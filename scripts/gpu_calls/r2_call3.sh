#!/bin/bash
# Round-2 third GPU call: the whole GPU suite on the new defaults + the fp32 parity gate, then the C2 line of the new defaults.
set -u
out=gpurun_out/r2c3
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
echo "== pytest fp32 gate"
timeout 1200 python -m pytest tests/test_fp32_gate_gpu.py -q -rxXsf -p no:cacheprovider > "$out/pytest_fp32.txt" 2>&1
echo "   exit $?"; tail -4 "$out/pytest_fp32.txt"
echo "== pytest -m gpu (rest)"
FDMI_RUN_DEV_KNOBS=1 timeout 1200 python -m pytest tests -m gpu -q -rxXsf -p no:cacheprovider --deselect tests/test_fp32_gate_gpu.py > "$out/pytest.txt" 2>&1
echo "   exit $?"; tail -4 "$out/pytest.txt"
echo "== C2 on the new defaults"
timeout 600 python scripts/knob_ab.py --rounds 2 --steps 3 --variants base --legs > "$out/knob_ab.txt" 2>&1
tail -6 "$out/knob_ab.txt"
