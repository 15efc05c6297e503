#!/bin/bash
set -u
out=gpurun_out/r2c5
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-240))"; }
run 01_pytest timeout 1500 python -m pytest tests -m gpu -q -rxXsf -p no:cacheprovider --deselect tests/test_fp32_gate_gpu.py --deselect tests/test_multiproc_gpu.py
run 02_knob_ab timeout 600 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base,unfused_qkv --legs
cat "$out/02_knob_ab.log" | tail -8
