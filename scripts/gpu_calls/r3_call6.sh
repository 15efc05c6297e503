#!/bin/bash
# Round-3 sixth GPU call: half-tile 32x32x16 attention forward: parity, op A/B, C2 step A/B.
set -u
out=gpurun_out/r3c6
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest timeout 600 python -m pytest tests/test_kernels_gpu.py -q -rxXsf -p no:cacheprovider -k "attention"
grep -h "FAILED\|passed\|failed" "$out/01_pytest.log" | tail -8
run 02_attn_bench timeout 600 python scripts/attn_bench.py
cat "$out/02_attn_bench.log" | cut -c1-400
run 04_knob_ab timeout 900 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "old_attn:26=1;qf1:27=1;r2_attn:26=1+27=1"
tail -6 "$out/04_knob_ab.log" | cut -c1-300
