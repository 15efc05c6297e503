#!/bin/bash
set -u
out=gpurun_out/r2c19
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q -rxXsf -p no:cacheprovider -k "attention or unet"
grep -h "FAILED" "$out/01_pytest.log"
run 02_attn_bench timeout 300 python scripts/attn_bench.py
grep -h "attn fwd" "$out/02_attn_bench.log"
run 03_knob_ab timeout 600 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base
