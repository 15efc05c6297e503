#!/bin/bash
set -u
out=gpurun_out/r2c7
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_flash_gpu.py tests/test_fullsize_gpu.py tests/test_unet_gpu.py -q -rxXsf -p no:cacheprovider
run 02_knob_ab timeout 600 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base,no_side_stream
cat "$out/02_knob_ab.log" | tail -4
