#!/bin/bash
set -u
out=gpurun_out/r2c12
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q -rxXsf -p no:cacheprovider -k "groupnorm or splitk or unet or gemm3 or gemm4"
run 02_knob_ab timeout 600 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "oldfin:22=1;oldgn:23=1"
