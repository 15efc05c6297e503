#!/bin/bash
set -u
out=gpurun_out/r2c17
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q -rxXsf -p no:cacheprovider -k "attention or unet"
run 02_attn_bench_3 timeout 300 python scripts/attn_bench.py
run 03_attn_bench_2 env FDMI_TUNE=27=1 timeout 300 python scripts/attn_bench.py
grep -h "attn fwd" "$out/02_attn_bench_3.log" | sed 's/^/3 stages: /'; grep -h "attn fwd" "$out/03_attn_bench_2.log" | sed 's/^/2 stages: /'
run 04_knob_ab timeout 600 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base,attn_two_stages
