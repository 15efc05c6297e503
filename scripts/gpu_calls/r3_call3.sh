#!/bin/bash
# Round-3 third GPU call: the 32x32x16 attention forward (parity against fp32 torch and against the 16x16x32 kernel it replaces,
# op-level A/B, C2 step A/B), the tests whose bars were re-measured after call 2, the CLIP text encoder tests.
set -u
out=gpurun_out/r3c3
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest_attn timeout 900 python -m pytest tests/test_kernels_gpu.py -q -rxXsf -p no:cacheprovider -k "attention"
grep -h "FAILED\|passed\|failed\|Error" "$out/01_pytest_attn.log" | tail -20
run 02_attn_bench timeout 600 python scripts/attn_bench.py
cat "$out/02_attn_bench.log" | cut -c1-300
run 03_pytest_fixed timeout 1500 python -m pytest tests/test_nets_gpu.py tests/test_clip_gpu.py tests/test_multiproc_gpu.py tests/test_flash_gpu.py -q -rxXsf -p no:cacheprovider
grep -h "FAILED\|passed\|failed" "$out/03_pytest_fixed.log" | tail -20
run 04_knob_ab timeout 900 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "old_attn:26=1;qf2:27=1"
tail -25 "$out/04_knob_ab.log" | cut -c1-300
run 05_pytest_unet timeout 1500 python -m pytest tests/test_unet_gpu.py tests/test_fullsize_parity_gpu.py -q -rxXsf -p no:cacheprovider
grep -h "FAILED\|passed\|failed" "$out/05_pytest_unet.log" | tail -20
cp gpurun_out/*.txt "$out/" 2>/dev/null
