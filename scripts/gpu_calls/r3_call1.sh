#!/bin/bash
# Round-3 first GPU call: the whole GPU suite on the round's first build (128x160 two-blocks-per-CU geometry, two-segment A
# operand: virtual concat + folded LoRA, new parity fixtures), the short-K row-GEMM micro-benchmark, and the in-process A/B of the
# three new switches on C2.
set -u
out=gpurun_out/r3c1
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest_new timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_zz_dit_gpu.py -q -rxXsf -p no:cacheprovider -x -k "gemm5 or two_segment or concatenation or geglu or folded or virtual or epilogue_groupnorm or gemm4 or gemm3"
grep -h "FAILED\|passed\|failed" "$out/01_pytest_new.log" | tail -5
run 02_rowbench timeout 300 python scripts/rowbench.py 30
cat "$out/02_rowbench.log"
run 03_knob_ab timeout 900 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "no_g5:28=1;no_cat:30=1;no_fold:31=1;r2:28=1+30=1+31=1" --legs
grep -v "^KNOB_AB_JSON" "$out/03_knob_ab.log" | tail -14
run 04_pytest_all timeout 2400 python -m pytest tests -m gpu -q -rxXsf -p no:cacheprovider
grep -h "FAILED\|passed\|failed\|ERROR" "$out/04_pytest_all.log" | tail -30
cp gpurun_out/*.txt "$out/" 2>/dev/null
