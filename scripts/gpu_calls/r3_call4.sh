#!/bin/bash
# Round-3 fourth GPU call: counters of the two attention-forward kernels; the re-barred tests.
set -u
out=gpurun_out/r3c4
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_attn_pmc timeout 900 bash scripts/attn_pmc.sh "$out/attn_pmc"
cat "$out/attn_pmc/table.txt"
run 02_pytest timeout 600 python -m pytest tests/test_kernels_gpu.py -q -rxXsf -p no:cacheprovider -k "attention"
grep -h "FAILED\|passed\|failed" "$out/02_pytest.log" | tail -8
