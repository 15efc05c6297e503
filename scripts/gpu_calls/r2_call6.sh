#!/bin/bash
set -u
out=gpurun_out/r2c6
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-240))"; }
run 01_pytest timeout 900 python -m pytest tests/test_fp32_gate_gpu.py -q -rxXsf -p no:cacheprovider -k "transformer_denoiser or sd3_step"
for leg in teacher student; do
  echo "== rocprofv3 stats: $leg"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof_$leg" -o leg -- python scripts/leg_prof.py --leg $leg --iters 4 > "$out/prof_$leg.txt" 2>&1
  tail -1 "$out/prof_$leg.txt"
done
