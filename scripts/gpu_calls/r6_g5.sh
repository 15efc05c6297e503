#!/bin/bash
# Round-6: the 128x320 two-blocks-per-CU row kernel (gemm5.hip) against the 256x320 / 256x160 / 128x128 kernels (scripts/rowbench.py)
set -u
out=gpurun_out/r6g5
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
timeout 600 python scripts/rowbench.py ${1:-20} > "$out/rowbench_${2:-a}.txt" 2>&1; echo "exit $?"
grep "^M=" "$out/rowbench_${2:-a}.txt" | cut -c1-260
grep -v "^M=" "$out/rowbench_${2:-a}.txt" | tail -5
