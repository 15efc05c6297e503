#!/bin/bash
# Round-6: gemm5 + the 32-deep ring micro-benchmark after the lane-group-aware swizzle of the 64-byte rows
set -u
out=gpurun_out/r6g5
mkdir -p "$out" gpurun_out/r6kloop
cd "$(dirname "$0")/../.." || exit 1
timeout 600 scripts/ubench/gemm_kloop 10 0 1 > gpurun_out/r6kloop/kloop32_b.txt 2>&1; echo "exit $?"
cat gpurun_out/r6kloop/kloop32_b.txt
timeout 600 python scripts/rowbench.py 20 > "$out/rowbench_b.txt" 2>&1; echo "exit $?"
grep "^M=" "$out/rowbench_b.txt" | cut -c1-200
