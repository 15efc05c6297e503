#!/bin/bash
# K-loop micro-benchmark, 32-deep counted ring variants only (scripts/ubench/gemm_kloop REPS FILL 1)
set -u
out=gpurun_out/r6kloop
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
timeout 600 scripts/ubench/gemm_kloop ${1:-10} 0 1 > "$out/kloop32_${2:-a}.txt" 2>&1; echo "exit $?"
cat "$out/kloop32_${2:-a}.txt"
