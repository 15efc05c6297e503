#!/bin/bash
# K-loop micro-benchmark: fill-traffic ablations of the product structure (scripts/ubench/gemm_kloop REPS 0 2)
set -u
out=gpurun_out/r6kloop
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
timeout 600 scripts/ubench/gemm_kloop ${1:-10} 0 ${2:-2} > "$out/kloop_fill${2:-2}.txt" 2>&1; echo "exit $?"
cat "$out/kloop_fill${2:-2}.txt"
