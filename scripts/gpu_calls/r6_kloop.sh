#!/bin/bash
# K-loop structure micro-benchmark (scripts/ubench/gemm_kloop.hip; built in this container, the binary travels)
# usage: r6_kloop.sh REPS TAG "FILLS"   (fill 0: uniform random, 1: zeros, 2: constants)
set -u
out=gpurun_out/r6kloop
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
tag=${2:-a}
for fill in ${3:-0}; do
  timeout 600 scripts/ubench/gemm_kloop ${1:-10} $fill > "$out/kloop_${tag}_fill$fill.txt" 2>&1
  echo "exit $?"
  cat "$out/kloop_${tag}_fill$fill.txt"
done
