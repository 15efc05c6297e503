#!/bin/bash
# Round-4 GPU call 6: the 128 x 320 deferred-epilogue kernel (gemm6.hip) against the 256 x 320 kernel, per shape (forced tiles)
set -u
out=gpurun_out/r4c6
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout -s KILL 240 python scripts/rowbench.py g6 30 > "$out/rowbench_g6.txt" 2>&1
echo "rc=$?"; cut -c1-400 "$out/rowbench_g6.txt"
