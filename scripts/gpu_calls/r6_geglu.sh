#!/bin/bash
# Round-6: packed-fp32 GEGLU epilogue -- bit-identity against the general epilogue + timing (scripts/rowbench.py dev), GEMM kernel tests
set -u
out=gpurun_out/r6geglu
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
timeout 600 python scripts/rowbench.py dev 20 > "$out/rowbench_dev.txt" 2>&1; echo "exit $?"
grep "GEGLU\|MISMATCH" "$out/rowbench_dev.txt" | cut -c1-220
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or geglu or conv" 2>&1 | tail -4
