#!/bin/bash
# Round-4 third GPU call: what bounds the epilogue of the short-K row GEMMs?  (1) HBM streaming rate by access pattern
# (scripts/ubench/stream_patterns.hip), (2) the lean epilogue with / without its stores (knob 40 = 128)
set -u
out=gpurun_out/r4c3
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result -o /tmp/stream_patterns scripts/ubench/stream_patterns.hip && timeout 120 /tmp/stream_patterns > "$out/stream_patterns.txt" 2>&1
cat "$out/stream_patterns.txt"
timeout 400 python scripts/rowbench.py dev 30 > "$out/rowbench_dev.txt" 2>&1
cut -c1-800 "$out/rowbench_dev.txt"
