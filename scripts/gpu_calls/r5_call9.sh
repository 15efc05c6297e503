#!/bin/bash
# Round-5 ninth GPU call: does the matrix pipe run beside the VALU?  (scripts/ubench/mfma_valu_overlap.hip)
set -u
out=gpurun_out/r5c9
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_valu_overlap scripts/ubench/mfma_valu_overlap.hip && /tmp/mfma_valu_overlap > "$out/mfma_valu_overlap.txt" 2>&1
cat "$out/mfma_valu_overlap.txt"
/tmp/mfma_valu_overlap >> "$out/mfma_valu_overlap.txt" 2>&1
tail -4 "$out/mfma_valu_overlap.txt"
