#!/bin/bash
# Round-6: LDS bank conflicts per kernel over whole steps of C2 / C4 / C5 (one --pmc pass each, --kernel-trace only)
set -u
out=gpurun_out/r6s3ldsscan
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for arch in sd15 pixart sd3; do
  timeout -s KILL 420 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -f csv -d "$out/$arch" -o a -- \
    python bench.py --arch $arch --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > "$out/$arch.log" 2>&1
  python scripts/pmc_by_kernel.py "$out/$arch" 45 > "$out/lds_by_kernel_$arch.txt" 2>&1
  find "$out/$arch" -name '*.csv' -delete
  echo "== $arch"; cut -c1-200 "$out/lds_by_kernel_$arch.txt" | head -50
done
