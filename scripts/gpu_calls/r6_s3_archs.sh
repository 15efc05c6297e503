#!/bin/bash
# Round-6 closing lines of the other single-GPU configurations (C3 SDXL, C4 PixArt, C5 SD3) on ONE box, 5 timed steps each.
set -u
out=gpurun_out/r6s3archs
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for arch in sdxl pixart sd3; do
  timeout 900 python bench.py --arch $arch --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2> "$out/$arch.err" | grep '^{"metric"' > "$out/r6_bench_final_$arch.json"
  python -c "
import json,sys
d=json.loads(open('$out/r6_bench_final_$arch.json').readline())
print('$arch', round(d['ms_per_step'],1), 'ms', round(d['value'],2), 'images/s', round(d['roofline']['whole_step']['frac_of_peak'],3))
"
done
