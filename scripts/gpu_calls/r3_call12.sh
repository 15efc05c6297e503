#!/bin/bash
# Round-3 twelfth GPU call: end-to-end effect of the 32x32x16 attention backward kernels (C2 in-process A/B, PixArt / SD3 / SDXL legs A/B by FDMI_TUNE)
set -u
out=gpurun_out/r3c12
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 900 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "old_bwd:35=1+36=1;r2_attn:26=1+27=1+35=1+36=1+34=1" > "$out/knob_ab.log" 2>&1
grep -E "^(base|old_bwd|r2_attn) " "$out/knob_ab.log" | cut -c1-200
for arch in pixart sd3 sdxl; do
  for tune in "" "35=1,36=1"; do
    FDMI_TUNE="$tune" timeout 600 python bench.py --arch $arch --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_${arch}_${tune//[=,]/_}.json" 2> "$out/bench_${arch}_${tune//[=,]/_}.err"
    python - <<PY
import json
l=[x for x in open("$out/bench_${arch}_${tune//[=,]/_}.json") if x.startswith('{')][-1]
d=json.loads(l); print("$arch FDMI_TUNE='$tune'", round(d['ms_per_step'],1), 'ms', round(d['value'],3), d['unit'])
PY
  done
done
