#!/bin/bash
# Round-3 ninth GPU call: planner-chosen split-K in the transformer denoisers' projections: tests + PixArt / SD3 legs (A/B by FDMI_DIT_SPLITK)
set -u
out=gpurun_out/r3c9
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 02_epi timeout 900 python -m pytest tests/test_kernels_gpu.py -q -rxXsf -p no:cacheprovider -k "gate_and_gelu"
grep -h "FAILED\|passed\|failed" "$out/02_epi.log" | tail -8
run 03_dit timeout 1500 python -m pytest tests/test_zz_dit_gpu.py tests/test_sd3_mmdit_step_gpu.py tests/test_pixart_step_gpu.py -q -rxXsf -p no:cacheprovider -k "dit or mmdit or pixart or sd3"
grep -h "FAILED\|passed\|failed" "$out/03_dit.log" | tail -12
for arch in pixart sd3; do
  for sk in 0 1; do
    FDMI_DIT_SPLITK=$sk timeout 600 python bench.py --arch $arch --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_${arch}_sk$sk.json" 2> "$out/bench_${arch}_sk$sk.err"
    python - <<PY
import json
l=[x for x in open("$out/bench_${arch}_sk$sk.json") if x.startswith('{')][-1]
d=json.loads(l); print("$arch splitk=$sk", round(d['ms_per_step'],1), 'ms', round(d['value'],3), d['unit'])
PY
  done
done
