#!/bin/bash
# Round-3 eleventh GPU call: the 80-wide instantiations of the 32x32x16 attention forward (d = 72 / 80): parity, op A/B, PixArt leg, C2 A/B
set -u
out=gpurun_out/r3c11
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest timeout 600 python -m pytest tests/test_kernels_gpu.py -q -rxXsf -p no:cacheprovider -k "attention"
grep -h "FAILED\|passed\|failed" "$out/01_pytest.log" | tail -12
run 02_attn_bench timeout 600 python scripts/attn_bench.py
cat "$out/02_attn_bench.log" | cut -c1-400
for k in 0 1; do
  FDMI_TUNE="26=$k" timeout 600 python bench.py --arch pixart --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_pixart_26_$k.json" 2> "$out/bench_pixart_26_$k.err"
  python - <<PY
import json
l=[x for x in open("$out/bench_pixart_26_$k.json") if x.startswith('{')][-1]
d=json.loads(l); print("pixart knob26=$k", round(d['ms_per_step'],1), 'ms', round(d['value'],3), d['unit'], {k:(round(v['ms_per_step'],1), round(v['tflops'])) for k,v in d['roofline'].get('families',{}).items()})
PY
done
run 04_knob_ab timeout 900 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "old_attn:26=1"
tail -4 "$out/04_knob_ab.log" | cut -c1-300
