#!/bin/bash
# Round-5 second GPU call: the conv kernel's own timing ablations (A from L2, no epilogue), the tuned row split of the streaming
# weight-gradient kernel, where the GPU idles during a step (kernel trace -> scripts/trace_gaps.py), the C4 / C5 lines with the new
# weight-gradient kernel.
set -u
out=gpurun_out/r5c2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_conv_ablation timeout 400 python scripts/r5_exp1.py conv 20
cat "$out/01_conv_ablation.log" | cut -c1-500
run 02_wgrad_rates timeout 300 python scripts/wgrad_rates.py 46=1,0
cat "$out/02_wgrad_rates.log" | cut -c1-400
run 03_bench timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
run 04_trace timeout 600 rocprofv3 --kernel-trace -f csv -d "$out/tg" -o tg -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary
run 05_gaps python scripts/trace_gaps.py "$out/tg" 0.3
cat "$out/05_gaps.log"
find "$out/tg" -name '*kernel_trace.csv' -delete
for arch in pixart sd3; do
  run 06_bench_$arch timeout 900 python bench.py --arch $arch --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5c2/0[36]_bench*.log")):
    try:
        d=json.loads([x for x in open(f) if x.startswith('{')][-1]); print(f, round(d['ms_per_step'],1),'ms', round(d['value'],2), d['unit'])
    except Exception as e: print(f,'failed',e)
PY
