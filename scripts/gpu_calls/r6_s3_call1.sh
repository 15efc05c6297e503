#!/bin/bash
# Round-6, third session, first call: C2 line of the tree on this box; library-level A/B of s_setprio in the ring kernels' K loops
# (build/variants/nosetprio: -DFDMI_NO_SETPRIO on gemm3/4/5) on the step; kernel trace of the timed steps -> idle-gap attribution
# after the ops.upload fix (scripts/trace_gaps2.py).
set -u
out=gpurun_out/r6s3c1
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
b() { timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],2))"; }
for r in 1 2 3; do
  unset FDMI_LIB; echo "intree_$r $(b)"
  export FDMI_LIB="$PWD/build/variants/nosetprio/libfdmi.so"; echo "nosetprio_$r $(b)"
done | tee "$out/setprio_ab.txt"
unset FDMI_LIB
timeout -s KILL 420 rocprofv3 --kernel-trace -f csv -d "$out/tg" -o tg -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > "$out/tg_bench.json" 2> "$out/tg.err"
python scripts/trace_gaps2.py "$out/tg" 2 3 > "$out/trace_gaps_after.txt" 2>&1
find "$out" -name '*kernel_trace.csv' -delete
head -70 "$out/trace_gaps_after.txt"
