#!/bin/bash
# Round-6, third session, call 2: how the bench line depends on --steps (the deferred backward's fill / drain is inside the timed
# region once per run), and the library-level A/B of s_setprio in the ring kernels' K loops on the step.
set -u
out=gpurun_out/r6s3c2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
b() { timeout 600 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('steps', d['steps'], 'ms_per_step', round(d['ms_per_step'],2))"; }
for r in 1 2; do
  for k in 5 20 40; do echo "steps_sweep_$r $(b $k 5)"; done
done | tee "$out/steps_sweep.txt"
for r in 1 2 3; do
  unset FDMI_LIB; echo "intree_$r $(b 20 5)"
  export FDMI_LIB="$PWD/build/variants/nosetprio/libfdmi.so"; echo "nosetprio_$r $(b 20 5)"
done | tee "$out/setprio_ab.txt"
