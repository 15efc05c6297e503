#!/bin/bash
# Round-2 first GPU call: the truthful test run (no module-wide xfail, knob-gated tests ON), the in-process knob A/B on C2, and
# per-leg kernel statistics (student fwd+bwd / teacher 2B forward).
set -u
out=gpurun_out/r2c1
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
echo "== pytest -m gpu with FDMI_RUN_DEV_KNOBS=1"
FDMI_RUN_DEV_KNOBS=1 timeout 900 python -m pytest tests -m gpu -q -rxXsf -p no:cacheprovider > "$out/pytest.txt" 2>&1
echo "   exit $?"; tail -5 "$out/pytest.txt"
echo "== knob A/B"
timeout 600 python scripts/knob_ab.py --rounds 3 --steps 3 --legs > "$out/knob_ab.txt" 2>&1
echo "   exit $?"; tail -16 "$out/knob_ab.txt"
for leg in student teacher; do
  echo "== rocprofv3 stats: $leg"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof_$leg" -o leg -- python scripts/leg_prof.py --leg $leg --iters 4 > "$out/prof_$leg.txt" 2>&1
  tail -1 "$out/prof_$leg.txt"
  find "$out/prof_$leg" -name '*kernel_trace.csv' -delete
done
du -sh "$out"
