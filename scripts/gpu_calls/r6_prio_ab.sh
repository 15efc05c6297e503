#!/bin/bash
# Round-6: hardware queue priority of the teacher loop's stream, A/B on one box (20 steps each, two rounds)
set -u
out=gpurun_out/r6prio
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
one() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > "$out/$tag.log" 2>&1
  python - "$out/$tag.log" "$tag" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],2), "images/s", round(d["value"],2))
PY
}
for r in 1 2; do
one "default_$r" FDMI_TEACHER_PRIORITY=0
one "teacher_high_$r" FDMI_TEACHER_PRIORITY=-1
done
