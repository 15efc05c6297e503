#!/bin/bash
# Round-6: stream-schedule A/B of the C2 step once the host no longer synchronises per step (ops.upload)
set -u
out=gpurun_out/r6streams
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
one() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > "$out/$tag.log" 2>&1
  python - "$out/$tag.log" "$tag" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],2), "images/s", round(d["value"],2))
PY
}
for r in 1 2; do
one "default_$r" FDMI_X=0
one "no_teacher_stream_$r" FDMI_TEACHER_STREAM=0
one "no_defer_$r" FDMI_DEFER_BACKWARD=0
one "single_stream_$r" FDMI_TEACHER_STREAM=0 FDMI_DEFER_BACKWARD=0
done
