#!/bin/bash
# Round-5 tenth GPU call: in-process A/B of the GEGLU phase stagger on the C2 step (knob 50 = (groups << 8) | delay in us).
set -u
out=gpurun_out/r5c10
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 900 python scripts/knob_ab.py --rounds 4 --steps 3 --variants base --extra "st3x3:50=771;st3x6:50=774;st2x6:50=518" > "$out/knob_ab.log" 2>&1
grep -v "amdgpu.ids\|KNOB_AB_JSON" "$out/knob_ab.log" | tail -8 | cut -c1-300
