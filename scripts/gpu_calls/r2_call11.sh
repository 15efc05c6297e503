#!/bin/bash
set -u
out=gpurun_out/r2c11
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_trace timeout 600 rocprofv3 --kernel-trace -f csv -d "$out/ft" -o ft -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary
run 02_fill_sources python scripts/fill_sources.py "$out/ft"
cat "$out/02_fill_sources.log"
find "$out/ft" -name '*kernel_trace.csv' -delete
run 03_knob_ab timeout 600 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "sk25:21=25;sk100:21=100;nosk:21=100000"
