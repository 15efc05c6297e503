#!/bin/bash
set -u
out=gpurun_out/r2c13
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_knob_ab timeout 600 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "wg512:24=512;wg2048:24=2048;wg512s8:24=512+25=8;wg256:24=256+25=16"
