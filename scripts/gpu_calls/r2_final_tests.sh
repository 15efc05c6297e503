#!/bin/bash
# the round-end check, as the driver runs it: the whole GPU suite, then smoke()
set -u
out=gpurun_out/r2final
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest timeout 2400 python -m pytest tests -m gpu -q -rxXsf -p no:cacheprovider
run 02_smoke timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
