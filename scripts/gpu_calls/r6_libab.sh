#!/bin/bash
# A/B of library builds on the row-kernel bench: r6_libab.sh "lib1 lib2 ..." (paths relative to the repo root; "-" = the in-tree library)
set -u
out=gpurun_out/r6libab
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
for round in 1 2; do
for lib in $1; do
  if [ "$lib" = "-" ]; then unset FDMI_LIB; else export FDMI_LIB="$PWD/$lib"; fi
  echo "== round $round lib $lib"
  timeout 300 python scripts/rowbench.py 20 2>&1 | grep "^M=" | cut -c1-110
done
done
