#!/bin/bash
# Round-6: host-vs-GPU timeline of the C2 step (scripts/host_timeline.py) and the stream's launch look-ahead (scripts/queue_depth.py)
set -u
out=gpurun_out/r6host
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
timeout 400 python scripts/host_timeline.py 4 2 > "$out/host_timeline.txt" 2>&1; echo "exit $?"
cat "$out/host_timeline.txt" | tail -80
timeout 300 python scripts/queue_depth.py > "$out/queue_depth.txt" 2>&1; echo "exit $?"
cat "$out/queue_depth.txt"
