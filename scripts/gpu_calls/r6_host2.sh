#!/bin/bash
# Round-6: after the non-blocking uploads (ops.upload): host timeline again, the C2 line, the step parity tests
set -u
out=gpurun_out/r6host2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
timeout 400 python scripts/host_timeline.py 4 2 > "$out/host_timeline.txt" 2>&1; echo "exit $?"
grep -v "conditioning" "$out/host_timeline.txt" | tail -45
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > "$out/bench.log" 2>&1; echo "exit $?"
tail -1 "$out/bench.log" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_sampler_gpu.py -x -q -m gpu 2>&1 | tail -5
