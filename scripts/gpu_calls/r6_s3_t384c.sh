#!/bin/bash
# Round-6: the 256 x 384 tile with two-segment A operands, after the per-item re-derivation of the lane offsets (no stray vmcnt wait in the
# K loop): parity, transformer-plan + four-step fixtures, and the step A/B on C5 / C4: 51=1 (no wide tile) | 52=1 (wide tile, two-segment
# problems on 256 x 192) | default (wide tile for both).
set -u
out=gpurun_out/r6s3t384c
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "384 or planner_takes or a2" > "$out/pytest_384.log" 2>&1; tail -3 "$out/pytest_384.log"
timeout 1500 python -m pytest tests/test_zz_dit_gpu.py tests/test_step4_parity_gpu.py -x -q -m gpu > "$out/pytest_dit.log" 2>&1; tail -3 "$out/pytest_dit.log"
b() { timeout 900 python bench.py --arch $1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); f=d['roofline']['families'].get('gemm4_192_row',{}); print('ms_per_step', round(d['ms_per_step'],1), 'gemm4_192_row ms', round(f.get('ms_per_step',0),1), 'frac', round(f.get('frac',0),3))"; }
for r in 1 2; do
  for arch in sd3 pixart; do
    export FDMI_TUNE=51=1; echo "$arch tile192_$r $(b $arch)"
    export FDMI_TUNE=52=1; echo "$arch tile384_one_segment_$r $(b $arch)"
    unset FDMI_TUNE; echo "$arch tile384_$r $(b $arch)"
  done
done | tee "$out/t384_step_ab.txt"
