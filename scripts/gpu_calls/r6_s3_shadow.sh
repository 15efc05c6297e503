#!/bin/bash
# Round-6: fp32-stream GEMMs whose consumer is a LayerNorm skip the bf16 shadow store (tape-less runs; switch 53 = 1 stores it always):
# transformer parity (plans, fixtures, full-size forwards), then the step A/B on C4 / C5.
set -u
out=gpurun_out/r6s3shadow
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_zz_dit_gpu.py tests/test_step4_parity_gpu.py tests/test_fullsize_parity_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
b() { timeout 900 python bench.py --arch $1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); f=d['roofline']['families'].get('gemm4_192_row',{}); print('ms_per_step', round(d['ms_per_step'],1), 'gemm4_192_row ms', round(f.get('ms_per_step',0),1))"; }
for r in 1 2; do
  for arch in pixart sd3; do
    export FDMI_TUNE=53=1; echo "$arch shadow_always_$r $(b $arch)"
    unset FDMI_TUNE; echo "$arch shadow_skipped_$r $(b $arch)"
  done
done | tee "$out/shadow_step_ab.txt"
