#!/bin/bash
set -u
out=gpurun_out/r2c4
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-240))"; }
run 01_pytest timeout 1500 python -m pytest tests/test_fp32_gate_gpu.py tests/test_sd3_mmdit_step_gpu.py tests/test_multiproc_gpu.py tests/test_flash_gpu.py "tests/test_zz_dit_gpu.py::test_step_with_vae_and_lpips_matches_reference_golden" -q -rxXsf -p no:cacheprovider
run 02_bench_sd3_c5_true timeout 900 python bench.py --arch sd3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run 03_bench_pixart_bn192 env FDMI_TUNE=12=1 timeout 900 python bench.py --arch pixart --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run 04_bench_sd3_bn192 env FDMI_TUNE=12=1 timeout 900 python bench.py --arch sd3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
