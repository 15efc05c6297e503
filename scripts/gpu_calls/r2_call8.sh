#!/bin/bash
set -u
out=gpurun_out/r2c8
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest_a timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_sd3_mmdit_step_gpu.py tests/test_flash_sd3_gpu.py tests/test_fullsize_gpu.py -q -rxXsf -p no:cacheprovider
run 02_pytest_b timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_sampler_gpu.py tests/test_zz_dit_gpu.py -q -rxXsf -p no:cacheprovider
run 03_pytest_fp32 timeout 900 python -m pytest tests/test_fp32_gate_gpu.py -q -rxXsf -p no:cacheprovider -k "step"
run 04_bench_pixart timeout 900 python bench.py --arch pixart --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run 05_bench_sd3 timeout 900 python bench.py --arch sd3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run 06_bench_sdxl timeout 900 python bench.py --arch sdxl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
