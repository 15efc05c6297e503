#!/bin/bash
set -u
out=gpurun_out/r2c14
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest timeout 900 python -m pytest tests/test_flash_gpu.py tests/test_fullsize_gpu.py tests/test_multiproc_gpu.py tests/test_flash_sd3_gpu.py tests/test_sd3_mmdit_step_gpu.py -q -rxXsf -p no:cacheprovider
run 02_bench_sd3_defer timeout 900 python bench.py --arch sd3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run 03_bench_sd3_nodefer env FDMI_DEFER_BACKWARD=0 timeout 900 python bench.py --arch sd3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run 04_bench_c2_2opt_defer timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
run 05_bench_c2_2opt_nodefer env FDMI_DEFER_BACKWARD=0 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
