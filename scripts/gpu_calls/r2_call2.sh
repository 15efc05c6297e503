#!/bin/bash
# Round-2 second GPU call: first runs of C3 / C4 / C5 at their own shapes, the new multi-process tests, the LPIPS-step
# diagnosis, host-thread sweep for the CPU baseline.
set -u
out=gpurun_out/r2c2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest_new timeout 900 python -m pytest tests/test_multiproc_gpu.py tests/test_unet_gpu.py "tests/test_zz_dit_gpu.py::test_step_with_vae_and_lpips_matches_reference_golden" -q -rxXsf -p no:cacheprovider
run 02_bench_sdxl_c3 timeout 600 python bench.py --arch sdxl --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run 03_bench_pixart_c4 timeout 900 python bench.py --arch pixart --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run 04_bench_sd3_c5 timeout 900 python bench.py --arch sd3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run 05_cpu_threads timeout 400 python scripts/cpu_thread_sweep.py
lscpu | head -25 > "$out/lscpu.txt"
