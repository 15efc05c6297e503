#!/bin/bash
# Round-3 second GPU call: the tests that failed in call 1 (virtual-concat eligibility, test bugs) on the fixed build without the
# 128x160 geometry, the new frozen-net plans (VAE decoder / LPIPS-VGG16 / T2I adapter: tests/test_nets_gpu.py), and the bench line.
set -u
out=gpurun_out/r3c2
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest_nets timeout 1500 python -m pytest tests/test_nets_gpu.py -q -rxXsf -p no:cacheprovider
grep -h "FAILED\|passed\|failed" "$out/01_pytest_nets.log" | tail -20
cat gpurun_out/nets_parity.txt 2>/dev/null | cut -c1-400
run 02_pytest_fixed timeout 1800 python -m pytest tests/test_flash_gpu.py tests/test_multiproc_gpu.py tests/test_sampler_gpu.py tests/test_unet_gpu.py tests/test_zz_dit_gpu.py tests/test_kernels_gpu.py tests/test_discriminator_gpu.py tests/test_fullsize_gpu.py -q -rxXsf -p no:cacheprovider
grep -h "FAILED\|passed\|failed" "$out/02_pytest_fixed.log" | tail -20
run 03_pytest_c2 timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py -q -rxXsf -p no:cacheprovider -k "c2 or c1"
grep -h "FAILED\|passed\|failed" "$out/03_pytest_c2.log" | tail -5
run 04_bench timeout 1200 python bench.py
tail -c 3000 "$out/04_bench.log"
cp gpurun_out/*.txt "$out/" 2>/dev/null
