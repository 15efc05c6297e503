#!/bin/bash
# Round-5 closing validation, second half: the GPU test files the interrupted closing run (r5_final_tests.sh, stopped by -x at the
# d_lsgan loss bar) had not reached, on the final tree, without -x so that every failure shows.
set -u
out=gpurun_out/r5final4
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/fullsize_parity.txt gpurun_out/unet_parity.txt gpurun_out/multiproc_parity.txt gpurun_out/batch_invariance.txt gpurun_out/flash_parity.txt
SECONDS=0
timeout 285 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_step4_parity_gpu.py tests/test_batch_invariance_gpu.py tests/test_flash_gpu.py \
  tests/test_unet_gpu.py tests/test_pixart_step_gpu.py tests/test_sd3_mmdit_step_gpu.py tests/test_flash_sd3_gpu.py tests/test_fullsize_gpu.py \
  tests/test_multiproc_gpu.py tests/test_sampler_gpu.py tests/test_discriminator_gpu.py tests/test_clip_gpu.py tests/test_t5_gpu.py \
  tests/test_zzz_multigpu_rccl_gpu.py -q -m gpu > "$out/01_pytest.log" 2>&1
echo "   pytest exit $? ; wall: $SECONDS s"; tail -6 "$out/01_pytest.log" | cut -c1-300
