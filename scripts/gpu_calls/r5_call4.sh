#!/bin/bash
# Round-5 fourth GPU call: the in-launch split-K reduction -- its parity tests, the whole kernel test file, the unet tests, and the
# in-process A/B on the C2 step (knob 48 = -1: every split GEMM goes to the finalize kernel, as before).
set -u
out=gpurun_out/r5c4
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_pytest_sk timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "inlaunch or splitk" -n 0
tail -15 "$out/01_pytest_sk.log" | cut -c1-300
run 02_pytest_kernels timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q
tail -8 "$out/02_pytest_kernels.log" | cut -c1-300
run 03_knob_ab timeout 900 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "finalize_kernel:48=-1;fuse_upto8:48=8"
tail -12 "$out/03_knob_ab.log" | cut -c1-300
