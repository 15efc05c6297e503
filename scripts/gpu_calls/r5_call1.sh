#!/bin/bash
# Round-5 first GPU call: baseline C2 line of the tree (with the new "comm" block: single-rank RCCL group), the developer
# experiments on the 256x320 kernel (phase stagger, conv K order), the GEMM problem histogram of one step, and the first run of the
# new parity tests (four-teacher-step B = 2 fixtures as far as generated, batch invariance at the benchmarked batch, multiproc).
set -u
out=gpurun_out/r5c1
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > "$out/$name.log" 2>&1; echo "   exit $? ($(tail -1 "$out/$name.log" | cut -c1-300))"; }
run 01_bench timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
run 02_exp1 timeout 400 python scripts/r5_exp1.py all 20
cat "$out/02_exp1.log" | cut -c1-400
run 02b_wgrad_probe timeout 300 python scripts/wgrad_probe.py
tail -12 "$out/02b_wgrad_probe.log" | cut -c1-300
run 02c_wgrad_rates timeout 300 python scripts/wgrad_rates.py 46=1,0
cat "$out/02c_wgrad_rates.log" | cut -c1-400
run 02d_wgrad_rates_blocks timeout 300 python scripts/wgrad_rates.py 45=128,512,1024
cat "$out/02d_wgrad_rates_blocks.log" | cut -c1-400
run 02e_pytest_wgrad timeout 600 python -m pytest tests/test_zz_dit_gpu.py -q -k wgrad_tn
run 03_gemmlog env FDMI_GEMM_LOG=1 timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary
grep GEMMLOG "$out/03_gemmlog.log" | sort -t= -k9 -n > "$out/gemmlog.txt"; wc -l "$out/gemmlog.txt"
run 04_pytest_new timeout 1500 python -m pytest tests/test_batch_invariance_gpu.py tests/test_step4_parity_gpu.py tests/test_multiproc_gpu.py tests/test_multigpu_rccl_gpu.py -q -rxXsf -k "pixart-fp32 or batch or multiproc or rccl"
tail -30 "$out/04_pytest_new.log" | cut -c1-600
cat gpurun_out/batch_invariance.txt gpurun_out/fullsize_parity.txt 2>/dev/null | tail -20 | cut -c1-700
tail -15 gpurun_out/test_durations.txt
