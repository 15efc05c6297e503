#!/bin/bash
# Round-4 first GPU call: (1) gemm4 row-kernel variants (L2 prefetch touches / burst issue / timing ablations) on the HBM-bound
# shapes, (2) which kernel family breaks `rocprofv3 --pmc` (scripts/pmc_probe.py under one FETCH_SIZE pass), (3) the variants in the
# whole C2 step (in-process A/B), (4) the GEMM kernel tests on the default build
set -u
out=gpurun_out/r4c1
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 400 python scripts/rowbench.py dev 30 > "$out/rowbench_dev.txt" 2>&1
cat "$out/rowbench_dev.txt" | cut -c1-900
echo "== pmc probe"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/pmc_probe" -o p -- python scripts/pmc_probe.py > "$out/pmc_probe.log" 2>&1
echo "rc=$?"; grep -E "^(launching|ok |all families)" "$out/pmc_probe.log" | tail -4; grep -c INVALID_PACKET "$out/pmc_probe.log"
find "$out/pmc_probe" -name '*.csv' -size +1M -delete 2>/dev/null
echo "== knob A/B"
timeout 700 python scripts/knob_ab.py --rounds 3 --steps 3 --variants base --extra "pf2r:40=6;pf3r:40=7;pf2:40=2;burst_pf2r:40=14;res_only:40=4" > "$out/knob_ab.log" 2>&1
grep -E "^(base|pf|burst|res_only|variant)" "$out/knob_ab.log" | cut -c1-200
echo "== gemm kernel tests (default build)"
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > "$out/pytest_gemm.log" 2>&1; tail -3 "$out/pytest_gemm.log"
