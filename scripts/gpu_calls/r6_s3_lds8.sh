#!/bin/bash
# Round-6: the 16x16x32 attention kernels' 8-byte V^T / X^T fragment reads as single ds_read_b64 (lds_read8) instead of the paired
# ds_read2(st64)_b64 hipcc made of them: isolated rates old (build/variants/pairedlds) vs new, attention parity, bank-conflict counters.
set -u
out=gpurun_out/r6s3lds8
mkdir -p "$out"
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for r in 1 2; do
  export FDMI_LIB="$PWD/build/variants/pairedlds/libfdmi.so"; echo "== paired (old) round $r"; timeout 600 python scripts/attn_bench.py 2>&1 | grep "^attn" | cut -c1-230
  unset FDMI_LIB; echo "== single (new) round $r"; timeout 600 python scripts/attn_bench.py 2>&1 | grep "^attn" | cut -c1-230
done > "$out/attn_bench_ab.txt" 2>&1
cat "$out/attn_bench_ab.txt"
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > "$out/pytest_attn.log" 2>&1; tail -3 "$out/pytest_attn.log"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES \
  --kernel-trace -f csv -d "$out/p2" -o a -- python scripts/attn_pmc.py run > "$out/p2.log" 2>&1
python scripts/attn_pmc.py table "$out/p2" > "$out/table_new.txt" 2>&1
grep -A12 "attn_fwd_kernel<64, 64" "$out/table_new.txt" | head -40
find "$out" -name "*.csv" -size +2M -delete
