#!/bin/bash
# Counter passes over the two attention-forward kernels (separate --pmc passes, --kernel-trace only).
set -u
out=${1:-gpurun_out/attn_pmc}
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 -L > "$out/counters_available.txt" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
  --kernel-trace -f csv -d "$out/p1" -o a -- python scripts/attn_pmc.py run > "$out/p1.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM \
  --kernel-trace -f csv -d "$out/p2" -o a -- python scripts/attn_pmc.py run > "$out/p2.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM \
  --kernel-trace -f csv -d "$out/p3" -o a -- python scripts/attn_pmc.py run > "$out/p3.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$out/p4" -o a -- python scripts/attn_pmc.py run > "$out/p4.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace -f csv -d "$out/p0" -o a -- python scripts/attn_pmc.py run > "$out/p0.log" 2>&1
python scripts/attn_pmc.py table "$out/p0" "$out/p1" "$out/p2" "$out/p3" "$out/p4" > "$out/table.txt" 2>&1
cat "$out/table.txt"
tail -3 "$out"/p?.log
find "$out" -name "*.csv" -size +2M -delete
