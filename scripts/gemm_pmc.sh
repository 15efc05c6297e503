#!/bin/bash
# Counter passes over the GEMM / conv kernels (separate --pmc passes, --kernel-trace only): scripts/gemm_pmc.py.
set -u
out=${1:-gpurun_out/gemm_pmc}
mkdir -p "$out"
export TMPDIR=/tmp
pass() { name=$1; shift; timeout -s KILL 300 rocprofv3 "$@" --kernel-trace -f csv -d "$out/$name" -o a -- python scripts/gemm_pmc.py run > "$out/$name.log" 2>&1; echo "pass $name exit $?"; }
pass p0
pass p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
pass p2 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM
pass p3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL
python scripts/gemm_pmc.py table "$out/p0" "$out/p1" "$out/p2" "$out/p3" > "$out/table.txt" 2>&1
tail -2 "$out"/p?.log
find "$out" -name "*.csv" -size +2M -delete
