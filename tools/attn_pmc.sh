#!/bin/bash
# SQ counters of the attention kernels (two --pmc passes; gpurun refuses pmc + sys-trace, so kernel-trace only).  usage: tools/attn_pmc.sh r03
set -u
R=${1:-rXX}
export TMPDIR=/tmp
OUT=gpurun_out/attn_pmc_$R
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/p1 -o p1 -- python tools/attn_pmc_run.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p2 -o p2 -- python tools/attn_pmc_run.py > $OUT/p2.log 2>&1
P1=$(find $OUT/p1 -name "*_results.db" | head -1); P2=$(find $OUT/p2 -name "*_results.db" | head -1)
PMC_FILTER="" python tools/pmc_sq.py $OUT/${R}_attn_pmc_raw.md $P1 $P2 | grep -E "win_|seq_|kernel|---" | cut -c1-400
