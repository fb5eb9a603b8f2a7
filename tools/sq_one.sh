#!/bin/bash
# SQ counters (two --pmc passes, kernel-trace only) of one command, kernels filtered by name: tools/sq_one.sh <tag> <filter> <cmd...>
TAG=$1; FIL=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/sq_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/p1 -o p1 -- "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p2 -o p2 -- "$@" > $OUT/p2.log 2>&1
P1=$(find $OUT/p1 -name "*_results.db" | head -1); P2=$(find $OUT/p2 -name "*_results.db" | head -1)
PMC_FILTER="$FIL" python tools/pmc_sq.py $OUT/sq.md $P1 $P2 | cut -c1-600
