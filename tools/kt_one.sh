#!/bin/bash
# per-kernel average duration of one command: tools/kt_one.sh <tag> <cmd...>   (rocprofv3 --kernel-trace --stats)
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/kt_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o k -- "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT -name "*_results.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/stats.md | head -${KT_ROWS:-12}
