#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root): kernel trace + two PMC passes of bench.py.
# usage: tools/collect_profiles.sh r02
set -u
R=${1:-rXX}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ref-loop > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ref-loop > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ref-loop > $OUT/write.log 2>&1
KT=$(find $OUT/kt -name "*_results.db" | head -1); F=$(find $OUT/fetch -name "*_results.db" | head -1); W=$(find $OUT/write -name "*_results.db" | head -1)
echo "dbs: $KT $F $W"
python tools/rocpd_stats.py $KT $OUT/${R}_bench_kernel_stats.md > /dev/null
# bench.py with --steps 1 --warmup 1 runs 4 steps in all (1 warm-up, 1 timed, 2 roofline sampling passes)
python tools/pmc_traffic.py $F $W $OUT/${R}_pmc_hbm_traffic.md $OUT/${R}_pmc_hbm_traffic.json 4 > /dev/null
tail -n 3 $OUT/kt.log | cut -c1-400
ls -la $OUT
