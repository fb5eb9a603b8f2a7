#!/bin/bash
# kernel-trace summaries of the side workloads (cfg4 Swin-L 384^2, cfg5 retrieval).  usage: tools/collect_side.sh r03
set -u
R=${1:-rXX}
export TMPDIR=/tmp
for W in cfg4 cfg5; do
  OUT=gpurun_out/prof_${R}_$W
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --workload $W --steps 6 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
  KT=$(find $OUT/kt -name "*_results.db" | head -1)
  python tools/rocpd_stats.py $KT $OUT/${R}_${W}_kernel_stats.md > /dev/null
  grep -h "^{\"metric\"" $OUT/kt.log > $OUT/${R}_${W}_bench_line.json
  head -c 400 $OUT/${R}_${W}_bench_line.json; echo
done
