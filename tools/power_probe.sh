#!/bin/bash
# Power / clock samples of the GPU while bench.py runs its timed steps (rocm-smi every 0.5 s; the first samples cover start-up).
# usage: tools/power_probe.sh out.txt
OUT=${1:-gpurun_out/power.txt}
python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-ref-loop > ${OUT%.txt}_bench.json 2>/dev/null &
BP=$!
: > $OUT
for i in $(seq 1 200); do
  kill -0 $BP 2>/dev/null || break
  echo -n "t=$i " >> $OUT
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i -E "power|sclk|mclk|junction" | sed 's/.*GPU\[0\][ :]*//' | tr -s ' ' | tr '\n' ';' >> $OUT
  echo >> $OUT
  sleep 0.5
done
wait $BP
cut -c1-300 $OUT | tail -45
cut -c1-300 ${OUT%.txt}_bench.json
