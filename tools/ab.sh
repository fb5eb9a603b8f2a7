#!/bin/bash
# A/B of one environment switch on the cfg2 bench, alternating on one box: tools/ab.sh VAR A_VALUE B_VALUE [rounds]
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-ref-loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['frac'])"
  done
done
