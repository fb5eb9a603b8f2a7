#!/bin/bash
# several environment settings of the cfg2 bench, alternating on one box: tools/ab_env.sh rounds "A=1 B=2" "A=3" ...   ("-" = no setting)
N=$1; shift
for i in $(seq $N); do
  for cfg in "$@"; do
    e=$cfg; [ "$cfg" = "-" ] && e=""
    env $e python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-ref-loop 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['ms_per_step'], d['roofline']['frac'])"
  done
done
