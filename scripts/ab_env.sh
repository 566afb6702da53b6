#!/bin/bash
# A/B of one environment switch on the same box: scripts/ab_env.sh VAR [repeats]  ->  ms_per_step / setup / iteration per run
VAR=$1; N=${2:-3}
for i in $(seq $N); do
  for on in 0 1; do
    if [ $on = 1 ]; then export $VAR=1; else unset $VAR; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-seam --no-c5 --no-rank-proxy --no-small --no-dist-one-rank 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$on', round(d['ms_per_step'],2), round(d['preconditioner']['setup_ms_per_step'],2), round(d['us_per_iteration'],1), d['iterations'])"
  done
done
