#!/bin/bash
# A/B of one environment variable between two values on the same box: scripts/ab_envval.sh VAR A B [repeats] [bench args]
VAR=$1; A=$2; B=$3; N=${4:-3}; shift; shift; shift; shift
for i in $(seq $N); do
  for v in "$A" "$B"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-seam --no-c5 --no-rank-proxy --no-small --no-dist-one-rank "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['ms_per_step'],2), round(d['preconditioner']['setup_ms_per_step'],2), round(d['us_per_iteration'],1), d['iterations'])"
  done
done
