#!/bin/bash
# config C5 (8 right-hand sides in lockstep) with and without the staged runs of X in the 8-wide products, same box:
#   scripts/c5_ab.sh [repeats]
N=${1:-2}
for i in $(seq $N); do
  for v in - spmm_gather; do
    if [ "$v" = "-" ]; then unset PADNE_FORCE; else export PADNE_FORCE=$v; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-seam --no-rank-proxy --no-small --no-dist-one-rank 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['c5']; print('PADNE_FORCE=$v', 'c5 ms', round(c['ms'],2), 'single', round(c['single_solve_ms'],2), 'equiv', round(c['solves_equiv'],2), 'parts', round(c['solve_parts_equiv'],2), 'spmm8 us', round(c['spmm8_us'],1), 'frac', round(c['spmm8_frac'],3), 'iters', c['iterations_total'])"
  done
done
