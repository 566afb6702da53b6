#!/bin/bash
# same-box A/B of two builds: scripts/ab_libs.sh LIB_A LIB_B [repeats] [bench args]
A=$1; B=$2; N=${3:-3}; shift; shift; shift
for i in $(seq $N); do
  for L in $A $B; do python scripts/ab_bench_lib.py $L --steps 20 --warmup 3 --no-c5 --no-rank-proxy --no-small --no-dist-one-rank "$@" 2>/dev/null | tail -1; done
done
