#!/bin/bash
# same-box A/B of two builds with medians: scripts/ab_median.sh LIB_A LIB_B [alternations] [bench args]
# (a box drifts by +-0.3 ms of setup between runs: single pairs do not resolve 0.1 ms)
A=$1; B=$2; N=${3:-7}; shift; shift; shift
TMP=$(mktemp)
for i in $(seq $N); do
  for L in $A $B; do python scripts/ab_bench_lib.py $L --steps 12 --warmup 2 --no-c5 --no-rank-proxy --no-small --no-dist-one-rank "$@" 2>/dev/null | tail -1 >> $TMP; done
done
python3 - $TMP <<'PY'
import sys, statistics, collections
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    rows[p[0]].append([float(x) for x in p[1:5]])
for k, v in rows.items():
    med = [statistics.median(c) for c in zip(*v)]
    mn = [min(c) for c in zip(*v)]
    print(f"{k:24s} n={len(v)} median: {med[0]:.2f} solves/s {med[1]:.2f} ms/step setup {med[2]:.2f} ms  {med[3]:.1f} us/it | best setup {mn[2]:.2f}")
PY
rm -f $TMP
