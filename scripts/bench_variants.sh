#!/bin/bash
# default bench under several environment variants (no profiler)
cd $GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1))
  ( export $v; timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bv_$i.log 2>&1 ) || exit 1
  echo "variant $i: $v"; grep '^{' gpurun_out/bv_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), d.get('iterations'), d.get('preconditioner'))"
done
