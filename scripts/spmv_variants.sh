#!/bin/bash
# kernel-trace the default bench under several SpMV variants (environment switches); one rocpd database per variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1))
  ( export $v; timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/var_$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/var_$i.log 2>&1 ) || exit 1
  echo "variant $i: $v"; grep '^{' gpurun_out/var_$i.log | cut -c1-130
done
