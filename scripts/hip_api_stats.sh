#!/bin/bash
# Host-side cost of the HIP calls of a short bench, one-GPU path against the row-partitioned path with a one-rank RCCL
# communicator (18 us per launch there against 5): rocprofv3 --hip-trace --stats of both.
#   scripts/hip_api_stats.sh TAG
TAG=${1:-api}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in one dist; do
  EXTRA=""; [ $mode = dist ] && EXTRA="--force-distributed"
  rm -rf gpurun_out/${TAG}_$mode
  timeout -k 10 300 rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/${TAG}_$mode -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-seam --no-c5 --no-rank-proxy --no-small --no-dist-one-rank $EXTRA > gpurun_out/${TAG}_$mode.log 2>&1 || { tail -5 gpurun_out/${TAG}_$mode.log; exit 1; }
  f=$(find gpurun_out/${TAG}_$mode -name "*hip_api_stats.csv" | head -1)
  echo "== $mode: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_$mode.log | head -1)"
  head -12 $f
  find gpurun_out/${TAG}_$mode -name "*_trace.csv" -size +20M -delete
done
