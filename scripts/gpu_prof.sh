#!/bin/bash
# kernel trace of a short bench; prints kernel time per (name, grid) group per solve.
# Usage: scripts/gpu_prof.sh TAG [filter] [extra bench.py arguments, e.g. --workload C2]
TAG=${1:-prof}; FILTER=${2:-}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_${TAG}
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_${TAG} -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-seam --no-c5 --no-rank-proxy --no-small --no-dist-one-rank "$@" > gpurun_out/${TAG}_bench.log 2>&1 || { tail -20 gpurun_out/${TAG}_bench.log; exit 1; }
DB=$(ls gpurun_out/prof_${TAG}/*.db | head -1)
python scripts/prof_summary.py $DB "$FILTER" 5 > gpurun_out/${TAG}_summary.txt
python scripts/prof_iteration.py $DB iteration > gpurun_out/${TAG}_iteration.txt
python scripts/prof_iteration.py $DB setup > gpurun_out/${TAG}_setup.txt
head -60 gpurun_out/${TAG}_summary.txt
grep '^{' gpurun_out/${TAG}_bench.log | cut -c1-200
python scripts/prof_streams.py $DB > gpurun_out/${TAG}_streams.txt 2>&1; rm -f gpurun_out/prof_${TAG}/*.db
