#!/bin/bash
# GPU parity tests, then a kernel-trace of the default bench; summaries land in gpurun_out/.  Usage: scripts/gpu_check.sh TAG [pytest -k expr]
TAG=${1:-check}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu ${2:+-k "$2"} > gpurun_out/${TAG}_tests.log 2>&1 && \
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/prof_${TAG} -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1
rc=$?
tail -2 gpurun_out/${TAG}_tests.log
grep '^{' gpurun_out/${TAG}_bench.log | cut -c1-420
exit $rc
