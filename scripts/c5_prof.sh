#!/bin/bash
# kernel trace of the lockstep solve of config C5: per-kernel totals -> gpurun_out/c5_summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_c5 -o p -- python3 scripts/c5_only.py > gpurun_out/c5_only.log 2>&1 || { tail -20 gpurun_out/c5_only.log; exit 1; }
DB=$(ls gpurun_out/prof_c5/*.db | head -1)
python3 scripts/prof_summary.py $DB "" 1 > gpurun_out/c5_summary.txt
rm -f gpurun_out/prof_c5/*.db
grep '^{' gpurun_out/c5_only.log | tail -1 | cut -c1-400
head -40 gpurun_out/c5_summary.txt | cut -c1-170
