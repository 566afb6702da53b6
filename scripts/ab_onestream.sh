#!/bin/bash
# one-stream kernel trace of the setup (standalone kernel times): scripts/ab_onestream.sh TAG
export PADNE_AMG_ONE_STREAM=1
bash scripts/gpu_prof.sh $1 "" > gpurun_out/$1.log 2>&1
