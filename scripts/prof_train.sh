#!/bin/bash
# GPU box: kernel-level profile of the eager training step (rocprofv3 --kernel-trace, rocpd database) -> gpurun_out/<tag>/train_stats.txt
tag=${1:-train}
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/$tag
cd /tmp && GRAPH=0 rocprofv3 --kernel-trace -d $R/gpurun_out/$tag/prof_train -o run -- python3 $R/scripts/time_train_step.py > $R/gpurun_out/$tag/prof_train.log 2>&1
cd $R
python scripts/rocpd_stats.py $(ls gpurun_out/$tag/prof_train/*.db | head -1) 90 70 > gpurun_out/$tag/train_stats.txt
rm -rf gpurun_out/$tag/prof_train
