#!/bin/bash
# GPU box: kernel durations of one thin layer forward + backward (scripts/time_thin_graph.py) -> gpurun_out/<tag>/thin_stats.txt
tag=${1:-thin}
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/$tag
cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/$tag/prof_thin -o run -- python3 $R/scripts/time_thin_graph.py > $R/gpurun_out/$tag/prof_thin.log 2>&1
cd $R
python scripts/rocpd_stats.py $(ls gpurun_out/$tag/prof_thin/*.db | head -1) 1 40 > gpurun_out/$tag/thin_stats.txt
rm -rf gpurun_out/$tag/prof_thin
