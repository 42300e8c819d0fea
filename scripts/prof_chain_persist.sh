#!/bin/bash
# GPU box: kernel durations of the backbone chain (launch chain and persistent launches) -> gpurun_out/<tag>/chain_persist_stats_<ncl>.txt
tag=${1:-persist}
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/$tag
for NCL in ${NCLS:-18 22}; do
  export NCL
  cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/$tag/prof_cp -o run -- python3 $R/scripts/time_chain_persist.py > $R/gpurun_out/$tag/prof_cp_$NCL.log 2>&1
  cd $R
  python scripts/rocpd_stats.py $(ls gpurun_out/$tag/prof_cp/*.db | head -1) 1 40 > gpurun_out/$tag/chain_persist_stats_$NCL.txt
  rm -rf gpurun_out/$tag/prof_cp
done
