#!/bin/bash
# Tuning harness (GPU box): build knn.hip variants of the collect kernel (threads:chunks:cap[:extra flags]) and time them.
# (-DKNN_COLLECT selects knn_collect_kernel; add -DKNN_STATS for the per-wave counters)
cd "$(dirname "$0")/.."
mkdir -p /tmp/knnv
for v in "$@"; do
  IFS=: read th ch cp extra <<< "$v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -shared -DKNN_COLLECT -DKC_THREADS=$th -DKC_CHUNKS=$ch -DKC_CAP=$cp $extra \
      epc-net_amd/csrc/knn.hip epc-net_amd/csrc/sort.hip epc-net_amd/csrc/api.hip epc-net_amd/csrc/block.hip -o /tmp/knnv/libc_${th}_${ch}_${cp}.so 2>&1 | grep -E "error" 
  python scripts/time_knn.py /tmp/knnv/libc_${th}_${ch}_${cp}.so "thr=$th chunks=$ch cap=$cp $extra" 2>&1 | grep -v amdgpu.ids
done
