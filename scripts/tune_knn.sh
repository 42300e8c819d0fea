#!/bin/bash
# Tuning harness (GPU box): build knn.hip variants into throw-away libraries and time them on Z-ordered clouds.
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/knnv
for v in "$@"; do
  th=${v%%:*}; ct=${v##*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -shared -DKNN_THREADS=$th -DKNN_CT=$ct $KNN_EXTRA \
      epc-net_amd/csrc/knn.hip epc-net_amd/csrc/sort.hip epc-net_amd/csrc/api.hip epc-net_amd/csrc/block.hip -o /tmp/knnv/lib_${th}_${ct}.so
  python scripts/time_knn.py /tmp/knnv/lib_${th}_${ct}.so "threads=$th ct=$ct"
done
