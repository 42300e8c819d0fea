#!/bin/bash
# Tuning harness (GPU box): whole-library variants ("name:-DFLAGS") timed by scripts/time_train_step.py
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/libv
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -ffp-contract=on -shared $flags \
      epc-net_amd/csrc/*.hip -o /tmp/libv/lib_$name.so 2>/dev/null
  echo -n "$name: "; EPCNET_LIB=/tmp/libv/lib_$name.so python scripts/time_train_step.py 2>/dev/null | tail -1
done
