#!/bin/bash
# Tuning harness (GPU box): build whole-library variants with extra -D flags and print bench.py's stage times.
#   scripts/tune_lib.sh "name1:-DFLAG1 -DFLAG2" "name2:" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/libv
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -ffp-contract=on -shared $flags \
      epc-net_amd/csrc/*.hip -o /tmp/libv/lib_$name.so 2>/dev/null
  EPCNET_LIB=/tmp/libv/lib_$name.so python scripts/time_stages.py $name 2>/dev/null | tail -1
done
