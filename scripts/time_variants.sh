#!/bin/bash
# GPU box: stage times of every build_variants/lib_*.so given by name (scripts/build_variant.sh builds them on the CPU side).
#   PRECISION=f32 scripts/time_variants.sh base noepi ...
cd "$(dirname "$0")/.."
for name in "$@"; do
  EPCNET_LIB=$PWD/build_variants/lib_$name.so python scripts/time_stages.py $name 2>/dev/null | tail -1
done
