#!/bin/bash
# Build a throw-away variant of the library HERE (hipcc cross-compiles without a GPU; the .so travels to the GPU box with
# gpurun): recompile only the named kernel file with extra -D flags, link against the other objects of the tree.
#   scripts/build_variant.sh <name> <file.hip> "<flags>"   ->   build_variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
mkdir -p build_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -fno-slp-vectorize $flags -Iepc-net_amd/csrc \
    -c epc-net_amd/csrc/$src -o build_variants/${name}_${src%.hip}.o
objs=""
for f in api sort knn block conv5_vlad conv5_f32 head pack retrieval pipeline train_ops train_head train_head16 train_head32 train_chain train_chain_persist train_hidden; do
  if [ "$f.hip" == "$src" ]; then objs="$objs build_variants/${name}_$f.o"; else objs="$objs epc-net_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/lib_$name.so $objs
rm -f build_variants/${name}_${src%.hip}.o
echo built build_variants/lib_$name.so
