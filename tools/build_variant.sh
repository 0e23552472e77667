#!/bin/bash
# Build a developer variant of the library into build/variants/libhmj_<name>.so with extra -D flags
# (A/B experiments on the GPU box: HMJ_LIB=build/variants/libhmj_<name>.so python tools/...).
# Usage: tools/build_variant.sh <name> [-DFLAG ...]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants
mkdir -p $OUT/obj_$NAME
for f in radix probe gen gtable api exchange; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DHMJ_DEV "$@" \
    -c $ROOT/hashmergejoin_amd/csrc/$f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libhmj_$NAME.so $OUT/obj_$NAME/*.o -ldl
echo built $OUT/libhmj_$NAME.so
