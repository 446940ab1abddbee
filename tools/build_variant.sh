#!/bin/bash
# Build an alternative libeinx_hip.so with extra compiler flags for A/B runs on one GPU box:
#   tools/build_variant.sh NAME "-DEINX_SOMETHING=1"   ->  ab_libs/libeinx_NAME.so
# then:  EINX_LIB=ab_libs/libeinx_NAME.so python bench.py ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/ei-nexus_official_amd/csrc
OUT=/tmp/einx_var_$NAME
mkdir -p "$OUT" "$ROOT/ab_libs"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function $*"
for f in common conv detect desc mnn lightglue events metrics extract; do
  X=""; if [ $f = desc ]; then X="-fno-slp-vectorize"; fi  # as the Makefile
  if [ $f = lightglue ]; then X="-mllvm -amdgpu-mfma-vgpr-form=1"; fi
  /opt/rocm/bin/hipcc $FLAGS $X -c "$SRC/$f.hip" -o "$OUT/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OUT"/*.o -o "$ROOT/ab_libs/libeinx_$NAME.so"
echo "built ab_libs/libeinx_$NAME.so"
