#!/bin/bash
# Experiment builds of librmnet_hip.so with extra -D flags -> build/variants/lib_<name>.so (git-ignored; ships to
# the GPU box only while it exists).  Tools pick a variant up through RMNET_HIP_LIB.
#     tools/build_variant.sh clk -DBK_CLK=1
#     PATCH=tools/patches/bank_experiment_switches.patch tools/build_variant.sh noeq -DBK_PLAN_NOEQ=1
# PATCH (space-separated list, applied with `patch -p1` to a scratch copy of rmnet_amd/csrc + include under build/src_<name>/):
# the product sources carry no experiment switches any more (tools/strip_switches.py); the switch-laden kernel lives in
# tools/patches/.  Only bank.hip and memory_read.hip see the flags (the experiments live there); the other sources are
# compiled once into build/obj/ and re-used while they are newer than their source.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants build/obj
SRC=rmnet_amd/csrc
if [ -n "$PATCH" ]; then
  rm -rf build/src_$name && mkdir -p build/src_$name/rmnet_amd build/src_$name/include
  cp -r rmnet_amd/csrc build/src_$name/rmnet_amd/ && cp include/rmnet_hip.h build/src_$name/include/
  for p in $PATCH; do (cd build/src_$name && patch -p1 -s < ../../$p); done
  SRC=build/src_$name/rmnet_amd/csrc
fi
ROOT=$PWD
cd rmnet_amd/csrc
for f in capi region_map flow_affine epilogue; do
  o=$ROOT/build/obj/$f.o
  if [ ! -f $o ] || [ $f.hip -nt $o ] || [ common.h -nt $o ] || [ ../../include/rmnet_hip.h -nt $o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f.hip -o $o
  fi
done
cd $ROOT/$SRC
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c memory_read.hip -o $ROOT/build/obj/memory_read_$name.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c bank.hip -o $ROOT/build/obj/bank_$name.o
wait
cd $ROOT
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$name.so build/obj/capi.o build/obj/region_map.o \
  build/obj/flow_affine.o build/obj/epilogue.o build/obj/memory_read_$name.o build/obj/bank_$name.o
rm -f build/obj/memory_read_$name.o build/obj/bank_$name.o
echo build/variants/lib_$name.so "$@" ${PATCH:+(patched: $PATCH)}
