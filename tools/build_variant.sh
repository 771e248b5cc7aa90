#!/bin/bash
# Experiment builds of librmnet_hip.so with extra -D flags -> build/variants/lib_<name>.so (git-ignored; ships to
# the GPU box only while it exists).  Tools pick a variant up through RMNET_HIP_LIB.
#     tools/build_variant.sh clk -DBK_CLK=1
# Only bank.hip and memory_read.hip see the flags (the experiments live there); the other sources are compiled
# once into build/obj/ and re-used while they are newer than their source.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants build/obj
cd rmnet_amd/csrc
for f in capi region_map flow_affine epilogue; do
  o=../../build/obj/$f.o
  if [ ! -f $o ] || [ $f.hip -nt $o ] || [ common.h -nt $o ] || [ ../../include/rmnet_hip.h -nt $o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f.hip -o $o
  fi
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c memory_read.hip -o ../../build/obj/memory_read_$name.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c bank.hip -o ../../build/obj/bank_$name.o
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/lib_$name.so ../../build/obj/capi.o ../../build/obj/region_map.o \
  ../../build/obj/flow_affine.o ../../build/obj/epilogue.o ../../build/obj/memory_read_$name.o ../../build/obj/bank_$name.o
rm -f ../../build/obj/memory_read_$name.o ../../build/obj/bank_$name.o
echo build/variants/lib_$name.so "$@"
