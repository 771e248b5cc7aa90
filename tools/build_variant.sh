#!/bin/bash
# Experiment builds of librmnet_hip.so with extra -D flags -> build/variants/lib_<name>.so (git-ignored; ships to
# the GPU box only while it exists).  Tools pick a variant up through RMNET_HIP_LIB.
#     tools/build_variant.sh clk -DBK_CLK=1
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
cd rmnet_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o ../../build/variants/lib_$name.so \
  capi.hip region_map.hip flow_affine.hip memory_read.hip bank.hip epilogue.hip
echo build/variants/lib_$name.so "$@"
