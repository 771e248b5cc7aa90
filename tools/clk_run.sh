#!/bin/bash
# Per-workgroup time line of bk_main on a few launches, cold caches (tools/bk_clk.py on build/variants/lib_clk.so):
#     PATCH=tools/patches/clk_stamps_r6.patch tools/build_variant.sh clk;  gpurun -- 'tools/gpurun_call.sh <tag> cmd bash tools/clk_run.sh'
shapes="${SHAPES:-1 0 0 0 0 5;1 30 54 30 54 5;4 21 36 21 36 5;8 21 36 21 36 5;12 21 36 21 36 5;16 21 36 21 36 5}"
IFS=';' read -ra arr <<< "$shapes"
for mode in ${MODES:-f16}; do
  for shape in "${arr[@]}"; do
    echo "== $mode $shape"
    FLUSH=512 RMNET_HIP_LIB=$PWD/build/variants/lib_clk.so RMNET_BANK_PRECISION=$mode timeout 300 python tools/bk_clk.py $shape 2>&1 | tail -3
  done
done
