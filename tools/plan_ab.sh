#!/bin/bash
# Plans and times of the tree's library and build/variants/lib_*.so on a few launches (tools/chunk_bench.py prints the plan record):
#     gpurun -- 'tools/gpurun_call.sh <tag> cmd bash tools/plan_ab.sh'          (SHAPES="12 21 36 21 36 5;8 21 36 21 36 5" MODES="f16")
shapes="${SHAPES:-12 0 0 0 0 5;5 0 0 0 0 5;16 0 0 0 0 5;8 0 0 0 0 5;3 45 80 45 80 20;16 21 36 21 36 5}"
IFS=';' read -ra arr <<< "$shapes"
for mode in ${MODES:-f16 split}; do
  for lib in rmnet_amd/librmnet_hip.so $(ls build/variants/lib_*.so 2>/dev/null); do
    for shape in "${arr[@]}"; do
      echo "== $mode $lib $shape"
      FLUSH=512 RMNET_HIP_LIB=$PWD/$lib RMNET_BANK_PRECISION=$mode timeout 300 python tools/chunk_bench.py $shape 2>&1 | tail -2
    done
  done
done
