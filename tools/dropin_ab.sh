#!/bin/bash
# The drop-in entry (rmnet_memory_read_f32: one dense 480p object, T = 5) back to back: device us per call in the three arithmetics,
# the tree's library and build/variants/lib_*.so; then the per-kernel rows of the default arithmetic (rocprofv3).
#     gpurun -- 'tools/gpurun_call.sh <tag> cmd bash tools/dropin_ab.sh'
for flags in 0 4 8; do
  for lib in rmnet_amd/librmnet_hip.so $(ls build/variants/lib_*.so 2>/dev/null); do
    echo "== flags $flags $lib"
    FLAGS=$flags N=200 RMNET_HIP_LIB=$PWD/$lib timeout 300 python tools/dropin_trace.py 2>&1 | tail -1
  done
done
bash tools/exp_dropin.sh
