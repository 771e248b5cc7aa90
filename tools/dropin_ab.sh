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
# per-kernel rows of one call (rocprofv3 --kernel-trace --stats of tools/dropin_trace.py without its graph section)
(
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$OLDPWD}
mkdir -p $R/gpurun_out/dropin
NOGRAPH=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dropin/prof -o d -- python $R/tools/dropin_trace.py > $R/gpurun_out/dropin/prof.log 2>&1
f=$(find $R/gpurun_out/dropin/prof -name '*kernel_stats.csv' | head -1)
cat $f | head -20
)
