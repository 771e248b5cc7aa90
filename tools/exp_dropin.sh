#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/dropin
NOGRAPH=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dropin/prof -o d -- python $R/tools/dropin_trace.py > $R/gpurun_out/dropin/prof.log 2>&1
f=$(find $R/gpurun_out/dropin/prof -name '*kernel_stats.csv' | head -1)
cat $f | head -20
