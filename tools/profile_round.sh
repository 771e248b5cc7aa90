#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench.py run -> gpurun_out/prof/; tools/rocprof_summary.py
# turns the rocpd database into the per-kernel table committed under profiles/.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
cd /tmp
rm -rf /tmp/prof_rt
rocprofv3 --kernel-trace --stats -d /tmp/prof_rt -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/prof/bench_line.json 2> /tmp/prof_rt.err || true
db=$(find /tmp/prof_rt -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py "$db" 45 --steady 20 > $ROOT/gpurun_out/prof/timed_region.md
tail -60 $ROOT/gpurun_out/prof/timed_region.md
