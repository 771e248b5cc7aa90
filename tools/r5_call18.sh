#!/bin/bash
# SQ / TCC / TCP counters of bk_main<1> inside the frame loop at the default launch (16 clips) and at 8 (rocprofv3 --pmc, kernel-trace only)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
CLIPS=16 VARIANTS="main" bash tools/pmc_loop.sh r5r_16 > /dev/null 2>&1
CLIPS=8 VARIANTS="main" bash tools/pmc_loop.sh r5r_8 > /dev/null 2>&1
echo "=== 16 clips"; grep -v "available\|^TC" gpurun_out/r5r_16/summary.txt | tail -30
echo "=== 8 clips"; grep -v "available\|^TC" gpurun_out/r5r_8/summary.txt | tail -30
