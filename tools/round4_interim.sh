#!/bin/bash
# Interim record of round 4 in one gpurun call: default bench line, its rocprofv3 kernel table, PMC traffic of the fp16-operand read.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
out=gpurun_out/${1:-r04_interim}
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python bench.py > $out/bench_line.json 2> $out/bench.err; tail -c 400 $out/bench_line.json; echo; tail -3 $out/bench.err
bash tools/profile_round.sh > $out/profile_round.log 2>&1; cp gpurun_out/prof/timed_region.md $out/bench_default_timed_region.md; cp gpurun_out/prof/bench_line.json $out/bench_line_under_rocprof.json
PRECISION=f16 bash tools/pmc_traffic.sh > $out/pmc_traffic_f16.log 2>&1; tail -3 $out/pmc_traffic_f16.log
cp profiles/bk_main_f16_hbm_traffic.json $out/ 2>/dev/null
head -40 $out/bench_default_timed_region.md
