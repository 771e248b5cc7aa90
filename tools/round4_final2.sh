#!/bin/bash
# After the last source change of the round: GPU tests + smoke, PMC traffic of both modes (stamped with the final source hash), the default
# bench line and its rocprofv3 kernel table, and a quick A/B of the final library against the previous one inside the bench loop.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
out=gpurun_out/${1:-r04_final2}
mkdir -p $out
export TMPDIR=/tmp
rm -f gpurun_out/iou_bar_test_table.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
cp gpurun_out/iou_bar_test_table.txt $out/ 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
PRECISION=f16 bash tools/pmc_traffic.sh > $out/pmc_traffic_f16.log 2>&1; tail -1 $out/pmc_traffic_f16.log
bash tools/pmc_traffic.sh > $out/pmc_traffic_split.log 2>&1; tail -1 $out/pmc_traffic_split.log
cp profiles/bk_main_f16_hbm_traffic.json profiles/bk_main_hbm_traffic.json $out/
timeout 1500 python bench.py > $out/bench_line.json 2> $out/bench.err; tail -c 300 $out/bench_line.json; echo; grep "bench " $out/bench.err | tail -4
bash tools/profile_round.sh > $out/profile_round.log 2>&1; cp gpurun_out/prof/timed_region.md $out/bench_default_timed_region.md; cp gpurun_out/prof/bench_line.json $out/bench_line_under_rocprof.json
VARIANTS="main sv main sv" STEPS=40 bash tools/exp_bench_variants.sh $(basename $out)/ab > /dev/null 2>&1; cat $out/ab/log.txt
