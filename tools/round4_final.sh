#!/bin/bash
# The round's record in one gpurun call: GPU tests, default bench line, its rocprofv3 kernel table, PMC traffic (both modes), SQ counters,
# the hand-written kernels one by one under rocprofv3, stress tests, the DPP helper check, the loop time line.
# Before the call (build container): tools/build_variant.sh finalclk -DBK_CLK=1; cp rmnet_amd/librmnet_hip.so build/variants/lib_main.so;
# hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/dpp_check.hip -o tools/ubench/dpp_check   (build/ and the binary are git-ignored, they travel with gpurun)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
out=gpurun_out/${1:-r04_final}
mkdir -p $out
export TMPDIR=/tmp
rm -f gpurun_out/iou_bar_test_table.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
cp gpurun_out/iou_bar_test_table.txt $out/ 2>/dev/null
timeout 1500 python bench.py > $out/bench_line.json 2> $out/bench.err; tail -c 300 $out/bench_line.json; echo
bash tools/profile_round.sh > $out/profile_round.log 2>&1; cp gpurun_out/prof/timed_region.md $out/bench_default_timed_region.md; cp gpurun_out/prof/bench_line.json $out/bench_line_under_rocprof.json
PRECISION=f16 bash tools/pmc_traffic.sh > $out/pmc_traffic_f16.log 2>&1; tail -3 $out/pmc_traffic_f16.log
bash tools/pmc_traffic.sh > $out/pmc_traffic_split.log 2>&1; tail -3 $out/pmc_traffic_split.log
cp profiles/bk_main_f16_hbm_traffic.json profiles/bk_main_hbm_traffic.json $out/
bash tools/pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma/summary.txt $out/pmc_mfma_uniform_launch.txt
VARIANTS="main" bash tools/pmc_loop.sh $(basename $out)/pmc_loop > /dev/null 2>&1
cd /tmp; rm -rf /tmp/prof_rows
rocprofv3 --kernel-trace --stats -d /tmp/prof_rows -- python $ROOT/tools/kernel_rows.py > $ROOT/$out/kernel_rows.json 2> /tmp/prof_rows.err || true
db=$(find /tmp/prof_rows -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py "$db" 40 > $ROOT/$out/kernel_rows_rocprof.md 2>&1
cd $ROOT
tools/ubench/dpp_check > $out/dpp_check.log 2>&1; tail -1 $out/dpp_check.log
timeout 600 python tests/stress_race.py 400 > $out/stress_race.log 2>&1; tail -2 $out/stress_race.log
timeout 300 python tests/stress_bank.py > $out/stress_bank.log 2>&1; tail -1 $out/stress_bank.log
RMNET_BANK_PRECISION=f16 timeout 600 python tests/stress_race.py 400 > $out/stress_race_f16.log 2>&1; tail -2 $out/stress_race_f16.log
RMNET_BANK_PRECISION=f16 timeout 300 python tests/stress_bank.py > $out/stress_bank_f16.log 2>&1; tail -1 $out/stress_bank_f16.log
FIND=1 RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_finalclk.so timeout 600 python tools/loop_clk.py 28 > $out/loop_timeline.txt 2>&1; grep "in-loop\|plan inputs\|compute WGs" $out/loop_timeline.txt
