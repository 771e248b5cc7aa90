#!/bin/bash
# The round-5 record in one gpurun call: GPU tests + smoke, the default bench line, its rocprofv3 kernel table, PMC traffic of the three
# arithmetics (stamped with bench.source_hash()), every hand-written kernel alone under rocprofv3, stress tests, the loop time line.
# Before the call (build container): PATCH=tools/patches/bank_experiment_switches.patch tools/build_variant.sh clk -DBK_CLK=1
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
out=gpurun_out/${1:-r05_final}
mkdir -p $out
export TMPDIR=/tmp
rm -f gpurun_out/iou_bar_test_table.txt gpurun_out/live_iou_table.txt
timeout 2700 python -m pytest tests -m gpu -q --durations=8 > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
cp gpurun_out/iou_bar_test_table.txt gpurun_out/live_iou_table.txt $out/ 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 1800 python bench.py > $out/bench_line.json 2> $out/bench.err; tail -c 300 $out/bench_line.json; echo; grep "bench " $out/bench.err | tail -3
bash tools/profile_round.sh > $out/profile_round.log 2>&1; cp gpurun_out/prof/timed_region.md $out/bench_default_timed_region.md; cp gpurun_out/prof/bench_line.json $out/bench_line_under_rocprof.json
for p in f16 qx split; do PRECISION=$p bash tools/pmc_traffic.sh > $out/pmc_traffic_$p.log 2>&1; tail -1 $out/pmc_traffic_$p.log; done
cp profiles/bk_main_f16_hbm_traffic.json profiles/bk_main_qx_hbm_traffic.json profiles/bk_main_hbm_traffic.json $out/ 2>/dev/null
cd /tmp; rm -rf /tmp/prof_rows
rocprofv3 --kernel-trace --stats -d /tmp/prof_rows -- python $ROOT/tools/kernel_rows.py > $ROOT/$out/kernel_rows.json 2> /tmp/prof_rows.err || true
db=$(find /tmp/prof_rows -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py "$db" 40 > $ROOT/$out/kernel_rows_rocprof.md 2>&1
cd $ROOT
for p in split f16 qx; do
  RMNET_BANK_PRECISION=$p timeout 600 python tests/stress_race.py 300 2>/dev/null | tail -1 | sed "s/^/stress_race $p: /" >> $out/stress.log
  RMNET_BANK_PRECISION=$p timeout 300 python tests/stress_bank.py 2>/dev/null | tail -1 | sed "s/^/stress_bank $p: /" >> $out/stress.log
done
cat $out/stress.log
FIND=1 RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_clk.so timeout 600 python tools/loop_clk.py 28 2>/dev/null > $out/loop_timeline.txt; grep "in-loop\|plan inputs\|compute WGs" $out/loop_timeline.txt
python bench.py --gpus 2 --dist-backend gloo --steps 4 --warmup 2 --no-cpu-baseline --no-extras --clips-per-gpu 2 > $out/bench_2ranks_gloo.json 2> $out/bench_2ranks.err; tail -c 400 $out/bench_2ranks_gloo.json
