#!/bin/bash
# Everything the round's record needs, in one gpurun call: GPU tests, the default bench line, its rocprofv3 kernel
# table, the PMC traffic passes, the hand-written kernels one by one under rocprofv3, and the power / clock evidence.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
out=gpurun_out/r03_final
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
timeout 1200 python bench.py > $out/bench_line.json 2> $out/bench.err; tail -c 300 $out/bench_line.json; echo
bash tools/profile_round.sh > $out/profile_round.log 2>&1; cp gpurun_out/prof/timed_region.md $out/bench_default_timed_region.md; cp gpurun_out/prof/bench_line.json $out/bench_line_under_rocprof.json
bash tools/pmc_traffic.sh > $out/pmc_traffic.log 2>&1; tail -8 $out/pmc_traffic.log
PRECISION=f16 bash tools/pmc_traffic.sh > $out/pmc_traffic_f16.log 2>&1; tail -3 $out/pmc_traffic_f16.log
# the same default run with the fp16-operand read in the timed region: its own rocprofv3 kernel table
cd /tmp; rm -rf /tmp/prof_f16
rocprofv3 --kernel-trace --stats -d /tmp/prof_f16 -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --read-precision f16 > $ROOT/$out/bench_line_f16_under_rocprof.json 2> /tmp/prof_f16.err || true
db=$(find /tmp/prof_f16 -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py "$db" 45 --steady 20 > $ROOT/$out/bench_f16_timed_region.md 2>&1
cd $ROOT
cd /tmp; rm -rf /tmp/prof_rows
rocprofv3 --kernel-trace --stats -d /tmp/prof_rows -- python $ROOT/tools/kernel_rows.py > $ROOT/$out/kernel_rows.json 2> /tmp/prof_rows.err || true
db=$(find /tmp/prof_rows -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py "$db" 40 > $ROOT/$out/kernel_rows_rocprof.md 2>&1
cd $ROOT
bash tools/power_evidence.sh > /dev/null 2>&1; cp gpurun_out/r03_power/raw.txt $out/power_raw.txt
timeout 600 python tests/stress_race.py 400 > $out/stress_race.log 2>&1; tail -2 $out/stress_race.log
timeout 300 python tests/stress_bank.py > $out/stress_bank.log 2>&1; tail -1 $out/stress_bank.log
RMNET_BANK_PRECISION=f16 timeout 600 python tests/stress_race.py 400 > $out/stress_race_f16.log 2>&1; tail -2 $out/stress_race_f16.log
RMNET_BANK_PRECISION=f16 timeout 300 python tests/stress_bank.py > $out/stress_bank_f16.log 2>&1; tail -1 $out/stress_bank_f16.log
timeout 900 python tools/iou_terms.py 30 > $out/iou_modes.txt 2>&1; tail -10 $out/iou_modes.txt
