#!/bin/bash
# After the change of the default launch size (16 clips per GPU): the default bench line, its rocprofv3 kernel table, PMC traffic of the three
# arithmetics at that launch, the loop time line at 16 and at 8 clips, smoke.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
out=gpurun_out/${1:-r05_final2}
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 1800 python bench.py > $out/bench_line.json 2> $out/bench.err; tail -c 200 $out/bench_line.json; echo; grep "bench " $out/bench.err | tail -3
bash tools/profile_round.sh > $out/profile_round.log 2>&1; cp gpurun_out/prof/timed_region.md $out/bench_default_timed_region.md; cp gpurun_out/prof/bench_line.json $out/bench_line_under_rocprof.json
for p in f16 qx split; do PRECISION=$p bash tools/pmc_traffic.sh > $out/pmc_traffic_$p.log 2>&1; tail -1 $out/pmc_traffic_$p.log; done
cp profiles/bk_main_f16_hbm_traffic.json profiles/bk_main_qx_hbm_traffic.json profiles/bk_main_hbm_traffic.json $out/ 2>/dev/null
for c in 16 8; do
  CLIPS=$c FIND=1 RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_clk.so timeout 600 python tools/loop_clk.py 28 2>/dev/null > $out/loop_timeline_$c.txt; grep "in-loop\|plan inputs\|compute WGs\|set-aside" $out/loop_timeline_$c.txt | cut -c1-260
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bench" > $out/pytest_bench.txt 2>&1; tail -1 $out/pytest_bench.txt
