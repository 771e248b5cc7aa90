#!/bin/bash
# One parametrised wrapper for a round's `gpurun` calls (replaces the per-call one-shot scripts of round 5):
#     gpurun --timeout 1500 -- 'tools/gpurun_call.sh <tag> <recipe> [args...]'
# Every recipe writes its log(s) to gpurun_out/<tag>/ (merged back by gpurun); evidence is copied to profiles/ by hand.
# Recipes:
#   walk_ab        parity (fp16 / qx bank tests + stress) of the tree's kernel, then bk_main alone (tools/chunk_bench.py) with the
#                  tree's library and with every build/variants/lib_*.so, warm and with FLUSH=512 (cold caches, the loop's condition)
#   tests [-k e]   pytest -m gpu (optionally a -k expression)
#   bench [args]   python bench.py [args]
#   prof  [args]   rocprofv3 --kernel-trace --stats of bench.py [args] -> kernel table (tools/rocprof_summary.py)
#   conv_layout    the convolution half of 8f-3: fused / unfused / folded / channels_last frames/s + the NHWC run's kernel table
#   record         the round's record: suite + smoke, default bench line, rocprofv3 table, PMC traffic x 3, in-loop counters, stress
#   record_lite    the record without the PMC traffic / counter passes
#   pmc            the PMC half of the record: HBM traffic x 3 arithmetics, in-loop counters
#   cmd   <...>    any command line
set -u
cd "$(dirname "$0")/.."
tag=$1; recipe=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
log() { echo "### $*" | tee -a $out/log.txt; }
run() { log "$*"; ( "$@" ) >> $out/log.txt 2>&1; echo "rc=$?" | tee -a $out/log.txt; }

case $recipe in
walk_ab)
  for mode in ${STRESS_MODES:-f16 qx}; do
    run env RMNET_BANK_PRECISION=$mode timeout 600 python tests/stress_race.py 40
    run env RMNET_BANK_PRECISION=$mode timeout 300 python tests/stress_bank.py
  done
  run timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "f16_mode_vs_oracle or qx_mode_vs_oracle or peaked or qx_mode_on_large or strides_partial or dropin_memory_read_f16 or long_memory or one_launch_limit or repeatable"
  libs="rmnet_amd/librmnet_hip.so $(ls build/variants/lib_*.so 2>/dev/null)"
  for mode in ${MODES:-f16 qx}; do
    for lib in $libs; do
      for shape in "16 0 0 0 0 5" "8 0 0 0 0 5" "16 21 36 21 36 5" "1 30 54 30 54 5" "20 0 0 0 0 5" "5 0 0 0 0 5" ${MORE_SHAPES:+"1 0 0 0 0 5" "4 0 0 0 0 5" "12 0 0 0 0 5" "3 45 80 45 80 20"}; do
        log "$mode $lib chunk_bench $shape"
        RMNET_HIP_LIB=$PWD/$lib RMNET_BANK_PRECISION=$mode timeout 300 python tools/chunk_bench.py $shape 2>&1 | tail -1 | tee -a $out/log.txt
        log "$mode $lib FLUSH=512 chunk_bench $shape"
        FLUSH=512 RMNET_HIP_LIB=$PWD/$lib RMNET_BANK_PRECISION=$mode timeout 300 python tools/chunk_bench.py $shape 2>&1 | tail -1 | tee -a $out/log.txt
      done
    done
  done
  ;;
tests)
  run timeout 1500 python -m pytest tests -q -x -m gpu "$@"
  ;;
bench)
  log "python bench.py $*"
  timeout 1200 python bench.py "$@" > $out/bench.json 2> $out/bench.err; echo "rc=$?" | tee -a $out/log.txt
  tail -c 3000 $out/bench.json
  ;;
prof)
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/prof -o run -- python $OLDPWD/bench.py "$@" > $OLDPWD/$out/bench_under_rocprof.json 2> $OLDPWD/$out/prof.err
  cd $OLDPWD
  db=$(find $out/prof -name "*.db" | head -1)
  python tools/rocprof_summary.py $db 45 --steady 20 > $out/kernel_table.md 2>> $out/log.txt || true
  find $out/prof -type f -size +4M -delete
  ;;
conv_layout)
  # the convolution half of SURVEY 8f-3: frames/s of the bench loop with the elementwise glue fused (default, NCHW), unfused, with the
  # BatchNorms folded into the convolutions, and channels_last (unfused; MIOpen asked for NHWC); rocprofv3 kernel table of the last
  for v in "fused_nchw" "unfused_nchw --no-fuse-epilogue" "foldbn_nchw --fold-bn" "unfused_nhwc --channels-last"; do
    set -- $v; name=$1; shift
    log "$name: bench.py --no-extras --no-cpu-baseline $*"
    PYTORCH_MIOPEN_SUGGEST_NHWC=$([ "$name" = unfused_nhwc ] && echo 1 || echo 0) timeout 900 python bench.py --no-extras --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
    python tools/show_line.py $out/$name.json 2>&1 | head -1 | tee -a $out/log.txt
  done
  cd /tmp
  PYTORCH_MIOPEN_SUGGEST_NHWC=1 rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/prof_nhwc -o run -- python $OLDPWD/bench.py --no-extras --no-cpu-baseline --channels-last > $OLDPWD/$out/nhwc_under_rocprof.json 2> $OLDPWD/$out/prof.err
  cd $OLDPWD
  db=$(find $out/prof_nhwc -name "*.db" | head -1)
  python tools/rocprof_summary.py $db 45 --steady 20 > $out/kernel_table_nhwc.md 2>> $out/log.txt || true
  find $out/prof_nhwc -type f -size +4M -delete
  ;;
record)
  # a round's record on one box: GPU suite + smoke, the default bench line, its rocprofv3 kernel table (same command), PMC traffic of the
  # three arithmetics (stamped with bench.source_hash()), SQ / TCC counters of bk_main inside the loop, stress.  Evidence -> profiles/ by hand.
  run timeout 1500 python -m pytest tests -q -m gpu --durations=15
  run python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
  log "python bench.py"
  timeout 1200 python bench.py > $out/bench_line.json 2> $out/bench.err; python tools/show_line.py $out/bench_line.json | tee -a $out/log.txt
  bash tools/profile_round.sh > $out/profile_round.txt 2>&1; cp gpurun_out/prof/timed_region.md $out/timed_region.md; cp gpurun_out/prof/bench_line.json $out/bench_line_under_rocprof.json
  for prec in f16 qx split; do PRECISION=$prec bash tools/pmc_traffic.sh > $out/pmc_traffic_$prec.txt 2>&1; done
  cp profiles/bk_main*_hbm_traffic.json $out/
  mkdir -p build/variants; cp rmnet_amd/librmnet_hip.so build/variants/lib_main.so
  VARIANTS="main" bash tools/pmc_loop.sh $tag/counters > /dev/null 2>&1; cp gpurun_out/$tag/counters/summary.txt $out/counters.txt
  for mode in f16 qx split; do RMNET_BANK_PRECISION=$mode timeout 900 python tests/stress_race.py 150 2>&1 | tail -1; done | tee $out/stress.txt
  ;;
record_lite)
  # the record without the PMC passes (when GPU minutes are short): suite + smoke, default bench line, rocprofv3 table, drop-in rows, stress
  run timeout 1400 python -m pytest tests -q -m gpu --durations=15
  run python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
  log "python bench.py"
  timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; python tools/show_line.py $out/bench_line.json | tee -a $out/log.txt
  bash tools/profile_round.sh > $out/profile_round.txt 2>&1; cp gpurun_out/prof/timed_region.md $out/timed_region.md; cp gpurun_out/prof/bench_line.json $out/bench_line_under_rocprof.json
  for flags in 0 4 8; do echo "== drop-in flags $flags"; FLAGS=$flags N=200 timeout 300 python tools/dropin_trace.py 2>&1 | tail -1; done | tee $out/dropin.txt
  for mode in f16 qx split; do RMNET_BANK_PRECISION=$mode timeout 400 python tests/stress_race.py ${STRESS_N:-60} 2>&1 | tail -1; done | tee $out/stress.txt
  ;;
pmc)
  # the PMC half of the record: HBM traffic of bk_main in the three arithmetics (separate --pmc passes, stamped with bench.source_hash()),
  # then the SQ / TCC counters of bk_main inside the loop
  for prec in ${PRECS:-f16 qx split}; do PRECISION=$prec bash tools/pmc_traffic.sh > $out/pmc_traffic_$prec.txt 2>&1; tail -2 $out/pmc_traffic_$prec.txt; done
  cp profiles/bk_main*_hbm_traffic.json $out/
  mkdir -p build/variants; cp rmnet_amd/librmnet_hip.so build/variants/lib_main.so
  VARIANTS="main" bash tools/pmc_loop.sh $tag/counters > /dev/null 2>&1; cp gpurun_out/$tag/counters/summary.txt $out/counters.txt; tail -30 $out/counters.txt
  ;;
cmd)
  run "$@"
  ;;
*) echo "unknown recipe $recipe"; exit 2;;
esac
