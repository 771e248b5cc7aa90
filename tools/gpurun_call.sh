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
  for mode in f16 qx; do
    run env RMNET_BANK_PRECISION=$mode timeout 600 python tests/stress_race.py 40
    run env RMNET_BANK_PRECISION=$mode timeout 300 python tests/stress_bank.py
  done
  run timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "f16_mode_vs_oracle or qx_mode_vs_oracle or peaked or qx_mode_on_large or strides_partial or dropin_memory_read_f16 or long_memory or one_launch_limit or repeatable"
  libs="rmnet_amd/librmnet_hip.so $(ls build/variants/lib_*.so 2>/dev/null)"
  for mode in f16 qx; do
    for lib in $libs; do
      for shape in "16 0 0 0 0 5" "8 0 0 0 0 5" "16 21 36 21 36 5" "1 30 54 30 54 5" "20 0 0 0 0 5" "5 0 0 0 0 5"; do
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
  python tools/rocprof_summary.py $out/prof > $out/kernel_table.md 2>> $out/log.txt || true
  find $out/prof -name "*kernel_trace.csv" -size +8M -delete
  ;;
cmd)
  run "$@"
  ;;
*) echo "unknown recipe $recipe"; exit 2;;
esac
