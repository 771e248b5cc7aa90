# usage: VARIANTS="a b" bash tools/exp_bench_variants.sh <outdir>   -- bk_main inside the REAL bench loop (MIOpen find on), per variant library
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
{
for v in $VARIANTS; do
  echo -n "=== $v: "
  RMNET_HIP_LIB=build/variants/lib_$v.so timeout 600 python bench.py --no-extras --no-cpu-baseline --steps ${STEPS:-40} 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('fps %.1f  bk_main avg %.2f us (min bracket %.2f)  frac %.4f' % (l['value'], r['avg_us'], r['min_us_event_bracket']-r['event_floor_us'], r['frac']))"
done
if [ -n "$CLKV" ]; then echo "=== timeline $CLKV (FIND=1)"; FIND=1 RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_$CLKV.so timeout 600 python tools/loop_clk.py 28 2>&1 | grep "in-loop\|plan inputs\|compute WGs\|set-aside" | sed 's/tickets \[.*\]/tickets [...]/'; fi
} > $out/log.txt 2>&1
cat $out/log.txt
if [ -n "$TESTS" ]; then RMNET_HIP_LIB=build/variants/lib_$TESTS.so timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "${TESTK:-bank or f16 or memory_read or fp16_window}" 2>&1 | tail -3 | tee -a $out/log.txt; fi
