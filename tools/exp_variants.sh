# usage: VARIANTS="a b c" [PREC=f16] [STEPS=10] [TESTS=variant] bash tools/exp_variants.sh <outdir>
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
{
export RMNET_BANK_PRECISION=${PREC:-f16}
W="${W:-8 21 36 21 36 5}"
for v in $VARIANTS; do
  echo "=== $v"
  RMNET_HIP_LIB=build/variants/lib_$v.so timeout 300 python tools/loop_clk.py ${STEPS:-10} 2>&1 | grep -v "^\[\|^ \[\|amdgpu.ids\|plan records" | grep -v "tickets \[" | tail -7
  for F in ${FLUSHES:-0 128}; do
    echo -n "chunk_bench FLUSH=$F: "; FLUSH=$F RMNET_HIP_LIB=build/variants/lib_$v.so timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
  done
done
if [ -n "$TESTS" ]; then
  echo "=== tests on $TESTS"; RMNET_HIP_LIB=build/variants/lib_$TESTS.so timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${TESTK:-f16 or bank}" 2>&1 | tail -5
fi
} > $out/log.txt 2>&1
cat $out/log.txt
