cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4plan
{
for v in plan plan2 plan plan2; do
echo "=== $v"
RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_$v.so timeout 300 python tools/loop_clk.py 28 2>&1 | grep "in-loop\|plan inputs\|compute WGs"
RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_$v.so timeout 120 python tools/chunk_bench.py 8 21 36 21 36 5 2>&1 | tail -1
done
RMNET_HIP_LIB=build/variants/lib_tplan.so timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bank or f16 or memory_read or long_memory or repeatable" 2>&1 | tail -4
} > gpurun_out/r4plan/log.txt 2>&1
cat gpurun_out/r4plan/log.txt
