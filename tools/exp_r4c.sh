cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c
{
export RMNET_BANK_PRECISION=f16
echo "== loop main lib"; timeout 300 python tools/loop_clk.py 10 2>&1 | tail -3
echo "== loop clk lib"; RMNET_HIP_LIB=build/variants/lib_clk.so timeout 300 python tools/loop_clk.py 10 2>&1 | tail -20
W="8 21 36 21 36 5"
for F in 0 32 128 512; do
echo "== clk flush $F"; FLUSH=$F RMNET_HIP_LIB=build/variants/lib_clk.so timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
done
} > gpurun_out/r4c/log.txt 2>&1
cat gpurun_out/r4c/log.txt
