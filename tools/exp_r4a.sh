cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
{
export RMNET_BANK_PRECISION=f16
for W in "8 21 36 21 36 5" "8 0 0 0 0 5"; do
 echo "== main lib $W"; timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
 echo "== clk $W"; RMNET_HIP_LIB=build/variants/lib_clk.so timeout 120 python tools/bk_clk.py $W 2>&1 | tail -3
 RMNET_HIP_LIB=build/variants/lib_clk.so timeout 120 python tools/bk_clk_dump.py $W 2>&1 | tail -6
done
echo "== bench f16 in loop"
timeout 900 python bench.py --read-precision f16 --no-cpu-baseline --no-extras --steps 20 2>&1 | tail -2
} > gpurun_out/r4a/log.txt 2>&1
tail -40 gpurun_out/r4a/log.txt
