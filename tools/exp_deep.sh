cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
{
for WW in "8 21 36 21 36 5" "8 0 0 0 0 5"; do
for v in clk deep; do
echo "== $v $WW"; RMNET_HIP_LIB=build/variants/lib_$v.so timeout 120 python tools/chunk_bench.py $WW 2>&1 | tail -1
RMNET_HIP_LIB=build/variants/lib_$v.so timeout 120 python tools/bk_clk.py $WW 2>&1 | tail -1
done; done
RMNET_HIP_LIB=build/variants/lib_deep.so timeout 300 python tools/dbg_bank.py 2>&1 | tail -3
} > gpurun_out/t1/log_deep.txt 2>&1
cat gpurun_out/t1/log_deep.txt
