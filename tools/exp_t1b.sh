cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
{
W="8 21 36 21 36 5"
for v in t1 t1a128 t1a256 t1a1024 t1a2 t1a4 t1a17; do
echo "== $v"; RMNET_HIP_LIB=build/variants/lib_$v.so python tools/chunk_bench.py $W 2>&1 | tail -1
RMNET_HIP_LIB=build/variants/lib_$v.so python tools/bk_clk.py $W 2>&1 | tail -2
done
} > gpurun_out/t1/log_b.txt 2>&1
cat gpurun_out/t1/log_b.txt
