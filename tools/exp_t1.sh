cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
{
W="8 21 36 21 36 5"
echo "== default lib"; python tools/chunk_bench.py $W 2>&1 | tail -1
echo "== t1"; RMNET_HIP_LIB=build/variants/lib_t1.so python tools/chunk_bench.py $W 2>&1 | tail -1
RMNET_HIP_LIB=build/variants/lib_t1.so python tools/bk_clk.py $W 2>&1 | tail -2
RMNET_HIP_LIB=build/variants/lib_t1.so python tools/dbg_bank.py 2>&1 | tail -12
echo "== iou default"; python tools/iou_terms.py 30 2>&1 | tail -6
echo "== iou t1"; RMNET_HIP_LIB=build/variants/lib_t1.so python tools/iou_terms.py 30 2>&1 | tail -6
} > gpurun_out/t1/log.txt 2>&1
tail -40 gpurun_out/t1/log.txt
