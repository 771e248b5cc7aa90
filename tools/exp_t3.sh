cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4t3
{
echo "=== timeline with plan stamps"
RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_cur.so timeout 300 python tools/loop_clk.py 28 2>&1 | grep -v "tickets \[\|amdgpu.ids\|plan records\|^\[\|^ \[" | tail -8
timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "iou_bar or capi or grows_the_memory" --durations=6 2>&1 | tail -15
cat gpurun_out/iou_bar_test_table.txt
} > gpurun_out/r4t3/log.txt 2>&1
cat gpurun_out/r4t3/log.txt
