cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4calib
{
echo "=== timeline with plan stamps"
RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_cur.so timeout 300 python tools/loop_clk.py 28 2>&1 | grep -v "tickets \[\|amdgpu.ids\|plan records\|^\[\|^ \[" | tail -8
} > gpurun_out/r4calib/timeline.txt 2>&1
cat gpurun_out/r4calib/timeline.txt
timeout 3000 python tools/iou_calib.py 20 32 3o480,5o480 2.1 > gpurun_out/r4calib/calib_size21.txt 2>&1
timeout 3000 python tools/iou_calib.py 20 32 3o480,5o480 1.6 > gpurun_out/r4calib/calib_size16.txt 2>&1
timeout 3000 python tools/iou_calib.py 30 32 3o480,5o480 1.1 > gpurun_out/r4calib/calib_n30.txt 2>&1
grep -h "^[0-9]o\|vs CPU" gpurun_out/r4calib/calib_size21.txt gpurun_out/r4calib/calib_size16.txt gpurun_out/r4calib/calib_n30.txt | cut -c1-150
