cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
{ timeout 900 python tools/iou_terms.py ${N:-30} 2>&1 | grep -v amdgpu.ids; } > gpurun_out/t1/log_iou.txt 2>&1
cat gpurun_out/t1/log_iou.txt
