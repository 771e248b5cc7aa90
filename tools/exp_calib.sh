cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4calib
nproc > gpurun_out/r4calib/nproc.txt
timeout 3000 python tools/iou_calib.py ${N:-20} ${NT:-32} ${CASES:-3o480,5o480,3o720} ${SIZE:-1.1} > gpurun_out/r4calib/calib_${TAG:-a}.txt 2>&1
tail -40 gpurun_out/r4calib/calib_${TAG:-a}.txt
