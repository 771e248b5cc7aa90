cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4tests
timeout 3000 python -m pytest tests -m gpu -q --durations=8 ${TESTARGS} > gpurun_out/r4tests/log.txt 2>&1
tail -40 gpurun_out/r4tests/log.txt
