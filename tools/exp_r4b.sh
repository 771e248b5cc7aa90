cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4b
{
export RMNET_BANK_PRECISION=f16
W="8 21 36 21 36 5"
echo "== base"; timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
echo "== staged"; STAGED=1 timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
echo "== flush 64MB"; FLUSH=64 timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
echo "== flush 1024MB"; FLUSH=1024 timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
echo "== flush 1024MB staged"; FLUSH=1024 STAGED=1 timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
export RMNET_BANK_PRECISION=split
echo "== split base"; timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
echo "== split flush 1024MB"; FLUSH=1024 timeout 120 python tools/chunk_bench.py $W 2>&1 | tail -1
} > gpurun_out/r4b/log.txt 2>&1
cat gpurun_out/r4b/log.txt
