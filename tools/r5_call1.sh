#!/bin/bash
# round 5, GPU call 1: the new arithmetics (mixed, qx) -- kernel-level parity, in-loop time, IoU on the live fixtures and the multi-object clips
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mixed or mutated or graph_replay_is_refused or bench_launches or native_library or live_boundary" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for p in f16 qx mixed split; do
  RMNET_BANK_PRECISION=$p timeout 600 python tools/loop_clk.py 16 2>&1 | head -1 >> $O/loop.txt
  RMNET_BANK_PRECISION=$p timeout 300 python tools/chunk_bench.py 8 0 0 0 0 5 2>&1 | tail -1 | sed "s/^/$p bench-shaped warm: /" >> $O/chunk.txt
  RMNET_BANK_PRECISION=$p FLUSH=600 timeout 300 python tools/chunk_bench.py 8 0 0 0 0 5 2>&1 | tail -1 | sed "s/^/$p bench-shaped cold: /" >> $O/chunk.txt
  RMNET_BANK_PRECISION=$p timeout 300 python tools/chunk_bench.py 5 0 0 0 0 5 2>&1 | tail -1 | sed "s/^/$p cfg3 (5 objects) warm: /" >> $O/chunk.txt
done
for p in f16 qx; do
  RMNET_HIP_LIB=build/variants/lib_noeq.so RMNET_BANK_PRECISION=$p timeout 600 python tools/loop_clk.py 16 2>&1 | head -1 | sed "s/^/plain plan (no equalised blocks): /" >> $O/loop.txt
done
RMNET_HIP_LIB=build/variants/lib_clk.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 > $O/loop_clk_f16.txt 2>&1
RMNET_HIP_LIB=build/variants/lib_clk.so RMNET_BANK_PRECISION=qx timeout 600 python tools/loop_clk.py 16 > $O/loop_clk_qx.txt 2>&1
MODES=exact,split,mixed,qx,f16 timeout 900 python tools/iou_calib.py 0 16 live480-a,live480-b,live480-c > $O/calib_live.txt 2>&1
MODES=exact,mixed,qx,f16 timeout 900 python tools/iou_calib.py 20 16 3o480,5o480 1.1 > $O/calib_multi_11.txt 2>&1
MODES=exact,mixed,qx,f16 timeout 900 python tools/iou_calib.py 20 16 5o480 1.6 > $O/calib_multi_16.txt 2>&1
tail -3 $O/pytest.txt; cat $O/loop.txt $O/chunk.txt
