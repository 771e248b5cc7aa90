#!/bin/bash
# round 5, GPU call 2: tests of the round, plan A/B in the loop, the remaining calibration clips, the drop-in entry
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
rm -f gpurun_out/live_iou_table.txt gpurun_out/iou_bar_test_table.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "qx or live_boundary or mutated or graph_replay_is_refused or bench_launches or iou_bar_against or native_library" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for p in f16 qx; do
  for i in 1 2; do
    RMNET_BANK_PRECISION=$p timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/equalised plan: /" >> $O/loop.txt
    RMNET_HIP_LIB=build/variants/lib_noeq.so RMNET_BANK_PRECISION=$p timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/plain plan:     /" >> $O/loop.txt
  done
  RMNET_BANK_PRECISION=$p timeout 300 python tools/chunk_bench.py 5 0 0 0 0 5 2>/dev/null | tail -1 | sed "s/^/$p cfg3 warm, equalised: /" >> $O/chunk.txt
  RMNET_HIP_LIB=build/variants/lib_noeq.so RMNET_BANK_PRECISION=$p timeout 300 python tools/chunk_bench.py 5 0 0 0 0 5 2>/dev/null | tail -1 | sed "s/^/$p cfg3 warm, plain:     /" >> $O/chunk.txt
done
for f in 0 4 8; do FLAGS=$f N=100 timeout 300 python tools/dropin_trace.py 2>/dev/null | tail -1 | sed "s/^/flags $f: /" >> $O/dropin.txt; done
MODES=exact,split,qx,f16 timeout 900 python tools/iou_calib.py 30 16 3o480 1.1 > $O/calib_3o_30.txt 2>&1
MODES=exact,split,qx,f16 timeout 900 python tools/iou_calib.py 20 16 3o480 1.6 > $O/calib_3o_16.txt 2>&1
MODES=exact,split,qx,f16 timeout 900 python tools/iou_calib.py 20 16 3o480,5o480 2.1 > $O/calib_35o_21.txt 2>&1
MODES=exact,split,qx,f16 timeout 1200 python tools/iou_calib.py 20 16 3o720 1.1 > $O/calib_3o720.txt 2>&1
MODES=exact,split,qx,f16 timeout 1200 python tools/iou_calib.py 30 16 5o480 1.1 > $O/calib_5o_30.txt 2>&1
MODES=exact,split,qx,f16 EVERY=5 timeout 1200 python tools/iou_calib.py 67 16 live480-a > $O/calib_live_67.txt 2>&1
MODES=exact,split,qx,f16 timeout 900 python tools/iou_calib.py 30 16 live480-b,live480-c > $O/calib_live_30.txt 2>&1
tail -3 $O/pytest.txt; cat $O/loop.txt $O/chunk.txt $O/dropin.txt
