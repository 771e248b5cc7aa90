#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
for v in kend1 kend2; do
  RMNET_HIP_LIB=build/variants/lib_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bank_read_f16 or bank_read_qx or peaked" 2>&1 | tail -1 | sed "s/^/$v: /"
  RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tests/stress_race.py 100 2>/dev/null | tail -1 | sed "s/^/$v: /"
done
for i in 1 2 3 4; do
  for v in base0 kend1 kend2; do
    RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 16 clips: /" >> $O/loop.txt
  done
done
for v in base0 kend1 kend2; do
  CLIPS=8 RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 8 clips: /" >> $O/loop.txt
  RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 300 python tools/chunk_bench.py 8 0 0 0 0 5 2>/dev/null | tail -1 | sed "s/^/$v warm 8 obj: /" >> $O/loop.txt
done
sort $O/loop.txt
