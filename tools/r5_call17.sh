#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5q; mkdir -p $O
RMNET_HIP_LIB=build/variants/lib_kdma.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bank_read_f16 or bank_read_qx or peaked or strides or chunks" 2>&1 | tail -2 | sed "s/^/kdma: /"
for p in f16 qx; do
  RMNET_HIP_LIB=build/variants/lib_kdma.so RMNET_BANK_PRECISION=$p timeout 600 python tests/stress_race.py 300 2>/dev/null | tail -2 | sed "s/^/kdma $p: /"
  RMNET_HIP_LIB=build/variants/lib_kdma.so RMNET_BANK_PRECISION=$p timeout 600 python tests/stress_bank.py 2>/dev/null | tail -1 | sed "s/^/kdma $p: /"
done
for i in 1 2 3 4; do
  for v in base0 kdma; do
    RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 16 clips: /" >> $O/loop.txt
  done
done
for v in base0 kdma; do
  CLIPS=8 RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 8 clips: /" >> $O/loop.txt
  RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=qx timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 16 clips: /" >> $O/loop.txt
  RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 300 python tools/chunk_bench.py 8 0 0 0 0 5 2>/dev/null | tail -1 | sed "s/^/$v warm 8 obj: /" >> $O/loop.txt
done
sort $O/loop.txt
