#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
RMNET_HIP_LIB=build/variants/lib_peek.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bank_read or static_half or repeatable" 2>&1 | tail -1 | sed "s/^/peek: /"
for i in 1 2 3 4; do
  for v in base0 peek; do
    RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 16 clips: /" >> $O/loop.txt
  done
done
for v in base0 peek; do
  CLIPS=8 RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 8 clips: /" >> $O/loop.txt
  RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=qx timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 16 clips: /" >> $O/loop.txt
  RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=split timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 16 clips: /" >> $O/loop.txt
done
sort $O/loop.txt
