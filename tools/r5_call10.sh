#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
for i in 1 2 3; do
  for v in base0 pf4 pf10 pf14 vnt; do
    RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 16 clips: /" >> $O/loop.txt
  done
done
sort $O/loop.txt
