#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5p; mkdir -p $O
for v in tr3w4 tr3w8; do
  echo "=== $v f16 (1 object, boxes 46 %, T = 5)" >> $O/trace.txt
  TRACE2=2 RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 300 python tools/bk_trace.py 0.46 5 2>/dev/null | grep -A6 "consumer" >> $O/trace.txt
  echo "=== $v f16 dense T=20" >> $O/trace.txt
  TRACE2=2 RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 300 python tools/bk_trace.py 1.0 20 2>/dev/null | grep -A6 "consumer" >> $O/trace.txt
done
cat $O/trace.txt
