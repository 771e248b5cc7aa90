#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
for v in tr1 tr1w8; do
  for p in f16 qx; do
    echo "=== $v $p (1 object, boxes 46 %, T = 5)" >> $O/trace.txt
    RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=$p timeout 300 python tools/bk_trace.py 0.46 5 2>/dev/null >> $O/trace.txt
  done
done
echo "=== tr1 f16 dense T=20" >> $O/trace.txt
RMNET_HIP_LIB=build/variants/lib_tr1.so RMNET_BANK_PRECISION=f16 timeout 300 python tools/bk_trace.py 1.0 20 2>/dev/null >> $O/trace.txt
cat $O/trace.txt
