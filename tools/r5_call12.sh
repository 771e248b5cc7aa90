#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
RMNET_HIP_LIB=build/variants/lib_plan2.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bank_read or static_half or repeatable or bank_full or chunks or new_object or five_objects" 2>&1 | tail -1 | sed "s/^/plan2: /"
for p in split f16; do RMNET_HIP_LIB=build/variants/lib_plan2.so RMNET_BANK_PRECISION=$p timeout 600 python tests/stress_bank.py 2>/dev/null | tail -1 | sed "s/^/plan2 stress_bank $p: /"; done
for i in 1 2 3 4; do
  for v in base0 plan2; do
    RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 16 clips: /" >> $O/loop.txt
  done
done
for v in base0 plan2; do
  CLIPS=8 RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v 8 clips: /" >> $O/loop.txt
done
sort $O/loop.txt
