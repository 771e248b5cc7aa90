#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in clk clkqt; do for p in split f16; do
  echo "== $v $p"; RMNET_BANK_PRECISION=$p RMNET_HIP_LIB=build/variants/lib_$v.so timeout 120 python tools/dense_clk.py 1 2>&1 | grep "bk_main\|compute WGs\|^  " | head -6
done; done
