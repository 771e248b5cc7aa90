#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
EPILOGUE=1 FIND=1 RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_clkep.so timeout 600 python tools/loop_clk.py 20 2>&1 | grep -v "^$" | tail -30
