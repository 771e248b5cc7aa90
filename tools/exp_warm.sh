#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in $VARIANTS; do echo "== $v"; DIST=1 FIND=1 RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_$v.so timeout 600 python tools/loop_clk.py 40 2>&1 | grep "in-loop\|per launch\|compute WGs\|set-aside\|tickets served" | sed 's/tickets \[.*\]/tickets [...]/'; done
