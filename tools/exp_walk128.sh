#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
for b in walk128 walk128_a1 walk128_a2 walk128_a4 walk128_a8 walk128_a15; do for p in 0 1; do echo -n "$b: "; timeout 60 ./$b 200 1 $p | head -1; done; done
