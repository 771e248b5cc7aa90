#!/bin/bash
# The default bench line (launch sizes, modes, single stream) with the tree's library and with build/variants/lib_old.so, interleaved:
#     gpurun -- 'tools/gpurun_call.sh <tag> cmd bash tools/bench_ab.sh <tag>'
out=gpurun_out/$1
for rep in $(seq 1 ${REPS:-2}); do
  for lib in rmnet_amd/librmnet_hip.so build/variants/lib_old.so; do
    [ -f $lib ] || continue
    n=$(basename $lib .so)_$rep
    RMNET_HIP_LIB=$PWD/$lib timeout 900 python bench.py --no-cpu-baseline > $out/bench_$n.json 2> $out/bench_$n.err
    echo "== $lib rep $rep"; python tools/show_line.py $out/bench_$n.json | head -14
  done
done
