#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
for c in 12 14 16 20 24 32; do
  timeout 900 python bench.py --clips-per-gpu $c --steps 16 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('clips/GPU $c: %.1f frames/s, %.2f ms/step, bk_main %.2f us, frac %.4f' % (j['value'], j['ms_per_step'], r['avg_us'], r['frac']))" >> $O/clips.txt
done
cat $O/clips.txt
