#!/bin/bash
# round 5, GPU call 5: direct output stores A/B, launch-size sensitivity (clips per GPU), kernel rows without a profiler attached
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bank_read_vs_oracle or static_half or bank_read_f16" > $O/pytest_base.txt 2>&1; tail -1 $O/pytest_base.txt
RMNET_HIP_LIB=build/variants/lib_direct.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bank_read_vs_oracle or static_half or bank_read_f16 or edge_rect" > $O/pytest_direct.txt 2>&1; tail -1 $O/pytest_direct.txt
for i in 1 2 3; do
  for v in base0 direct; do
    RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v: /" >> $O/loop.txt
  done
done
for v in base0 direct; do
  RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=split timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v: /" >> $O/loop.txt
  RMNET_HIP_LIB=build/variants/lib_$v.so RMNET_BANK_PRECISION=f16 timeout 300 python tools/chunk_bench.py 5 0 0 0 0 5 2>/dev/null | tail -1 | sed "s/^/$v cfg3 f16: /" >> $O/loop.txt
done
sort $O/loop.txt
timeout 900 python tools/kernel_rows.py 2>/dev/null > $O/kernel_rows.json; grep -A3 dropin $O/kernel_rows.json | head -12
for c in 8 9 10 12; do
  timeout 900 python bench.py --clips-per-gpu $c --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('clips/GPU $c: %.1f frames/s, bk_main %.2f us, frac %.4f' % (j['value'], r['avg_us'], r['frac']))" >> $O/clips.txt
done
cat $O/clips.txt
