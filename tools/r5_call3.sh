#!/bin/bash
# round 5, GPU call 3: the tail changes (arrival ticket inside the last step, early look at the static queue) -- correctness + A/B in the loop
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bank_read or bank_full or repeatable or static_half or chunks or bank_edge or one_launch_limit or counter_out_of_range or golden_vectors or strides" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for p in split f16 qx; do
  RMNET_BANK_PRECISION=$p timeout 900 python tests/stress_bank.py 2>/dev/null | tail -2 | sed "s/^/stress_bank $p: /" >> $O/stress.txt
  RMNET_BANK_PRECISION=$p timeout 900 python tests/stress_race.py 2>/dev/null | tail -2 | sed "s/^/stress_race $p: /" >> $O/stress.txt
done
for i in 1 2 3; do
  for v in base r4tail noticket nopeek noeq; do
    lib=build/variants/lib_$v.so; [ $v = base ] && lib=rmnet_amd/librmnet_hip.so
    RMNET_HIP_LIB=$lib RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v: /" >> $O/loop.txt
  done
done
for v in base r4tail; do
  lib=build/variants/lib_$v.so; [ $v = base ] && lib=rmnet_amd/librmnet_hip.so
  RMNET_HIP_LIB=$lib RMNET_BANK_PRECISION=split timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v: /" >> $O/loop.txt
  RMNET_HIP_LIB=$lib RMNET_BANK_PRECISION=qx timeout 600 python tools/loop_clk.py 16 2>/dev/null | grep "in-loop" | sed "s/^/$v: /" >> $O/loop.txt
done
RMNET_HIP_LIB=build/variants/lib_clk.so RMNET_BANK_PRECISION=f16 timeout 600 python tools/loop_clk.py 16 2>/dev/null > $O/loop_clk_f16.txt
for v in base pf4 pf12; do
  lib=build/variants/lib_$v.so; [ $v = base ] && lib=rmnet_amd/librmnet_hip.so
  RMNET_HIP_LIB=$lib timeout 900 python tools/kernel_rows.py 2>/dev/null > $O/kernel_rows_$v.json
done
tail -3 $O/pytest.txt; cat $O/stress.txt; sort $O/loop.txt
