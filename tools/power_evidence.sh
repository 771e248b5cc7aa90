#!/bin/bash
# Raw evidence for DESIGN.md section 5 ("what bounds bk_main"): socket power under load, in-kernel shader clock,
# the pure-MFMA ceiling on zero / random operands, and the energy-share table from ablated builds.
# Needs build/variants/lib_{clk,a1,a2,a4,a16,a17,a256,a1024,f16a1,...,f16tr,f16tr8}.so (tools/build_variants_all.sh)
# and tools/ubench/mfma_power.
# Output: gpurun_out/r03_power/raw.txt (copied, with the command lines, into profiles/r03_power_ceiling.md).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/r03_power
mkdir -p $out
{
echo "### kernel source hash (bench.source_hash): $(python -c 'import bench; print(bench.source_hash())')"
echo "### date: $(date -u)"
echo
echo "### \$ tools/power_probe.sh"
bash tools/power_probe.sh 2>&1
echo
echo "### \$ tools/ubench/mfma_power     (all 256 CUs, register operands; data = zero / random / hi-lo pairs)"
timeout 300 tools/ubench/mfma_power 2>&1
echo
W="8 21 36 21 36 5"
for args in "1 10 10 10 10 5" "$W" "8 21 36 21 36 10"; do
  echo "### \$ RMNET_HIP_LIB=build/variants/lib_clk.so python tools/bk_clk.py $args     (-DBK_CLK=1 build)"
  RMNET_HIP_LIB=build/variants/lib_clk.so python tools/bk_clk.py $args 2>&1 | tail -2
done
echo
echo "### energy shares: the bench-shaped launch (8 objects, 21x36 query box = 12 query tiles, 21x36 memory box x T=5 = 120 tiles)"
echo "### with parts of bk_main compiled out (BK_ABLATE bit mask: 1 no V reloads, 2 no PV MFMAs, 4 no S/soft-max,"
echo "### 16 no K tile loads, 256 no static part (q_val half / masked cells), 1024 publish + ticket but nobody merges);"
echo "### every variant also carries -DBK_CLK=1 so the clock is read in the same launch"
for v in clk a1 a2 a4 a16 a17 a256 a1024; do
  [ -f build/variants/lib_$v.so ] || continue
  echo "### \$ RMNET_HIP_LIB=build/variants/lib_$v.so python tools/chunk_bench.py $W ; ... bk_clk.py $W"
  RMNET_HIP_LIB=build/variants/lib_$v.so python tools/chunk_bench.py $W 2>&1 | tail -1
  RMNET_HIP_LIB=build/variants/lib_$v.so python tools/bk_clk.py $W 2>&1 | tail -1
done
echo
echo "### fp16-operand mode (RMNET_BANK_PRECISION=f16 -> RMNET_BANK_F16): the same launch and a launch with per-object random boxes as bench.py's"
echo "### clips have them; ablations of the fp16 loops (BK_ABLATE: 1 no V loads, 2 no PV MFMAs, 4 no S / soft-max, 16 no K ring, 17 = 1 + 16,"
echo "### 32 P fragments read once, 256 no static part); per-phase cycle stamps (-DBK_TRACE=1: producer wave 0 / consumer wave 4, wave 8)"
export RMNET_BANK_PRECISION=f16
for WW in "$W" "8 0 0 0 0 5"; do
  for v in clk f16a1 f16a2 f16a4 f16a16 f16a17 f16a32 f16a256; do
    [ -f build/variants/lib_$v.so ] || continue
    echo "### \$ RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_$v.so python tools/chunk_bench.py $WW ; ... bk_clk.py $WW"
    RMNET_HIP_LIB=build/variants/lib_$v.so python tools/chunk_bench.py $WW 2>&1 | tail -1
    RMNET_HIP_LIB=build/variants/lib_$v.so python tools/bk_clk.py $WW 2>&1 | tail -1
  done
done
for v in f16tr f16tr8; do
  [ -f build/variants/lib_$v.so ] || continue
  echo "### \$ RMNET_BANK_PRECISION=f16 RMNET_HIP_LIB=build/variants/lib_$v.so python tools/bk_trace.py 0.46 40"
  RMNET_HIP_LIB=build/variants/lib_$v.so python tools/bk_trace.py 0.46 40 2>&1 | grep -v amdgpu.ids | tail -10
done
unset RMNET_BANK_PRECISION
} > $out/raw.txt 2>&1
tail -40 $out/raw.txt
