#!/bin/bash
# Raw evidence for DESIGN.md section 5 ("what bounds bk_main"): socket power under load, in-kernel shader clock,
# the pure-MFMA ceiling on zero / random operands, and the energy-share table from ablated builds.
# Needs build/variants/lib_{clk,a1,a2,a4,a16,a256,a1024}.so (tools/build_variant.sh <name> -DBK_CLK=1 -DBK_ABLATE=<mask>)
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
} > $out/raw.txt 2>&1
tail -40 $out/raw.txt
