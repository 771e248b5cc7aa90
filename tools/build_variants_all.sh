#!/bin/bash
# All experiment builds tools/power_evidence.sh reads (build/variants/, git-ignored; remove the directory before a round ends).
cd "$(dirname "$0")/.."
rm -rf build/variants
export PATCH=tools/patches/bank_experiment_switches.patch   # [r5] the switches live in the patch, not in the product source
bash tools/build_variant.sh clk -DBK_CLK=1 > /dev/null &
for a in 1 2 4 16 17 256 1024; do bash tools/build_variant.sh a$a -DBK_CLK=1 -DBK_ABLATE=$a > /dev/null & done
wait
for a in 1 2 4 16 17 32 256; do bash tools/build_variant.sh f16a$a -DBK_CLK=1 -DBK_ABLATE=$a > /dev/null & done
bash tools/build_variant.sh f16tr -DBK_TRACE=1 > /dev/null &
bash tools/build_variant.sh f16tr8 -DBK_TRACE=1 -DBK_TRACE_WAVE=8 > /dev/null &
wait
ls build/variants
