#!/bin/bash
# HBM traffic of bk_main / mr_combine at the bench workload, the way MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), kernel-trace only.
# Writes gpurun_out/pmc/{fetch,write}/ and prints the per-launch averages; tools/pmc_traffic.py turns
# them into profiles/bk_main_hbm_traffic.json (stamped with bench.source_hash()).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
export TMPDIR=/tmp
# PRECISION=f16 bash tools/pmc_traffic.sh does the same for the fp16-operand mode -> gpurun_out/pmc_f16/, profiles/bk_main_f16_hbm_traffic.json
PREC=${PRECISION:-split}
out=gpurun_out/pmc; [ "$PREC" != split ] && out=gpurun_out/pmc_$PREC
mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --read-precision $PREC > /tmp/pmc_$c.log 2>&1 || true
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  grep -E "bk_main|mr_combine|bk_append|region_reduce|region_fill|flow_affine|soft_aggregate" "$f" > $ROOT/$out/$c.csv || true
  tail -1 /tmp/pmc_$c.log | cut -c1-300
done
name=bk_main_hbm_traffic.json; [ "$PREC" != split ] && name=bk_main_${PREC}_hbm_traffic.json
python $ROOT/tools/pmc_traffic.py $ROOT/$out --write-profile $PREC && cp $ROOT/profiles/$name $ROOT/$out/
