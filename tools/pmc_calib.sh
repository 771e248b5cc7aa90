#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate --pmc passes, kernel-trace only) of the known-byte-count launches of
# tools/pmc_calib.py -> gpurun_out/pmc_calib/{FETCH_SIZE,WRITE_SIZE}.csv + summary.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_calib
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal_$c -- python $ROOT/tools/pmc_calib.py > /tmp/cal_$c.log 2>&1 || true
  f=$(find /tmp/cal_$c -name "*counter_collection.csv" | head -1)
  head -1 "$f" > $ROOT/gpurun_out/pmc_calib/$c.csv
  grep -E "bk_main|mr_combine|copy|Copy|elementwise" "$f" >> $ROOT/gpurun_out/pmc_calib/$c.csv || true
  tail -1 /tmp/cal_$c.log
done
python - <<PY > $ROOT/gpurun_out/pmc_calib/summary.txt
import csv, collections
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = list(csv.reader(open('$ROOT/gpurun_out/pmc_calib/%s.csv' % c)))
    hdr, rows = rows[0], rows[1:]
    ki, vi = hdr.index('Kernel_Name'), hdr.index('Counter_Value')
    acc = collections.OrderedDict()
    for r in rows:
        acc.setdefault(r[ki][:70], []).append(float(r[vi]))
    for k, v in acc.items():
        print('%-11s %-70s n=%2d  last-4 mean %12.1f KB  (all: %s)' % (c, k, len(v), sum(v[-4:]) / len(v[-4:]), ' '.join('%.0f' % x for x in v)))
PY
cat $ROOT/gpurun_out/pmc_calib/summary.txt
