# -*- coding: utf-8 -*-
"""Average rocprofv3 --pmc counter values per kernel from counter_collection csv files.
    python tools/pmc_report.py <dir> <kernel-substring>"""
import csv, glob, sys, collections
d, key = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if key in row['Kernel_Name']:
            a = acc[row['Counter_Name']]
            a[0] += float(row['Counter_Value']); a[1] += 1
for k in sorted(acc):
    print('%-32s avg %16.1f  (n=%d)' % (k, acc[k][0] / acc[k][1], acc[k][1]))
