# -*- coding: utf-8 -*-
"""Average FETCH_SIZE / WRITE_SIZE per launch from the rows tools/pmc_traffic.sh extracted, and the traffic
file bench.py reads (profiles/bk_main_hbm_traffic.json, stamped with the kernel sources' hash).

    python tools/pmc_traffic.py gpurun_out/pmc [--write-profile]

Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB.  Calibration (round 3, tools/pmc_calib.sh, raw numbers in
profiles/r03_power_ceiling.md): on a launch that fetches every bank byte exactly once bk_main's FETCH_SIZE reads
10,962 KB against 20.43 MB of known bytes, and a 1 GiB device copy reads 524,300 KB -- the 1/2 rule of
MI355X_MICROARCH.md holds for this kernel's 16-byte-per-lane loads -> FETCH_SIZE x 2; WRITE_SIZE 6,553.6 KB against
6,553.5 KB of partials (copy: 1,048,576 KB per GiB) -> x 1.  (Rounds 1-2 used x 1 for both: their single-object
"calibration" launch fetched every tile from two XCDs.)"""
FETCH_FACTOR, WRITE_FACTOR = 2.0, 1.0
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(path):
    out = {}
    if not os.path.exists(path):
        return out
    for row in csv.reader(open(path)):
        name = [c for c in row if 'rmnet::' in c]
        if not name:
            continue
        m = re.search(r'namespace\)::(\w+)', name[0])
        short = m.group(1) if m else name[0][:40]
        # columns: ..., Counter_Name, Counter_Value, Start, End
        val = float(row[-3])
        out.setdefault(short, []).append(val)
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


def main():
    d = sys.argv[1]
    fetch, write = per_kernel(os.path.join(d, 'FETCH_SIZE.csv')), per_kernel(os.path.join(d, 'WRITE_SIZE.csv'))
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, (0, 0)), write.get(k, (0, 0))
        print('%-28s fetch %10.1f KB (%d launches)   write %10.1f KB (%d launches)' % (k, f[0], f[1], w[0], w[1]))
    if '--write-profile' in sys.argv and 'bk_main' in fetch and 'bk_main' in write:
        import bench
        H, W, K, T = bench.H, bench.W, bench.K_CH, bench.T_MEM
        abytes = bench.algorithmic_bytes(bench.DEFAULT_CLIPS * (K - 1), T, 30, 54)
        prec = sys.argv[-1] if sys.argv[-1] in ('f16', 'qx') else 'split'
        prof = {'kernel': 'bk_main<%d>' % {'f16': 1, 'qx': 2, 'split': 3}[prec], 'read_precision': prec, 'source_hash': bench.source_hash(),
                'command': 'tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, separate pass) --kernel-trace -- '
                           'python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --read-precision %s (%d object-frames per launch)' % (prec, bench.DEFAULT_CLIPS),
                'fetch_size_kb_per_launch': round(fetch['bk_main'][0], 1), 'write_size_kb_per_launch': round(write['bk_main'][0], 1),
                'hbm_bytes_per_launch': int(1024 * (FETCH_FACTOR * fetch['bk_main'][0] + WRITE_FACTOR * write['bk_main'][0])),
                'fetch_factor': FETCH_FACTOR, 'write_factor': WRITE_FACTOR,
                'algorithmic_bytes_per_launch': abytes,
                'calibration': __doc__.split('Units:')[1].strip(),
                'other_kernels_same_run': {k: {'fetch_size_kb_per_launch': round(fetch.get(k, (0, 0))[0], 1),
                                               'write_size_kb_per_launch': round(write.get(k, (0, 0))[0], 1)}
                                           for k in sorted(set(fetch) | set(write)) if k != 'bk_main'}}
        fname = {'f16': 'bk_main_f16_hbm_traffic.json', 'qx': 'bk_main_qx_hbm_traffic.json', 'split': 'bk_main_hbm_traffic.json'}[prec]
        json.dump(prof, open(os.path.join(ROOT, 'profiles', fname), 'w'), indent=1)
        print('wrote profiles/' + fname)


if __name__ == '__main__':
    main()
