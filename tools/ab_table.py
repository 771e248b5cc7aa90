"""Tabulates a `tools/gpurun_call.sh <tag> walk_ab` log: bk_main avg (min) us per mode / shape, tree library vs build/variants/lib_*.so.
    python tools/ab_table.py gpurun_out/<tag>/log.txt"""
import re, sys
rows, cur, libs = {}, None, []
for l in open(sys.argv[1]):
    m = re.match(r'### (\S+) (\S+) (FLUSH=512 )?chunk_bench (.*)', l)
    if m:
        lib = re.sub(r'.*/lib_?|\.so', '', m.group(2))
        if lib not in libs: libs.append(lib)
        cur = (m.group(1), m.group(4).strip(), 'cold' if m.group(3) else 'warm', lib)
        continue
    m = re.search(r'bk_main avg ([\d.]+) min ([\d.]+)', l)
    if m and cur:
        rows[cur] = (float(m.group(1)), float(m.group(2)))
        cur = None
shapes = []
for k in rows:
    if (k[0], k[1]) not in shapes: shapes.append((k[0], k[1]))
print('mode  shape              | ' + ' | '.join('%s: warm avg (min), cold avg' % x for x in libs))
for mode, sh in shapes:
    cells = []
    for lib in libs:
        w, c = rows.get((mode, sh, 'warm', lib), (0, 0)), rows.get((mode, sh, 'cold', lib), (0, 0))
        cells.append('%7.1f (%6.1f) %7.1f' % (w[0], w[1], c[0]))
    print('%-5s %-18s | %s' % (mode, sh, ' | '.join(cells)))
