# -*- coding: utf-8 -*-
"""Assemble profiles/r03_power_ceiling.md: reading guide + Part A (gpurun_out/r03_final/power_raw.txt, the current
kernel) + Parts B / C kept from the file as committed (round-2 kernel run, PMC calibration).
    python tools/make_power_md.py"""
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, 'profiles', 'r03_power_ceiling.md')
old = open(path).read()
if '## Part B' in old:
    tail = old[old.index('## Part B'):]
else:
    raw_old = old[old.index('---\n') + 4:]
    tail = ('## Part B — round-2 kernel sources (bk_main + mr_combine), same script, earlier in the round\n' +
            raw_old.replace('## FETCH_SIZE / WRITE_SIZE calibration', '## Part C — FETCH_SIZE / WRITE_SIZE calibration'))
new_raw = open(os.path.join(root, 'gpurun_out', 'r03_final', 'power_raw.txt')).read()
guide = open(os.path.join(root, 'tools', 'power_guide.md')).read()
fence = '`' * 3
open(path, 'w').write(guide + '\n## Part A — round-3 kernel (one launch per read)\n\n' + fence + '\n' + new_raw + fence + '\n\n' + tail)
print('wrote', path)
