# -*- coding: utf-8 -*-
"""Print the fields of a bench.py JSON line that a round's bookkeeping needs:  python tools/show_line.py <file>"""
import json, sys
l = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
r = l['roofline']
print('value %.2f %s | ms/step %.3f | bk_main %.2f us frac %.4f traffic %s' % (l['value'], l['unit'], l['ms_per_step'], r['avg_us'], r['frac'], r.get('traffic')))
print('dtype:', l['dtype'][:160], '...')
for k, v in (r.get('launch_sizes') or {}).items():
    if k != 'note':
        print('  launch size %2s: %s' % (k, {a: b for a, b in v.items() if a != 'object_frames'}))
for k, v in (r.get('modes') or {}).items():
    if k != 'note':
        print('  mode %-5s: %s' % (k, v))
print('auto_rule:', {k: v for k, v in (r.get('auto_rule') or {}).items() if k != 'note'})
c = l.get('cpu_baseline')
if c:
    print('cpu_baseline: %.3f %s at %d threads (8 threads %.3f); configs0 %s' % (c['value'], c['unit'], c['cores'], c.get('value_8_threads', 0), (c.get('configs0') or {}).get('sweep')))
e = l.get('extras') or {}
print('extras keys:', sorted(e)[:30])
for k in ('single_stream_fps', 'single_stream_graph_fps', 'timed_region_gpu_time_share'):
    if k in e:
        print(' ', k, e[k])
if 'multi_gpu' in l:
    print('multi_gpu:', {k: v for k, v in l['multi_gpu'].items() if k != 'note'})
