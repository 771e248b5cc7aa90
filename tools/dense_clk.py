# -*- coding: utf-8 -*-
"""Time line of ONE bank read of `no` dense (no boxes) 480p objects at T = 5 -- the launch the drop-in entry makes for one
object.  With a -DBK_CLK=1 library (RMNET_HIP_LIB) the per-workgroup stamps are printed.   python tools/dense_clk.py [no]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rmnet_amd import ops, _lib
no = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T, h, w = 5, 30, 54
prec = os.environ.get('RMNET_BANK_PRECISION', 'split')
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6).to(dev)
mv = torch.randn(no, 512, T, h, w, generator=g).to(dev)
qk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev)
qv = torch.randn(no, 512, h, w, generator=g).to(dev)
bank = ops.MemoryBank(no, T, h, w, dev, precision=prec)
for t in range(T):
    bank.append(t, mk[:, :, t].contiguous(), mv[:, :, t].contiguous(), None)
lib = _lib.load()
nb = lib.rmnet_bank_read_workspace_bytes(no, h, w)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
reps = 20
ev = bench.HipEvents(3 * reps)
floor = ev.floor_us(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(5):
    bank.read(T, qk, qv, None, ws=ws)
torch.cuda.synchronize()
for i in range(reps):
    bank.read(T, qk, qv, None, ws=ws, events=tuple(ev.ev[3 * i:3 * i + 3]))
torch.cuda.synchronize()
us_ = [ev.elapsed_ms(ev.ev[3 * i], ev.ev[3 * i + 1]) * 1e3 - floor for i in range(reps)]
print('dense no=%d T=%d %s: bk_main avg %.2f min %.2f max %.2f us' % (no, T, prec, np.mean(us_), np.min(us_), np.max(us_)))
plan_end = no * 12 * 4
clk = os.environ.get('RMNET_HIP_LIB', '').find('clk') >= 0
base = nb - (16384 if clk else 0) - ((plan_end + 255) // 256 * 256)
plan = ws[base:][:plan_end].view(torch.int32).cpu().numpy().reshape(no, 12)
print('plan records (Mq nqt njt M rect.. slot C):'); print(plan)
if clk:
    raw = ws[base + plan_end + 64:][:256 * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
    us = raw / 100.0
    comp = us[:, 6] > 0
    def st(col, sel):
        v = us[sel, col]
        return '%.1f [%.1f..%.1f]' % (np.median(v), v.min(), v.max())
    print('plan inputs in LDS %s' % st(7, comp))
    print('compute WGs (%d): plan done %s; first walk starts %s; last walk over %s; compute part over %s; left %s'
          % (comp.sum(), st(4, comp), st(5, comp), st(6, comp), st(1, comp), st(3, comp)))
    aside = ~comp & (us[:, 3] > 0)
    if aside.any():
        print('set-aside WGs (%d): left %s' % (aside.sum(), st(3, aside)))
    print('tickets served by compute WGs: %d' % raw[comp, 2].sum())
    o = np.argsort(us[comp, 1])
    print('compute part over, sorted (us): ' + ' '.join('%.0f' % x for x in us[comp, 1][o][::8]))
    print('walk length (us) sorted: ' + ' '.join('%.0f' % x for x in np.sort((us[:, 6] - us[:, 5])[comp])[::8]))
    top = np.argsort(-us[:, 3])[:12]
    print('last to leave: (wg, walk start, walk over, compute part over, tickets, left)')
    for i in top:
        print('  %3d  %.1f  %.1f  %.1f  %d  %.1f' % (i, us[i, 5], us[i, 6], us[i, 1], raw[i, 2], us[i, 3]))
    tk = raw[:, 2] > 0
    print('workgroups that served tickets: ' + ' '.join('%d:%d(%.0f->%.0f)' % (i, raw[i, 2], us[i, 1], us[i, 3]) for i in np.nonzero(tk)[0]))
