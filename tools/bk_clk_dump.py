# -*- coding: utf-8 -*-
"""Per-workgroup phase stamps of bk_main (-DBK_CLK=1 build): who finishes its tile walk late?  Grouped by XCD
(blockIdx % 8) and by chunk kind.  python tools/bk_clk_dump.py <no> <q_h> <q_w> <m_h> <m_w> [T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rmnet_amd import ops, _lib
no, qh, qw, mh, mw = [int(x) for x in sys.argv[1:6]]
T = int(sys.argv[6]) if len(sys.argv) > 6 else 5
h, w = 30, 54
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6).to(dev)
mv = torch.randn(no, 512, T, h, w, generator=g).to(dev)
qk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev)
qv = torch.randn(no, 512, h, w, generator=g).to(dev)
qr = torch.tensor([(2, 2 + qw - 1, 1, 1 + qh - 1)] * no, dtype=torch.int32, device=dev)
mr = torch.tensor([(3, 3 + mw - 1, 2, 2 + mh - 1)] * no, dtype=torch.int32, device=dev)
if qh == 0:        # per-object random boxes as bench.py's clips have them (tools/comb_bench.py), the same box for memory and query
    rng = np.random.RandomState(1)
    rects = []
    for o in range(no):
        rh, rw = rng.randint(17, 24), rng.randint(32, 42)
        y0, x0 = rng.randint(0, h - rh + 1), rng.randint(0, w - rw + 1)
        rects.append((x0, x0 + rw - 1, y0, y0 + rh - 1))
    qr = mr = torch.tensor(rects, dtype=torch.int32, device=dev)
bank = ops.MemoryBank(no, T, h, w, dev, precision=os.environ.get('RMNET_BANK_PRECISION', 'split'))
for t in range(T):
    bank.append(t, mk[:, :, t].contiguous(), mv[:, :, t].contiguous(), mr)
lib = _lib.load()
nb = lib.rmnet_bank_read_workspace_bytes(no, h, w)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
for _ in range(20):
    bank.read(T, qk, qv, qr, ws=ws)
torch.cuda.synchronize()
plan_end = (no * 12 * 4)
raw = ws[nb - 16384 - ((plan_end + 255) // 256 * 256) + plan_end + 64:][:256 * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
plan = ws[nb - 16384 - ((plan_end + 255) // 256 * 256):][:plan_end].view(torch.int32).cpu().numpy().reshape(no, 12)
print('plan record of object 0: Mq %d nqt %d njt %d M %d ... first slot %d C %d' % (plan[0, 0], plan[0, 1], plan[0, 2], plan[0, 3], plan[0, 8], plan[0, 9]))
us = raw / 100.0
walk = us[:, 6] - us[:, 5]
comp = us[:, 6] > 0
print('tile walk duration (us) by XCD (blockIdx % 8): ' + '  '.join('%d: med %.1f max %.1f' % (x, np.median(walk[comp & (np.arange(256) % 8 == x)]), walk[comp & (np.arange(256) % 8 == x)].max()) for x in range(8)))
order = np.argsort(-us[:, 6])
print('latest 24 tile-walk ends: ' + ' '.join('b%d:%.1f' % (i, us[i, 6]) for i in order[:24]))
print('earliest 24 (of the computing ones): ' + ' '.join('b%d:%.1f' % (i, us[i, 6]) for i in order[::-1] if comp[i])[:400])
cyc = raw[:, 0] / (raw[:, 1] * 10.0 + 1e-9)
print('clock GHz by XCD: ' + '  '.join('%d: %.2f' % (x, np.median(cyc[comp & (np.arange(256) % 8 == x)])) for x in range(8)))
