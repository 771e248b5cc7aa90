# -*- coding: utf-8 -*-
"""Per-phase cycle stamps of bk_main (needs a library built with -DBK_TRACE=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rmnet_amd import ops, _lib
no, h, w = 1, 30, 54
T = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6).to(dev)
mv = torch.randn(no, 512, T, h, w, generator=g).to(dev)
qk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev)
qv = torch.randn(no, 512, h, w, generator=g).to(dev)
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rh, rw = max(1, int(round(h * frac ** 0.5))), max(1, int(round(w * frac ** 0.5)))
y0, x0 = (h - rh) // 2, (w - rw) // 2
rect = torch.tensor([[x0, x0 + rw - 1, y0, y0 + rh - 1]] * no, dtype=torch.int32, device=dev)
bank = ops.MemoryBank(no, T, h, w, dev, precision=os.environ.get('RMNET_BANK_PRECISION', 'split'))
for t in range(T):
    bank.append(t, mk[:, :, t].contiguous(), mv[:, :, t].contiguous(), rect)
lib = _lib.load()
nb = lib.rmnet_bank_read_workspace_bytes(no, h, w)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
for _ in range(3):
    bank.read(T, qk, qv, rect, ws=ws)
torch.cuda.synchronize()
slots = 256 + no * ((h * w + 1 + 63) // 64)      # bank_total_slots(): the stamps live in the last slot
off = (slots - 1) * 512 * 64 * 4
tr = ws[off:off + 2048 * 8].view(torch.int64).cpu().numpy()
f16 = os.environ.get('RMNET_BANK_PRECISION') == 'f16'
plabels = (['top->head (mask)', 'S MFMAs + soft-max', 'K frags', 'barrier', '->next top'] if f16 else
           ['top->MFMAs issued', 'soft-max', 'K frags', 'barrier', '->next top'])
for name, a, labels in (('producer wave0', tr[:1024], plabels),
                        ('consumer wave (BK_TRACE_WAVE)', tr[1024:2048],
                         ['top->K ring fed', '->dt0 done', 'dt1', 'dt2', 'dt3', 'P frags + V addresses', 'barrier', '->next top']
                         if os.environ.get('TRACE2') == '1' else ['top->K data there', 'K stored', 'K requested', '->dt0 done', 'dt1', 'dt2', 'dt3', 'P frags + V addresses', 'barrier', '->next top'] if os.environ.get('TRACE2') == '2' else ['top->PV done', 'barrier', '->next top'])):
    a = a[a != 0]
    d = np.diff(a)
    print(name, 'n stamps', len(a), 'total', a[-1] - a[0])
    print('  first deltas', d[:14])
    per = len(labels)
    nfull = min(40, (len(d) - 6) // per)
    if nfull < 1:
        continue
    body = d[3:3 + per * nfull].reshape(-1, per)
    print('  labels:', labels)
    print('  per-iteration deltas (mean over its):', body.mean(0).round(0), 'sum', body.mean(0).sum())
    print('  tail deltas', d[-4:])
