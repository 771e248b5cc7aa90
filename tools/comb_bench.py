# -*- coding: utf-8 -*-
"""mr_combine / bk_main timing at n object-frames per launch with per-object random boxes (as bench.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rmnet_amd import ops
from bench import HipEvents
no = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, h, w = 5, 30, 54
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
rng = np.random.RandomState(1)
mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6).to(dev)
mv = torch.randn(no, 512, T, h, w, generator=g).to(dev)
qk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev)
qv = torch.randn(no, 512, h, w, generator=g).to(dev)
rects = []
for o in range(no):
    rh, rw = rng.randint(17, 24), rng.randint(32, 42)
    y0, x0 = rng.randint(0, h - rh + 1), rng.randint(0, w - rw + 1)
    rects.append((x0, x0 + rw - 1, y0, y0 + rh - 1))
qr = torch.tensor(rects, dtype=torch.int32, device=dev)
bank = ops.MemoryBank(no, T, h, w, dev)
for t in range(T):
    bank.append(t, mk[:, :, t].contiguous(), mv[:, :, t].contiguous(), qr)
reps = 40
ev = HipEvents(3 * reps)
for _ in range(5):
    bank.read(T, qk, qv, qr)
torch.cuda.synchronize()
for i in range(reps):
    bank.read(T, qk, qv, qr, events=tuple(ev.ev[3 * i:3 * i + 3]))
torch.cuda.synchronize()
bm = [ev.elapsed_ms(ev.ev[3 * i], ev.ev[3 * i + 1]) * 1e3 for i in range(reps)]
bc = [ev.elapsed_ms(ev.ev[3 * i + 1], ev.ev[3 * i + 2]) * 1e3 for i in range(reps)]
print('no=%d  bk_main avg %.2f min %.2f us | combine avg %.2f min %.2f us' % (no, np.mean(bm), np.min(bm), np.mean(bc), np.min(bc)))
