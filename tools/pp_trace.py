# -*- coding: utf-8 -*-
"""Interval time line of the ping-pong walk (a -DBK_PPTRACE=1 build: waves 0 / 4 / 8 of workgroup 0 stamp s_memtime when they arrive
at and leave every barrier of the walk; the stamps land in partial slot 200 of the read's workspace, which a launch of
single-segment pairs never touches).
    RMNET_HIP_LIB=build/variants/lib_trace.so RMNET_BANK_PRECISION=f16 python tools/pp_trace.py [no q_h q_w m_h m_w T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rmnet_amd import ops, _lib
arg = [int(x) for x in sys.argv[1:7]] if len(sys.argv) > 6 else [16, 21, 36, 21, 36, 5]
no, qh, qw, mh, mw, T = arg
h, w = 30, 54
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6).to(dev)
mv = torch.randn(no, 512, T, h, w, generator=g).to(dev)
qk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev)
qv = torch.randn(no, 512, h, w, generator=g).to(dev)
qr = torch.tensor([(2, 2 + qw - 1, 1, 1 + qh - 1)] * no, dtype=torch.int32, device=dev)
mr = torch.tensor([(3, 3 + mw - 1, 2, 2 + mh - 1)] * no, dtype=torch.int32, device=dev)
prec = os.environ.get('RMNET_BANK_PRECISION', 'f16')
bank = ops.MemoryBank(no, T, h, w, dev, precision=prec)
for t in range(T):
    bank.append(t, mk[:, :, t].contiguous(), mv[:, :, t].contiguous(), mr)
lib = _lib.load()
ws = torch.zeros(lib.rmnet_bank_read_workspace_bytes_for(no, h, w, T), dtype=torch.uint8, device=dev)
for _ in range(5):
    bank.read(T, qk, qv, qr, ws=ws)
torch.cuda.synchronize()
off = 200 * 32768 * 4
raw = ws[off:off + 3 * 1024 * 8].view(torch.int64).cpu().numpy().reshape(3, 1024)
names = ['producer w0', 'consumer A w4', 'consumer B w8']
nst = T * ((mh * mw + 31) // 32) // 2
print('precision %s, %d steps expected, clock stamps in shader cycles (s_memtime)' % (prec, nst))
rows = {}
for r in range(3):
    t = raw[r]
    n = int(np.count_nonzero(t))
    t = t[:n].astype(np.float64)
    if n < 3:
        print(names[r], 'no stamps'); continue
    t0 = t[0]
    arrive = t[1::2][: (n - 1) // 2]
    depart = t[2::2][: (n - 1) // 2]
    prev = np.concatenate(([t0], depart[:-1]))
    busy = arrive - prev
    wait = depart - arrive
    rows[r] = (busy, wait)
    k = len(busy)
    mid = slice(8, max(9, k - 8))
    print('%-14s intervals %3d  total %8.0f cycles | busy mean %6.0f (even %6.0f, odd %6.0f) | wait mean %6.0f (even %6.0f, odd %6.0f)' % (
        names[r], k, depart[-1] - t0, busy[mid].mean(), busy[mid][0::2].mean(), busy[mid][1::2].mean(),
        wait[mid].mean(), wait[mid][0::2].mean(), wait[mid][1::2].mean()))
if len(rows) == 3:
    k = min(len(rows[r][0]) for r in rows)
    print('interval: busy/wait producer | A | B   (first 24 and a middle stretch)')
    for i in list(range(0, min(24, k))) + list(range(60, min(72, k))):
        print('  %3d %s: P %5.0f/%5.0f  A %5.0f/%5.0f  B %5.0f/%5.0f' % (i, 'E' if i % 2 == 0 else 'O', rows[0][0][i], rows[0][1][i],
                                                                        rows[1][0][i], rows[1][1][i], rows[2][0][i], rows[2][1][i]))
