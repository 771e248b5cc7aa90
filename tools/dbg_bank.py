# -*- coding: utf-8 -*-
"""Where does a bank read differ from the oracle?  Splits the error by output region: read-out of the cells inside
the query box (written by the last arriver of a pair), read-out of masked cells (static part: column-sum mean),
the q_val half (static part).  python tools/dbg_bank.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rmnet_amd import ops
from oracle import oracle

dev = torch.device('cuda', 0)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cases = [(1, 1, 4, 5, 0), (2, 3, 9, 13, 1), (1, 5, 30, 54, 2), (5, 3, 30, 54, 3), (8, 5, 30, 54, 4), (14, 2, 6, 9, 5),
         (70, 1, 4, 5, 6), (3, 20, 12, 20, 7), (1, 70, 5, 6, 8)]
worst = 0.0
for no, T, h, w, seed in cases:
    rng = np.random.RandomState(seed)
    mk = (rng.randn(no, 128, T, h, w) * 0.6).astype(np.float32)
    mv = rng.randn(no, 512, T, h, w).astype(np.float32)
    qk = (rng.randn(no, 128, h, w) * 0.6).astype(np.float32)
    qv = rng.randn(no, 512, h, w).astype(np.float32)

    def rect(p_empty=0.1):
        if rng.rand() < p_empty:
            return (1, 0, 1, 0)
        x0, y0 = rng.randint(0, w), rng.randint(0, h)
        return (x0, rng.randint(x0, w), y0, rng.randint(y0, h))
    mr = np.array([[rect() for _ in range(T)] for _ in range(no)], np.int32)
    qr = np.array([rect(0.05) for _ in range(no)], np.int32)
    if no >= 2:
        qr[0] = (0, w - 1, 0, h - 1)
        mr[1, :] = (1, 0, 1, 0)                    # an object with nothing memorised inside its boxes
    want, _ = oracle.regional_memory_read(mk, mv, qk, qv, mr, qr)
    bank = ops.MemoryBank(no, T + 1, h, w, dev, precision=os.environ.get('RMNET_BANK_PRECISION', 'split'))
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), cu(mr[:, t]))
    got = bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy()
    inbox = np.zeros((no, 1, h, w), bool)
    for o in range(no):
        x0, x1, y0, y1 = qr[o]
        if x0 <= x1 and y0 <= y1:
            inbox[o, 0, y0:y1 + 1, x0:x1 + 1] = True
    err = np.abs(got - want)
    e_in = float((err[:, :512] * inbox).max())
    e_out = float((err[:, :512] * ~inbox).max())
    e_q = float(err[:, 512:].max())
    nan = int(np.isnan(got).sum())
    worst = max(worst, e_in, e_out, e_q) if not nan else float('inf')
    print('no=%2d T=%2d %2dx%2d: in-box %.2e  masked %.2e  q_val half %.2e  NaN %d' % (no, T, h, w, e_in, e_out, e_q, nan))
    if e_in > 3e-5 or nan:
        o, d, y, x = np.unravel_index(np.nanargmax(np.where(np.isnan(err[:, :512]), np.inf, err[:, :512]) * inbox), err[:, :512].shape)
        print('    worst in-box at object %d channel %d cell (%d,%d): got %.6f want %.6f ; query box %s' % (o, d, y, x, got[o, d, y, x], want[o, d, y, x], qr[o]))
print('WORST %.3e %s' % (worst, 'OK' if worst < 3e-5 else 'MISMATCH'))
