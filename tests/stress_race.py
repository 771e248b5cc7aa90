# -*- coding: utf-8 -*-
"""Repeat-read stress of the bank kernels: the same small banks (short tile lists, empty frames, a late
logit spike -- the shapes where a K-ring / P-buffer race shows up) are read many times and every read is
compared with the oracle.  Not collected by pytest: python tests/stress_race.py [reads per case]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rmnet_amd import ops
from oracle import oracle
# RMNET_BANK_PRECISION=f16 runs the fp16-operand mode: its bar is 2^-10 of the largest value (tests/test_gpu_parity.py)
ATOL = 5e-3 if os.environ.get('RMNET_BANK_PRECISION') in ('f16', 'qx') else 3e-5
dev = torch.device('cuda', 0)
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad_total = 0
for case, (no, T, h, w, seed) in enumerate([(2, 3, 6, 10, 5), (1, 5, 8, 8, 1), (3, 2, 12, 20, 2), (2, 9, 10, 7, 3),
                                            (1, 1, 4, 5, 4), (5, 3, 9, 13, 6), (8, 5, 30, 54, 7)]):
    rng = np.random.RandomState(seed)
    mk = (rng.randn(no, 128, T, h, w) * 0.6).astype(np.float32)
    mv = rng.randn(no, 512, T, h, w).astype(np.float32)
    qk = (rng.randn(no, 128, h, w) * 0.6).astype(np.float32)
    qv = rng.randn(no, 512, h, w).astype(np.float32)
    def rect():
        if rng.rand() < 0.2:
            return (1, 0, 1, 0)
        x0, y0 = rng.randint(0, w), rng.randint(0, h)
        return (x0, rng.randint(x0, w), y0, rng.randint(y0, h))
    mr = np.array([[rect() for _ in range(T)] for _ in range(no)], np.int32)
    mr[:, T - 1] = (0, w - 1, 0, h - 1)
    qr = np.array([(0, w - 1, 0, h - 1)] * no, np.int32)
    if case % 2 == 1:                       # every second case: regional queries (static part + merge in one launch)
        qr = np.array([rect() for _ in range(no)], np.int32)
    mk[:, :, T - 1, h - 1, w - 1] = qk[:, :, min(2, h - 1), min(3, w - 1)] * 9.0     # late spike
    want, _ = oracle.regional_memory_read(mk, mv, qk, qv, mr, qr)
    bank = ops.MemoryBank(no, T + 1, h, w, dev, precision=os.environ.get('RMNET_BANK_PRECISION', 'split'))
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), cu(mr[:, t]))
    qk_d, qv_d, qr_d = cu(qk), cu(qv), cu(qr)
    n_here = reads if no < 8 else max(reads // 10, 5)
    bad = 0
    for r in range(n_here):
        got = bank.read(T, qk_d, qv_d, qr_d).cpu().numpy()
        if not np.allclose(got, want, atol=ATOL, rtol=2e-5):
            bad += 1
    print('case %d %s: %d / %d reads wrong' % (case, (no, T, h, w), bad, n_here))
    bad_total += bad
print('RACE-FREE' if bad_total == 0 else 'FAILURES: %d' % bad_total)
sys.exit(1 if bad_total else 0)
