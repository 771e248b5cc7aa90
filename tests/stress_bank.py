# -*- coding: utf-8 -*-
"""Randomised stress of the bank read (launch-wide plan, fast and fallback paths, object groups) against
the oracle.  Not collected by pytest (takes minutes): python tests/stress_bank.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rmnet_amd import ops
from oracle import oracle
# RMNET_BANK_PRECISION=f16 runs the fp16-operand mode: its bar is 2^-10 of the largest value (tests/test_gpu_parity.py)
ATOL = 5e-3 if os.environ.get('RMNET_BANK_PRECISION') in ('f16', 'qx') else 3e-5
dev = torch.device('cuda', 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
worst = 0.0
for case in range(cases):
    no = int(rng.choice([1, 2, 3, 5, 8, 13, 20, 66]))
    T = int(rng.choice([1, 2, 3, 5, 9, 65])) if no < 20 else int(rng.choice([1, 2]))
    h, w = int(rng.randint(2, 13)), int(rng.randint(2, 15))
    if no * T * h * w > 60000:
        T = 1
    mk = (rng.randn(no, 128, T, h, w) * 0.7).astype(np.float32)
    mv = rng.randn(no, 512, T, h, w).astype(np.float32)
    qk = (rng.randn(no, 128, h, w) * 0.7).astype(np.float32)
    qv = rng.randn(no, 512, h, w).astype(np.float32)
    def rect(p_empty, p_full):
        u = rng.rand()
        if u < p_empty:
            return (1, 0, 1, 0)
        if u < p_empty + p_full:
            return (0, w - 1, 0, h - 1)
        x0, y0 = rng.randint(0, w), rng.randint(0, h)
        return (x0, rng.randint(x0, w), y0, rng.randint(y0, h))
    mr = np.array([[rect(0.2, 0.2) for _ in range(T)] for _ in range(no)], np.int32)
    qr = np.array([rect(0.1, 0.2) for _ in range(no)], np.int32)
    bank = ops.MemoryBank(no, T + int(rng.randint(0, 3)), h, w, dev, precision=os.environ.get('RMNET_BANK_PRECISION', 'split'))
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), cu(mr[:, t]))
    got = bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy()
    want, _ = oracle.regional_memory_read(mk, mv, qk, qv, mr, qr)
    err = float(np.abs(got - want).max())
    worst = max(worst, err)
    if not np.allclose(got, want, atol=ATOL, rtol=2e-5):
        print('MISMATCH case', case, (no, T, h, w), 'max err', err)
        sys.exit(1)
print('ok: %d cases, worst abs error %.3g' % (cases, worst))
