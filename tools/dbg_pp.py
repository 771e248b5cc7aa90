# -*- coding: utf-8 -*-
"""Where does a bank read differ from the oracle?  (round 6: bring-up of the ping-pong walk)
    RMNET_BANK_PRECISION=qx python tools/dbg_pp.py [case] [reads]
Prints, per read, the worst error and which (object, query tile, channel block) holds the wrong values."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rmnet_amd import ops
from oracle import oracle
dev = torch.device('cuda', 0)
case = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cases = [(2, 3, 6, 10, 5), (1, 5, 8, 8, 1), (3, 2, 12, 20, 2), (2, 9, 10, 7, 3), (1, 1, 4, 5, 4), (5, 3, 9, 13, 6), (8, 5, 30, 54, 7)]
no, T, h, w, seed = cases[case]
TUSE = int(os.environ.get('TUSE', T))     # read only the LAST TUSE frames (fewer tiles: no split of the pairs)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
# the random stream of tests/stress_race.py up to this case
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
if case % 2 == 1:
    qr = np.array([rect() for _ in range(no)], np.int32)
if os.environ.get('NOSPIKE') != '1':
    mk[:, :, T - 1, h - 1, w - 1] = qk[:, :, min(2, h - 1), min(3, w - 1)] * 9.0
if TUSE < T:
    mk, mv, mr, T = mk[:, :, T - TUSE:], mv[:, :, T - TUSE:], mr[:, T - TUSE:], TUSE
want, _ = oracle.regional_memory_read(mk, mv, qk, qv, mr, qr)
prec = os.environ.get('RMNET_BANK_PRECISION', 'split')
bank = ops.MemoryBank(no, T + 1, h, w, dev, precision=prec)
for t in range(T):
    bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), cu(mr[:, t]))
qk_d, qv_d, qr_d = cu(qk), cu(qv), cu(qr)
print('lib', os.environ.get('RMNET_HIP_LIB', 'tree'), 'precision', prec, 'case', (no, T, h, w), 'areas', bank.areas()[:, :T].cpu().numpy().tolist())
for r in range(reads):
    got = bank.read(T, qk_d, qv_d, qr_d).cpu().numpy()
    err = np.abs(got[:, :512] - want[:, :512])              # [no, 512, h, w]
    print('read %d: max err %.3e mean %.3e; static half max err %.3e' % (r, err.max(), err.mean(), np.abs(got[:, 512:] - want[:, 512:]).max()))
    bad = err > 5e-3
    if bad.any() and os.environ.get('DEEP') == '1':
        mean_all = mv.reshape(no, 512, -1).mean(axis=2)            # [no, 512]: the read-out of a masked query cell
        for o in range(no):
            if not bad[o].any():
                continue
            for ch in np.nonzero(bad[o].any(axis=(1, 2)))[0]:
                g_ = got[o, ch].reshape(-1); w_ = want[o, ch].reshape(-1)
                cells = np.nonzero(np.abs(g_ - w_) > 1e-3)[0]           # every visibly wrong query of this channel row
                tiles = sorted(set((cells // 64).tolist()))
                for qt in tiles:
                    sel = np.arange(qt * 64, min(qt * 64 + 64, h * w))
                    d = g_[sel] - w_[sel]
                    wrong = np.abs(d) > 1e-3
                    # does the wrong row equal another channel's right row on the same queries?
                    allw = want[o, :512].reshape(512, -1)[:, sel]
                    dist = np.abs(allw[:, wrong] - g_[sel][wrong][None, :]).max(axis=1)
                    best = int(dist.argmin())
                    print('   DEEP object %d channel %d (local %d of wave %d) query tile %d: %d / %d queries off (it groups %s); got mean %.5f rms %.5f, want rms %.5f, all-cell mean %.5f; '
                          'closest other channel %d (max dist %.2e)' % (o, ch, ch % 64, 4 + ch // 64, qt, int(wrong.sum()), len(sel), sorted(set((np.nonzero(wrong)[0] // 16).tolist())),
                          g_[sel][wrong].mean(), np.sqrt((g_[sel][wrong] ** 2).mean()), np.sqrt((w_[sel][wrong] ** 2).mean()), mean_all[o, ch], best, dist[best]))
    if bad.any():
        for o in range(no):
            if not bad[o].any():
                continue
            cells = np.nonzero(bad[o].any(axis=0).reshape(-1))[0]
            chans = np.nonzero(bad[o].any(axis=(1, 2)))[0]
            print('   object %d: %d bad values, cells %d..%d (query tiles %s), channels %d..%d (%d distinct), worst %.3e at %s; want %.4f got %.4f' % (
                o, int(bad[o].sum()), cells.min(), cells.max(), sorted(set((cells // 64).tolist()))[:12], chans.min(), chans.max(), len(chans),
                err[o].max(), np.unravel_index(err[o].argmax(), err[o].shape),
                want[o][np.unravel_index(err[o].argmax(), err[o].shape)], got[o][np.unravel_index(err[o].argmax(), err[o].shape)]))
