# -*- coding: utf-8 -*-
"""bk_main / mr_combine timing with identical, explicitly sized boxes for every object:
    python tools/chunk_bench.py <no> <q_h> <q_w> <m_h> <m_w> [T]
(lets a launch be made of exactly aligned chunks: e.g. 8 objects, query 21x24 (8 tiles), memory 24x32
(24 tiles per frame) -> 256 single-segment chunks of 30 tiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rmnet_amd import ops
from bench import HipEvents
no, qh, qw, mh, mw = [int(x) for x in sys.argv[1:6]]
T = int(sys.argv[6]) if len(sys.argv) > 6 else 5
h, w = (qh, qw) if qh > 30 else (30, 54)   # (a box larger than the 480p grid: that grid, dense)
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
reps = 40
ev = HipEvents(3 * reps)
floor = ev.floor_us(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(5):
    bank.read(T, qk, qv, qr)
torch.cuda.synchronize()
flush = int(os.environ.get('FLUSH', '0'))   # MB streamed between two reads (evicts L2 / MALL: the frame loop's condition)
staged = os.environ.get('STAGED', '0') == '1' # frame count from the device counter (read_staged), as the frame loop reads
if staged:
    bank.committed = T - 1
    bank.n_dev.fill_(T - 1)
if flush:
    fa = torch.empty(flush * 262144, device=dev)
    fb = torch.ones(flush * 262144, device=dev)
for i in range(reps):
    if flush:
        fa.copy_(fb)
        fa.mul_(1.0001)
    if staged:
        bank.read_staged(qk, qv, qr, events=tuple(ev.ev[3 * i:3 * i + 3]))
    else:
        bank.read(T, qk, qv, qr, events=tuple(ev.ev[3 * i:3 * i + 3]))
torch.cuda.synchronize()
bm = [ev.elapsed_ms(ev.ev[3 * i], ev.ev[3 * i + 1]) * 1e3 - floor for i in range(reps)]
bc = [ev.elapsed_ms(ev.ev[3 * i + 1], ev.ev[3 * i + 2]) * 1e3 - floor for i in range(reps)]
nqt = (qh * qw + 1 + 63) // 64
njt = T * ((mh * mw + 31) // 32)
# the plan bk_main left in the workspace (common.h: kPlanInts = 12 ints per object: ..., [9] chunk length, [10] chunks of the launch)
from rmnet_amd import _lib
lib = _lib.load()
ws = torch.zeros(int(lib.rmnet_bank_read_workspace_bytes_for(no, h, w, T)), dtype=torch.uint8, device=dev)
bank.read(T, qk, qv, qr, ws=ws)
torch.cuda.synchronize()
a256 = lambda x: (x + 255) & ~255
tsl = 256 + no * ((h * w + 63) // 64)
pl = ws[a256(tsl * 512 * 64 * 4) + a256(tsl * 2 * 64 * 4):][:no * 48].view(torch.int32).view(no, 12).cpu().numpy()
print('plan: chunks %d | per object (nqt, njt, chunk length): %s' % (pl[0, 10], ' '.join('%d,%d,%d' % (r[1], r[2], r[9]) for r in pl[:6])))
print('no=%d nqt=%d njt=%d pairs*tiles=%d | bk_main avg %.2f min %.2f us | combine avg %.2f min %.2f us (event floor %.2f us subtracted)'
      % (no, nqt, njt, no * nqt * njt, np.mean(bm), np.min(bm), np.mean(bc), np.min(bc), floor))
