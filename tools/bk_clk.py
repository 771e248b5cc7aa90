# -*- coding: utf-8 -*-
"""Time line and static-queue accounting of bk_main (needs the stamped library: `PATCH=tools/patches/clk_stamps_r6.patch tools/build_variant.sh clk`,
run with RMNET_HIP_LIB=build/variants/lib_clk.so; FLUSH=<MB> streams that much between two reads = the frame loop's cold caches): per workgroup,
elapsed shader cycles (s_memtime) vs elapsed constant-rate ticks (s_memrealtime, 100 MHz) of its compute part, the
number of static-queue tickets it served and when it left the kernel.
    python tools/bk_clk.py <no> <q_h> <q_w> <m_h> <m_w> [T]"""
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
flush = int(os.environ.get('FLUSH', '0'))
if flush:
    fa = torch.empty(flush * 262144, device=dev); fb = torch.ones(flush * 262144, device=dev)
for _ in range(20):
    if flush:
        fa.copy_(fb); fa.mul_(1.0001)
    bank.read(T, qk, qv, qr, ws=ws)
torch.cuda.synchronize()
# the stamps start 64 bytes after the plan records (which end somewhere inside the last 256-byte pad)
plan_end = (no * 12 * 4)
raw = ws[nb - 16384 - 256 - ((plan_end + 255) // 256 * 256) + plan_end + 64:][:256 * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
comp = raw[raw[:, 1] > 0]
ghz = np.zeros(len(comp))
print('no=%d: %d workgroups stamped; compute part: shader cycles %.0f..%.0f, real us %.1f..%.1f, clock GHz mean %.3f min %.3f max %.3f'
      % (no, len(comp), comp[:, 0].min(), comp[:, 0].max(), comp[:, 1].min() / 100.0, comp[:, 1].max() / 100.0, ghz.mean(), ghz.min(), ghz.max()))
tick, left = raw[:, 2], raw[:, 3] / 100.0
early = comp[:, 1] / 100.0 < 0.5 * np.median(comp[:, 1] / 100.0)           # workgroups without a chunk: set aside for the static part
print('static queue: %d tickets served in all; set-aside workgroups (%d): %s tickets each, left at %.1f..%.1f us; '
      'the others: %d tickets in all after their compute part, left at %.1f..%.1f us'
      % (tick.sum(), int(early.sum()), sorted(tick[(raw[:, 1] > 0)][early].tolist()), left[(raw[:, 1] > 0)][early].min() if early.any() else 0,
         left[(raw[:, 1] > 0)][early].max() if early.any() else 0, int(tick[(raw[:, 1] > 0)][~early].sum()),
         left[(raw[:, 1] > 0)][~early].min(), left[(raw[:, 1] > 0)][~early].max()))
c = raw[(raw[:, 1] > 0)][~early]
us = lambda col: c[:, col] / 100.0
print('compute workgroups, microseconds since kernel entry (median [min..max]): plan done %.1f [%.1f..%.1f]; first tile walk starts %.1f [%.1f..%.1f]; '
      'last tile walk over %.1f [%.1f..%.1f]; epilogue (publish / in-place merge / output) over %.1f [%.1f..%.1f]; deferred merges over %.1f [%.1f..%.1f]; '
      'left the kernel (static queue drained) %.1f [%.1f..%.1f]'
      % tuple(x for col in (4, 5, 6, 1, 7, 3) for x in (np.median(us(col)), us(col).min(), us(col).max())))
