# -*- coding: utf-8 -*-
"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on bk_main's access pattern
(MI355X_MICROARCH.md, HBM section: a wide coalesced stream reports 1/2 of its bytes on gfx950).

Launch A: ONE object, ONE query tile (7x9 query box = 63 cells + the mean slot), dense memory, T = 5: the launch is
   cut into 4-tile column blocks, one workgroup each, so every K/V byte of the bank is fetched by exactly one
   workgroup, once: 5 frames x 51 tiles x (2 x 8 KB keys + 2 x 32 KB values) = 20,889,600 B (+ 32 KB of query).
   The L2s are flushed between launches with a 1 GiB copy (the bank would otherwise stay L2-resident).
Launch B: torch's device copy of 1 GiB (16 B per lane, read 2^30 B + write 2^30 B): the guide's reference pattern.

Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python tools/pmc_calib.py
(and again with WRITE_SIZE)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmnet_amd import ops

dev = torch.device('cuda', 0)
no, T, h, w = 1, 5, 30, 54
g = torch.Generator().manual_seed(0)
k = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev)
v = torch.randn(no, 512, h, w, generator=g).to(dev)
bank = ops.MemoryBank(no, T, h, w, dev, precision=os.environ.get('RMNET_BANK_PRECISION', 'split'))
full = torch.tensor([[0, w - 1, 0, h - 1]], dtype=torch.int32, device=dev)
for t in range(T):
    bank.append(t, k, v, full)
q = torch.tensor([[10, 18, 5, 11]], dtype=torch.int32, device=dev)       # 9 x 7 = 63 query cells
big_a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
big_b = torch.empty_like(big_a)
for _ in range(2):
    bank.read(T, k, v, q)
torch.cuda.synchronize()
for _ in range(6):
    big_b.copy_(big_a)                      # launch B, and the L2 flush for launch A
    torch.cuda.synchronize()
    bank.read(T, k, v, q)                   # launch A
    torch.cuda.synchronize()
print('expected bytes: bk_main fetch %d ; copy fetch %d write %d' % (T * 51 * (2 * 8192 + 2 * 32768) + 64 * 128 * 4, 1 << 30, 1 << 30))
