# -*- coding: utf-8 -*-
"""Micro-benchmark of the memory-read op alone (mr_main / mr_combine via HIP events) on synthetic
K/V/Q for the BASELINE configs.  Usage: python tools/mr_bench.py [--cfg 2] [--frac 0.19] [--reps 50]"""
import argparse, ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rmnet_amd import ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import HipEvents, algorithmic_bytes

CFG = {1: (1, 3, 30, 54), 2: (1, 5, 30, 54), 3: (5, 5, 30, 54), 4: (4, 5, 30, 54), 5: (3, 20, 45, 80)}

def rect_of(frac, h, w):
    rh, rw = max(1, int(round(h * frac ** 0.5))), max(1, int(round(w * frac ** 0.5)))
    y0, x0 = (h - rh) // 2, (w - rw) // 2
    return (x0, x0 + rw - 1, y0, y0 + rh - 1)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', type=int, default=2)
    ap.add_argument('--fracs', type=str, default='0.19,0.45,1.0,dense')
    ap.add_argument('--reps', type=int, default=50)
    a = ap.parse_args()
    no, T, h, w = CFG[a.cfg]
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6).to(dev)
    mv = torch.randn(no, 512, T, h, w, generator=g).to(dev)
    qk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev)
    qv = torch.randn(no, 512, h, w, generator=g).to(dev)
    ev = HipEvents(3 * a.reps)
    ab = algorithmic_bytes(no, T, h, w)
    for f in a.fracs.split(','):
        if f == 'dense':
            mr = qr = None
        else:
            r = rect_of(float(f), h, w)
            mr = torch.tensor([[r] * T] * no, dtype=torch.int32, device=dev)
            qr = torch.tensor([r] * no, dtype=torch.int32, device=dev)
        for _ in range(5):
            ops.memory_read(mk, mv, qk, qv, mr, qr)
        torch.cuda.synchronize()
        for i in range(a.reps):
            ops.memory_read(mk, mv, qk, qv, mr, qr, events=tuple(ev.ev[3 * i:3 * i + 3]))
        torch.cuda.synchronize()
        m = [ev.elapsed_ms(ev.ev[3 * i], ev.ev[3 * i + 1]) * 1e3 for i in range(a.reps)]
        c = [ev.elapsed_ms(ev.ev[3 * i + 1], ev.ev[3 * i + 2]) * 1e3 for i in range(a.reps)]
        floor = [ev.elapsed_ms(ev.ev[3 * i + 2], ev.ev[3 * i + 3]) * 1e3 for i in range(a.reps - 1)]
        # split-fp16 bank path
        bank = ops.MemoryBank(no, T, h, w, dev)
        for t in range(T):
            bank.append(t, mk[:, :, t].contiguous(), mv[:, :, t].contiguous(), None if mr is None else mr[:, t].contiguous())
        for _ in range(5):
            bank.read(T, qk, qv, qr)
        torch.cuda.synchronize()
        for i in range(a.reps):
            bank.read(T, qk, qv, qr, events=tuple(ev.ev[3 * i:3 * i + 3]))
        torch.cuda.synchronize()
        bm = [ev.elapsed_ms(ev.ev[3 * i], ev.ev[3 * i + 1]) * 1e3 for i in range(a.reps)]
        bc = [ev.elapsed_ms(ev.ev[3 * i + 1], ev.ev[3 * i + 2]) * 1e3 for i in range(a.reps)]
        print('cfg%d frac=%-5s BANK  main avg %.2f min %.2f us | combine avg %.2f min %.2f us | %.0f GB/s (main)'
              % (a.cfg, f, np.mean(bm), np.min(bm), np.mean(bc), np.min(bc), ab / np.mean(bm) / 1e3))
        print('cfg%d frac=%-5s main avg %.2f min %.2f us | combine avg %.2f min %.2f us | %.0f GB/s (main) | gap-to-next avg %.2f us'
              % (a.cfg, f, np.mean(m), np.min(m), np.mean(c), np.min(c), ab / np.mean(m) / 1e3, np.mean(floor)))

if __name__ == '__main__':
    main()
