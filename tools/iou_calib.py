# -*- coding: utf-8 -*-
"""Calibration of the north star's IoU bar ("mask IoU within 1e-3 of the reference CPU path"): the CPU path
(oracle.OracleRMNet, plain torch + C ops) against the GPU frame loop with the bank read in its three arithmetics --
exact fp32 (TensorBank + mr_main), split fp16 (3 MFMA terms) and fp16 operands (1 term) -- on multi-object clips.
Prints one row per (clip, arithmetic): per-object label IoU vs the CPU path over the whole clip, the worst single-frame IoU,
and the largest probability difference; plus the same rows against the exact-fp32 GPU run (what the arithmetic alone does).
    python tools/iou_calib.py [N frames] [threads] [cases]   [blob size]   (cases: comma list of 3o480,5o480,3o720,1o480)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from rmnet_amd import networks
from rmnet_amd.rmnet import RMNet
from rmnet_amd.synthetic import synthetic_clip
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
want = sys.argv[3].split(',') if len(sys.argv) > 3 else ['3o480', '5o480', '3o720']
size = float(sys.argv[4]) if len(sys.argv) > 4 else 1.1          # blob radius scale of rmnet_amd.synthetic.synthetic_clip
CASES = {'1o480': (1, 480, 854, 5, 1), '3o480': (3, 480, 854, 5, 3), '5o480': (5, 480, 854, 2, 4), '3o720': (3, 720, 1280, 3, 5)}
torch.set_grad_enabled(False)
torch.set_num_threads(nt)
oracle.set_num_threads(nt)


def iou(a, b):
    u = (a | b).sum()
    return 1.0 if u == 0 else float((a & b).sum()) / float(u)


def row(tag, a, b, n_obj):
    la, lb = a.argmax(2).numpy(), b.argmax(2).numpy()
    d = (a - b).abs()
    clip = [iou(la[:, 1:] == k, lb[:, 1:] == k) for k in range(1, n_obj + 1)]
    worst = min(min(iou(la[0, t] == k, lb[0, t] == k) for k in range(1, n_obj + 1)) for t in range(1, a.shape[1]))
    print('  %-22s clip IoU per object %s  min %.5f | worst single frame %.5f | max prob diff %.2e | values > 1e-3: %d of %d'
          % (tag, ' '.join('%.5f' % c for c in clip), min(clip), worst, float(d.max()), int((d > 1e-3).sum()), d.numel()), flush=True)
    return min(clip)


cpu = networks.procedural_init_(oracle.OracleRMNet(reader='torch')).eval()
prod = networks.procedural_init_(RMNet(None)).to(dev).eval()
prod.fuse_epilogues()
for name in want:
    n_obj, H, W, every, seed = CASES[name]
    frames, masks, flows, n_objects = synthetic_clip(N, n_obj + 1, H, W, seed=seed, size=size)
    t0 = time.time()
    ref = cpu(frames, masks, flows, n_objects, every)
    print('%s (blob size %.1f): %d objects %dx%d, %d frames, memorize_every %d; CPU path %.0f s at %d threads; object cover of the last frame: %s'
          % (name, size, n_obj, H, W, N, every, time.time() - t0, nt,
             ' '.join('%.3f' % float((ref[0, -1].argmax(0) == k).float().mean()) for k in range(1, n_obj + 1))), flush=True)
    runs = {}
    runs['exact'] = prod(frames, masks, flows, n_objects, every, _exact=True).cpu()
    for mode in ('split', 'f16'):
        prod.read_precision = mode
        runs[mode] = prod(frames, masks, flows, n_objects, every).cpu()
    prod.read_precision = 'split'
    for mode in ('exact', 'split', 'f16'):
        row('%s vs CPU path' % mode, runs[mode], ref, n_obj)
    for mode in ('split', 'f16'):
        row('%s vs exact (GPU)' % mode, runs[mode], runs['exact'], n_obj)
