# -*- coding: utf-8 -*-
"""Calibration of the north star's IoU bar ("mask IoU within 1e-3 of the reference CPU path"): the CPU path
(oracle.OracleRMNet, plain torch + C ops) against the GPU frame loop with the bank read in its arithmetics -- exact fp32
(TensorBank + mr_main), split fp16 (3 MFMA terms), qx (fp16 operands with an exact query) and fp16 operands (1 term).
One row per (clip, arithmetic): per-object label IoU vs the CPU path over the whole clip, the worst single-frame IoU, the largest
probability difference and -- one-object clips -- the largest difference of the live foreground logits; then the same rows against
the exact-fp32 GPU run (what the arithmetic alone does).
    python tools/iou_calib.py [N frames] [threads] [cases] [blob size]
cases: comma list of  3o480,5o480,3o720,1o480 (round-4 clips: the one-object one is SATURATED)  and  live480-a|b|c (tests/live_fixture.py:
one-object clips with live mask boundaries; N = 0 keeps the fixture's own length)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import oracle
from rmnet_amd import networks
from rmnet_amd.rmnet import RMNet
from rmnet_amd.synthetic import synthetic_clip
import live_fixture as lf
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
want = sys.argv[3].split(',') if len(sys.argv) > 3 else ['3o480', '5o480', '3o720']
size = float(sys.argv[4]) if len(sys.argv) > 4 else 1.1          # blob radius scale of rmnet_amd.synthetic.synthetic_clip
every_override = int(os.environ.get('EVERY', '0'))
CASES = {'1o480': (1, 480, 854, 5, 1), '3o480': (3, 480, 854, 5, 3), '5o480': (5, 480, 854, 2, 4), '3o720': (3, 720, 1280, 3, 5)}
MODES = os.environ.get('MODES', 'exact,split,qx,f16').split(',')
torch.set_grad_enabled(False)
torch.set_num_threads(nt)
oracle.set_num_threads(nt)


def iou(a, b):
    u = (a | b).sum()
    return 1.0 if u == 0 else float((a & b).sum()) / float(u)


def row(tag, a, b, n_obj, la_=None, lb_=None):
    la, lb = a.argmax(2).numpy(), b.argmax(2).numpy()
    d = (a - b).abs()
    clip = [iou(la[:, 1:] == k, lb[:, 1:] == k) for k in range(1, n_obj + 1)]
    worst = min(min(iou(la[0, t] == k, lb[0, t] == k) for k in range(1, n_obj + 1)) for t in range(1, a.shape[1]))
    gap = '' if la_ is None or n_obj != 1 else ' | max live fg-logit diff %.2e' % lf.logit_gap(la_, lb_)
    print('  %-22s clip IoU per object %s  min %.5f | worst single frame %.5f | max prob diff %.2e | values > 1e-3: %d of %d%s'
          % (tag, ' '.join('%.5f' % c for c in clip), min(clip), worst, float(d.max()), int((d > 1e-3).sum()), d.numel(), gap), flush=True)
    return min(clip)


for name in want:
    cpu = networks.procedural_init_(oracle.OracleRMNet(reader='torch')).eval()
    prod = networks.procedural_init_(RMNet(None)).to(dev).eval()
    prod.fuse_epilogues()
    if name in lf.LIVE_CLIPS:
        frames, masks, flows, n_objects, every, delta = lf.make_clip(name, N=N or None, every=every_override or None)
        lf.shift_foreground_bias(cpu, delta)
        lf.shift_foreground_bias(prod, delta)
        n_obj, H, W = 1, frames.shape[3], frames.shape[4]
        what = 'live fixture, bias shift %.2f' % delta
    else:
        n_obj, H, W, every, seed = CASES[name]
        every = every_override or every
        frames, masks, flows, n_objects = synthetic_clip(N, n_obj + 1, H, W, seed=seed, size=size)
        what = 'blob size %.1f' % size
    t0 = time.time()
    ref, ref_l = cpu(frames, masks, flows, n_objects, every, return_logits=True)
    print('%s (%s): %d objects %dx%d, %d frames, memorize_every %d; CPU path %.0f s at %d threads; object cover of the last frame: %s'
          % (name, what, n_obj, H, W, frames.shape[1], every, time.time() - t0, nt,
             ' '.join('%.3f' % float((ref[0, -1].argmax(0) == k).float().mean()) for k in range(1, n_obj + 1))), flush=True)
    if n_obj == 1:
        lv = lf.liveness(ref)
        print('  foreground cover min %.3f max %.3f; pixels within 0.1 of the threshold: min %.4f max %.4f of the frame'
              % (min(c for c, _ in lv), max(c for c, _ in lv), min(n for _, n in lv), max(n for _, n in lv)), flush=True)
    runs = {}
    for mode in MODES:
        prod.read_precision = 'split' if mode == 'exact' else mode
        e, l = prod(frames, masks, flows, n_objects, every, _exact=(mode == 'exact'), return_logits=True)
        runs[mode] = (e.cpu(), l.cpu())
    for mode in MODES:
        row('%s vs CPU path' % mode, runs[mode][0], ref, n_obj, runs[mode][1], ref_l)
    if 'exact' in runs:
        for mode in MODES:
            if mode != 'exact':
                row('%s vs exact (GPU)' % mode, runs[mode][0], runs['exact'][0], n_obj, runs[mode][1], runs['exact'][1])
