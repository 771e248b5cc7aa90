# -*- coding: utf-8 -*-
"""What does the precision of the bank read do to the masks of a whole clip?  Runs the frame loop three times on the
GPU -- bank read in the default split-fp16 mode, in the fp16-operand mode, and the exact-fp32 read -- and prints, per
clip, label IoU per object against the exact run and the largest probability difference.  python tools/iou_terms.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rmnet_amd import networks
from rmnet_amd.rmnet import RMNet
from rmnet_amd.synthetic import synthetic_clip
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def iou(a, b):
    u = (a | b).sum()
    return 1.0 if u == 0 else float((a & b).sum()) / float(u)


prod = networks.procedural_init_(RMNet(None)).to(dev).eval()
prod.fuse_epilogues()
for n_obj, H, W, every, seed in [(1, 480, 854, 5, 1), (1, 480, 854, 1, 2), (3, 480, 854, 5, 3), (5, 480, 854, 2, 4), (3, 720, 1280, 3, 5)]:
    frames, masks, flows, n_objects = synthetic_clip(N, n_obj + 1, H, W, seed=seed, size=1.1)
    with torch.no_grad():
        b = prod(frames, masks, flows, n_objects, every, _exact=True).cpu()
        lb = b.argmax(2).numpy()
        for mode in ('split', 'f16'):
            prod.read_precision = mode
            a = prod(frames, masks, flows, n_objects, every).cpu()
            la = a.argmax(2).numpy()
            d = (a - b).abs()
            per_frame = [min(iou(la[0, t] == k, lb[0, t] == k) for k in range(1, n_obj + 1)) for t in range(N)]
            print('%-5s %d obj %dx%d every %d: clip IoU %s ; worst single-frame IoU %.5f ; max prob diff %.2e ; values > 1e-3: %d of %d ; label agreement %.6f'
                  % (mode, n_obj, H, W, every, ['%.5f' % iou(la[:, 1:] == k, lb[:, 1:] == k) for k in range(1, n_obj + 1)], min(per_frame[1:]),
                     float(d.max()), int((d > 1e-3).sum()), d.numel(), float((la == lb).mean())))
