# -*- coding: utf-8 -*-
"""What the sticky 'auto' rule saves on a network whose soft-max is peaked (keys x 4, tests/live_fixture.py): wall time of the first clip
(read in f16, logits measured > AUTO_LOGIT_BOUND, re-read in split) against the following clips of the same network (start in split).
    python tools/sticky_auto_timing.py [fixture] [scale]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from rmnet_amd import networks
from rmnet_amd.rmnet import RMNet
import live_fixture as lf

torch.set_grad_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else 'live480-a'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dev = torch.device('cuda', 0)
frames, masks, flows, n_objects, every, _ = lf.make_clip(name, N=12)
frames, masks, flows = frames.to(dev), masks.to(dev), flows.to(dev)


def timed(net):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net(frames, masks, flows, n_objects, every)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), dict(net.last_clip)


for label, s in (('near-uniform soft-max (keys x 1)', 1.0), ('peaked soft-max (keys x %g)' % scale, scale)):
    net = networks.procedural_init_(RMNet(None)).to(dev).eval()
    lf.scale_keys(net, s)
    net.fuse_epilogues()
    warm = networks.procedural_init_(RMNet(None, read_precision='split')).to(dev).eval()      # MIOpen / allocator warm-up on another network
    warm.fuse_epilogues()
    warm(frames, masks, flows, n_objects, every)
    print(label)
    for i in range(4):
        ms, info = timed(net)
        print('  clip %d: %7.1f ms  read %s%s  (largest logit %.1f)' % (i + 1, ms, info['read_precision'], ' -> re-read: ' + info['reread'] if info['reread'] else '', info['logit_max']), flush=True)
