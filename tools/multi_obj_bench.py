# -*- coding: utf-8 -*-
"""BASELINE configs[2]: multi-object 480p clips through RMNet.forward (frames/s and a sanity check that
nothing in the loop scales badly with the number of objects)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmnet_amd import networks
from rmnet_amd.rmnet import RMNet
from rmnet_amd.synthetic import synthetic_clip
from rmnet_amd.tiny_flownet import TinyFlowNet
dev = torch.device('cuda', 0)
torch.set_grad_enabled(False)
torch.backends.cudnn.benchmark = True
net = networks.procedural_init_(RMNet(None)).to(dev).eval().fuse_epilogues()
tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev).eval().fuse_epilogues()
for n_obj, B in ((1, 4), (3, 2), (5, 1)):
    N = 8
    clips = [synthetic_clip(N, n_obj + 1, 480, 854, seed=10 + i, size=1.6) for i in range(B)]
    frames = torch.cat([c[0] for c in clips]).to(dev)
    masks = torch.cat([c[1] for c in clips])
    n_objects = torch.cat([c[3] for c in clips])
    flows = tfn(frames)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        est = net(frames, masks, flows, n_objects, 2, device=dev)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('objects/clip %d, clips %d: %.1f ms per frame-step, %.1f frames/s, %.1f object-frames/s' %
          (n_obj, B, dt / (N - 1) * 1e3, B * (N - 1) / dt, B * n_obj * (N - 1) / dt))
