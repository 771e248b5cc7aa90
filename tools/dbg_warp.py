# -*- coding: utf-8 -*-
"""Bitwise comparison of the fused warp (rmnet_region_map_warped_f32, `warped` output) with
RMNet.warp evaluated by torch on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmnet_amd import ops
from rmnet_amd.rmnet import RMNet
dev = torch.device('cuda', 0)
net = RMNet(None)
g = torch.Generator().manual_seed(1)
for (B, K, H, W, amp) in [(1, 2, 480, 854, 3.0), (2, 3, 97, 131, 8.0), (1, 2, 480, 854, 0.5)]:
    m = torch.rand(B, K, H, W, generator=g).to(dev)
    f = (torch.randn(B, 2, H, W, generator=g) * amp).to(dev)
    want = net.warp(m, f)[0]
    _, bb, _, got = ops.region_map(m, want_map=False, flow=f, want_warped=True)
    d = (got[:, 1:] - want[:, 1:]).abs()
    ne = int((got[:, 1:] != want[:, 1:]).sum())
    _, bb2, _ = ops.region_map(want.contiguous(), want_map=False)
    print((B, K, H, W, amp), 'mismatching values', ne, 'of', d.numel(), 'max abs', float(d.max()), 'boxes equal', bool(torch.equal(bb, bb2)))
