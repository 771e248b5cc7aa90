# -*- coding: utf-8 -*-
"""5-object 480x854 loop: GPU (fused / un-fused / exact-fp32 bank) vs the CPU path, per frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from rmnet_amd import networks
from rmnet_amd.rmnet import RMNet
from rmnet_amd.synthetic import synthetic_clip
dev = torch.device('cuda', 0)
prod = networks.procedural_init_(RMNet(None))
ref = oracle.OracleRMNet()
ref.load_state_dict(prod.state_dict())
prod = prod.to(dev).eval(); ref = ref.eval()
n_obj, K, H, W, N = 5, 6, 480, 854, 3
frames, masks, flows, n_objects = synthetic_clip(N, n_obj + 1, H, W, seed=K, size=1.1)
with torch.no_grad():
    est_cpu = ref(frames, masks, flows, n_objects, 1)
    outs = {'unfused': prod(frames, masks, flows, n_objects, 1).cpu(),
            'unfused_exact': prod(frames, masks, flows, n_objects, 1, _exact=True).cpu()}
    prod.fuse_epilogues()
    outs['fused'] = prod(frames, masks, flows, n_objects, 1).cpu()
    outs['fused_exact'] = prod(frames, masks, flows, n_objects, 1, _exact=True).cpu()
for name, e in outs.items():
    d = (e - est_cpu).abs()
    lab, lab_cpu = e.argmax(2).numpy(), est_cpu.argmax(2).numpy()
    print('%-14s max diff per frame %s ; #pixels > 1e-3: %s ; label agreement %.6f ; IoU %s' % (
        name, ['%.2e' % float(d[0, t].max()) for t in range(N)], [int((d[0, t] > 1e-3).sum()) for t in range(N)],
        float((lab == lab_cpu).mean()), ['%.5f' % oracle.iou(lab[:, 1:] == k, lab_cpu[:, 1:] == k) for k in range(1, K)]))
a, b = outs['fused'], outs['fused_exact']
print('fused vs fused_exact max diff', float((a - b).abs().max()))
