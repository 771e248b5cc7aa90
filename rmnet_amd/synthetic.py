# -*- coding: utf-8 -*-
"""Seeded synthetic clips (no dataset is reachable offline).  Shapes and
dtypes follow the reference's test loader (utils/data_loaders.py:40-71): frames f32 [1,N,3,H,W],
one-hot masks u8 [1,N,K,H,W], flows f32 [1,N,2,H,W], n_objects i64 [1,N]."""

import torch


def synthetic_clip(N, K, H, W, seed=0, size=1.0):
    """Moving-blob clip: frames f32 [1,N,3,H,W], one-hot masks u8 [1,N,K,H,W], constant-drift
    flows f32 [1,N,2,H,W], n_objects i64 [1,N].  ``size`` scales the blob radii: 1.0 gives small
    objects (~4 % of the frame, used by the golden fixtures); 2.1 gives ~18 % of the frame, the object
    size SURVEY.md section 8d asks for in the fps workload (box + 128 px covers 35-55 % of 480p)."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.arange(H).view(H, 1).float()
    xs = torch.arange(W).view(1, W).float()
    frames = torch.zeros(1, N, 3, H, W)
    masks = torch.zeros(1, N, K, H, W, dtype=torch.uint8)
    flows = torch.zeros(1, N, 2, H, W)
    base = torch.rand(3, H, W, generator=g) * 0.4 - 0.2
    for t in range(N):
        label = torch.zeros(H, W, dtype=torch.long)
        img = base + 0.3 * torch.sin(xs / 17.0 + t * 0.1) * torch.cos(ys / 23.0)
        for o in range(1, K):
            cy = H * (0.25 + 0.5 * ((o * 37) % 100) / 100.0) + 2.0 * t
            cx = W * (0.2 + 0.6 * ((o * 61) % 100) / 100.0) + 3.0 * t
            ry, rx = H * (0.10 + 0.03 * o) * size, W * (0.08 + 0.02 * o) * size
            inside = ((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2 <= 1.0
            label[inside] = o
            img = img + inside.float() * torch.tensor([0.8, -0.5, 0.3]).view(3, 1, 1) * (1.0 if o % 2 else -1.0)
        frames[0, t] = img
        for k in range(K):
            masks[0, t, k] = (label == k).to(torch.uint8)
        flows[0, t, 0] = -3.0   # backward flow t -> t-1 (object moved +3 px in x)
        flows[0, t, 1] = -2.0
    n_objects = torch.full((1, N), K - 1, dtype=torch.long)
    return frames, masks, flows, n_objects


