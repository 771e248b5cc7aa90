# -*- coding: utf-8 -*-
"""Seeded input generators shared by tests/golden/make_golden.py (which feeds them to the REFERENCE
in the build container) and by the tests (which feed the same inputs to the oracle and to the HIP
kernels).  Only ``numpy.random.RandomState`` (the frozen legacy generator) is used, so the inputs are
bit-identical wherever they are rebuilt; the fixtures store the reference's OUTPUTS only."""

import numpy as np

# (B, K, H, W) of the region-map cases pinned by utils/helpers.py:93-102 get_bounding_boxes
REGION_BOX_SHAPES = [
    (1, 2, 480, 864), (1, 2, 480, 854), (1, 11, 480, 854), (1, 11, 480, 864), (2, 3, 150, 250),
    (1, 4, 33, 47), (1, 2, 720, 1280), (3, 2, 1, 5), (1, 6, 5, 1), (1, 3, 96, 160),
    (1, 11, 64, 96), (2, 11, 40, 56), (1, 2, 17, 129), (1, 5, 200, 320), (1, 2, 480, 854),
    (1, 11, 120, 214), (1, 3, 64, 64), (1, 2, 65, 65), (4, 2, 30, 54), (1, 8, 45, 80),
    (1, 2, 480, 854), (1, 11, 240, 427), (1, 4, 100, 100), (1, 2, 2, 2),
]


def region_box_case(i):
    """Soft mask [B,K,H,W] f32 of case ``i``: per channel nothing (20 %), a soft blob whose values
    straddle the 0.5 threshold, stray single pixels, and some values at exactly 0.5 (the comparison is
    inclusive, reg_att_map_generator.cu:41)."""
    B, K, H, W = REGION_BOX_SHAPES[i]
    rng = np.random.RandomState(9000 + i)
    m = np.zeros((B, K, H, W), np.float32)
    for b in range(B):
        for k in range(K):
            u = rng.rand()
            if u < 0.2:
                continue                                   # empty channel
            y0, x0 = rng.randint(0, H), rng.randint(0, W)
            y1, x1 = rng.randint(y0, H) + 1, rng.randint(x0, W) + 1
            blob = (rng.rand(y1 - y0, x1 - x0) * 1.2).astype(np.float32)
            blob[blob > 1.0] = 0.5                         # exactly at the threshold
            m[b, k, y0:y1, x0:x1] = blob
            for _ in range(rng.randint(0, 4)):             # stray pixels, below and above the threshold
                m[b, k, rng.randint(0, H), rng.randint(0, W)] = np.float32(rng.choice([0.49999, 0.5, 0.7, 1.0]))
    return m


def boxes_from_reference_tight(tight, K, H, W):
    """What reg_att_map_generator.cu:30-77 yields for (prob_threshold, n_pts_threshold = 1,
    n_bbox_loose_pixels = 0) given the reference's tight boxes ``tight`` [B*K,4] (-1 rows = the Python
    function returned None): channel 0 is never touched (.cu:25-27, zeros from the host), an empty
    channel falls back to the full frame (.cu:57-61), anything else is the tight box itself
    (.cu:63-74 with L = 0)."""
    out = np.zeros((tight.shape[0], 4), np.int32)
    for r in range(tight.shape[0]):
        if r % K == 0:
            continue
        out[r] = (0, W - 1, 0, H - 1) if tight[r, 0] < 0 else tight[r]
    return out


# ---- region-map fuzz: the loosen / clamp / point-count branches (reg_att_map_generator.cu:55-77) ---------------
REGION_FUZZ_LOOSE = [0, 1, 63, 64, 65]
REGION_FUZZ_NPTS = [1, 9, 10, 11]
REGION_FUZZ_SHAPES = [(1, 3, 150, 200), (1, 4, 96, 160), (2, 2, 70, 66), (1, 3, 200, 131), (1, 2, 480, 854),
                      (1, 5, 131, 140), (1, 3, 64, 64), (1, 11, 140, 180)] * 5


def region_fuzz_case(i):
    """Soft mask [B,K,H,W] f32 of fuzz case ``i``.  Every channel k >= 1 is one of
      * a handful of isolated above-threshold pixels: exactly 0, 8, 9, 10, 11 or 12 of them (the point-count
        fallback, .cu:57-61, on both sides of every n_pts_threshold of REGION_FUZZ_NPTS);
      * a blob whose tight box has its edges AT, one before and one after the distances at which the four clamp
        expressions of .cu:63-74 switch (x_min <= L, x_max + L >= W, same for y) for every L of REGION_FUZZ_LOOSE,
        incl. boxes touching each border; values straddle the threshold, some are exactly 0.5."""
    B, K, H, W = REGION_FUZZ_SHAPES[i]
    rng = np.random.RandomState(7000 + i)
    m = (rng.rand(B, K, H, W) * 0.49).astype(np.float32)          # below-threshold clutter everywhere
    edge = [0, 1, 2, 62, 63, 64, 65, 66]
    for b in range(B):
        for k in range(1, K):
            if rng.rand() < 0.35:
                n = int(rng.choice([0, 8, 9, 10, 11, 12]))
                pix = rng.choice(H * W, size=n, replace=False)
                m[b, k].reshape(-1)[pix] = rng.choice([0.5, 0.51, 0.9, 1.0], size=n).astype(np.float32)
                continue
            def span(n):
                lo = int(rng.choice(edge + [rng.randint(0, n)]))
                hi = n - 1 - int(rng.choice(edge + [rng.randint(0, n)]))
                lo, hi = min(lo, n - 1), max(min(hi, n - 1), 0)
                return (lo, hi) if lo <= hi else (hi, lo)
            x0, x1 = span(W)
            y0, y1 = span(H)
            blob = (rng.rand(y1 - y0 + 1, x1 - x0 + 1) * 1.1).astype(np.float32)
            blob[blob > 1.0] = 0.5
            m[b, k, y0:y1 + 1, x0:x1 + 1] = blob
            for (yy, xx) in ((y0, x0), (y1, x1), (y0, x1), (y1, x0)):     # the tight box really is (x0, x1, y0, y1)
                m[b, k, yy, xx] = 0.5 if rng.rand() < 0.5 else 0.75
    return m


def boxes_after_loosen(tight, npts, K, H, W, n_pts_threshold, loose):
    """reg_att_map_generator.cu:55-77 evaluated on the REFERENCE's tight boxes ``tight`` [B*K,4] (utils/helpers.py:93-102;
    -1 rows = None) and above-threshold pixel counts ``npts`` [B*K]: channel 0 untouched (zeros), fewer than
    ``n_pts_threshold`` points -> the full frame (.cu:57-61), else the four clamp expressions (.cu:63-74)."""
    out = np.zeros((tight.shape[0], 4), np.int32)
    L = int(loose)
    for r in range(tight.shape[0]):
        if r % K == 0:
            continue
        if int(npts[r]) < n_pts_threshold:
            out[r] = (0, W - 1, 0, H - 1)
            continue
        x0, x1, y0, y1 = (int(v) for v in tight[r])
        out[r] = (0 if x0 <= L else x0 - L, W - 1 if x1 + L >= W else x1 + L,
                  0 if y0 <= L else y0 - L, H - 1 if y1 + L >= H else y1 + L)
    return out


# multi_scale_inference cases: (name, FRAME_SCALES, FLIP_LR)
MSI_CASES = [('s1', [1.0], False), ('s075_1_flip', [0.75, 1.0], True)]
MSI_CLIP = dict(N=3, K=3, H=96, W=128, seed=21, size=1.4, memorize_every=2)
