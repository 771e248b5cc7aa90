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


# multi_scale_inference cases: (name, FRAME_SCALES, FLIP_LR)
MSI_CASES = [('s1', [1.0], False), ('s075_1_flip', [0.75, 1.0], True)]
MSI_CLIP = dict(N=3, K=3, H=96, W=128, seed=21, size=1.4, memorize_every=2)
