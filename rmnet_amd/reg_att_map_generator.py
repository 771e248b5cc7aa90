# -*- coding: utf-8 -*-
"""Drop-in for the reference's ``extensions/reg_att_map_generator`` package.

Reference interface (extensions/reg_att_map_generator/__init__.py:14-33)::

    RegionalAttentionMapGenerator()(mask, prob_threshold=0.5, n_pts_threshold=10,
                                    n_bbox_loose_pixels=64) -> (att_map f32 [B,K,H,W], bbox i32 [B,K,4])

and the compiled module it wraps, ``reg_att_map_generator.forward(mask, thr, n_pts, loose)``
(reg_att_map_generator_cuda.cpp:26-38).  Both are provided here on top of the gfx950 kernels in
``csrc/region_map.hip``.  Differences, all deliberate: outputs are allocated on ``mask.device``
(the reference uses the *current* device, .cu:104-109), launch errors raise instead of being
printed (.cu:117-121), and dtype is checked.  ``backward`` keeps the reference's stub semantics
(all-ones gradient for the mask, __init__.py:22-24).
"""

import torch

from . import ops


def forward(mask, prob_threshold, n_pts_threshold, n_bbox_loose_pixels):
    """Same contract as the pybind ``forward``: returns ``[reg_att_map, bboxes]``."""
    att, bboxes, _ = ops.region_map(mask, prob_threshold, n_pts_threshold, n_bbox_loose_pixels)
    return [att, bboxes]


class RegionalAttentionMapGeneratorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, prob_threshold, n_pts_threshold, n_bbox_loose_pixels):
        att_map, bbox = forward(mask, prob_threshold, n_pts_threshold, n_bbox_loose_pixels)
        ctx.mark_non_differentiable(bbox)
        return att_map, bbox

    @staticmethod
    def backward(ctx, grad_att_map, grad_bbox):
        return torch.ones_like(grad_att_map), None, None, None


class RegionalAttentionMapGenerator(torch.nn.Module):
    def forward(self, mask, prob_threshold=0.5, n_pts_threshold=10, n_bbox_loose_pixels=64):
        return RegionalAttentionMapGeneratorFunction.apply(mask, prob_threshold, n_pts_threshold,
                                                           n_bbox_loose_pixels)
