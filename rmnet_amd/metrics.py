# -*- coding: utf-8 -*-
"""Region similarity J (Jaccard index) evaluated where the label maps live.

Restates ``utils/metrics.py:84-102`` of the reference (J = |seg & ann| / |seg | ann|, and 1 when both
are empty) with tensor ops that run on the GPU without a host synchronisation, so that a sharded
evaluation exchanges scalars over RCCL (``rmnet_amd.dist.sum_over_ranks``) instead of label maps.
"""

import torch


def jaccard_per_object(pred_labels, gt_labels, n_objects):
    """pred_labels, gt_labels: integer label maps [N, H, W] (0 = background) on any device.
    Returns J as float64 [N, n_objects] for objects 1..n_objects (device of the inputs, no sync)."""
    if pred_labels.shape != gt_labels.shape or pred_labels.dim() != 3:
        raise RuntimeError('expected two [N, H, W] label maps of the same shape')
    ids = torch.arange(1, n_objects + 1, device=pred_labels.device).view(1, -1, 1, 1)
    seg = pred_labels.unsqueeze(1) == ids
    ann = gt_labels.unsqueeze(1) == ids
    inter = (seg & ann).flatten(2).sum(2).to(torch.float64)
    union = (seg | ann).flatten(2).sum(2).to(torch.float64)
    return torch.where(union == 0, torch.ones_like(union), inter / union.clamp(min=1))


def mean_jaccard(pred_labels, gt_labels, n_objects, skip_first_and_last=True):
    """Mean J over objects and frames; DAVIS convention drops the first (given) and the last frame."""
    j = jaccard_per_object(pred_labels, gt_labels, n_objects)
    if skip_first_and_last and j.shape[0] > 2:
        j = j[1:-1]
    return j.mean()
