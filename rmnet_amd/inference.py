# -*- coding: utf-8 -*-
"""Sharded per-video inference (BASELINE configs[3]).

The reference segments its test set one clip at a time on one GPU (core/inference.py:49-75: flows from
TinyFlowNet, RMNet.forward, argmax, save).  Clips are independent, so here every rank takes the clips
``rmnet_amd.dist.assign_videos`` gives it (longest-first greedy by frames x objects), runs the
device-resident frame loop on them, and only the uint8 label maps -- or, with ground truth, the J
scalars -- cross RCCL at the end.  No communication while a clip runs.
"""

import torch

from . import dist as rd
from . import metrics


def video_costs(videos):
    """Cost of a clip ~ frames x objects (what the frame loop's time is proportional to)."""
    return [int(v['frames'].shape[0]) * max(int(v['n_objects']), 1) for v in videos]


@torch.no_grad()
def segment_video(net, flownet, video, memorize_every=5):
    """One clip: ``video`` = {'frames' [N,3,H,W] f32 normalised, 'masks' [N,K,H,W] one-hot (only frame 0
    is read), 'n_objects' int}.  Returns the label maps uint8 [N,H,W] on the model's device
    (core/inference.py:52-63 without the file output)."""
    frames = video['frames'].unsqueeze(0)
    masks = video['masks'].unsqueeze(0)
    N = frames.shape[1]
    n_objects = torch.full((1, N), int(video['n_objects']), dtype=torch.long)
    dev = next(net.parameters()).device
    frames = frames.to(dev, non_blocking=True)
    flows = flownet(frames)
    est = net(frames, masks, flows, n_objects, memorize_every, device=dev)
    return est[0].argmax(dim=1).to(torch.uint8)


def segment_videos(videos, segment_fn, rank=None, world_size=None, gather=True, dst=0):
    """Run ``segment_fn(video) -> uint8 [N,H,W]`` on this rank's share of ``videos`` (a list, identical
    on every rank) and gather the label maps on rank ``dst``.  Returns {video index: label maps} on
    ``dst`` (every rank's own share when ``gather`` is False or no process group exists)."""
    if rank is None or world_size is None:
        rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        world_size = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    mine = rd.my_videos(video_costs(videos), rank, world_size)
    results = {v: segment_fn(videos[v]) for v in mine}
    if not gather:
        return results
    return rd.gather_label_maps(results, len(videos), dst=dst)


def evaluate_videos(videos, segment_fn, rank=None, world_size=None):
    """Mean region similarity J over all clips and objects (utils/metrics.py:84-102; first and last
    frame of a clip skipped as in DAVIS) with ground truth ``video['labels']`` uint8 [N,H,W]: every
    rank scores its own clips on its device and two scalars are all-reduced.  Returns a float on every
    rank."""
    if rank is None or world_size is None:
        rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        world_size = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    mine = rd.my_videos(video_costs(videos), rank, world_size)
    total, count = 0.0, 0.0
    for v in mine:
        pred = segment_fn(videos[v])
        gt = videos[v]['labels'].to(pred.device)
        j = metrics.jaccard_per_object(pred.long(), gt.long(), int(videos[v]['n_objects']))
        if j.shape[0] > 2:
            j = j[1:-1]
        total += float(j.sum())
        count += float(j.numel())
    total = rd.sum_over_ranks(total)
    count = rd.sum_over_ranks(count)
    return total / max(count, 1.0)
