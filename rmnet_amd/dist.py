# -*- coding: utf-8 -*-
"""Per-video data parallelism over the GPUs of one node.

Clips are independent (frame t needs the mask of frame t-1, but clips share nothing --
models/rmnet.py:410-450), so the path shards across videos: one process per GPU, video v goes to a
rank chosen by a longest-first greedy balance of its cost N*n_objects, weights are replicated and
there is NO communication while a clip runs.  The only collectives are a 32-byte-per-video header
all_gather and ONE ``gather`` of the per-video outputs (uint8 label maps) to rank 0 at the end, which on
MI355X uses each peer's direct xGMI link (the backend is "nccl" = RCCL on ROCm; "gloo" on CPU for the
tests).

The reference has no counterpart: its DataParallel wrapper runs batch size 1 on one GPU
(core/inference.py:23-37) and utils/eval_server.py:78-87 shards checkpoints, not videos.
"""

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force_group=False):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank).  A single process without
    those variables is rank 0 of 1 and no process group is created -- unless ``force_group`` asks for a
    ONE-rank group (the collectives below then really go through the backend: how the RCCL branch is
    executed on a box with one GPU)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force_group) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            if world > 1:
                os.environ['MASTER_PORT'] = '29500'
            else:                                       # a one-rank group: any free port will do
                import socket
                with socket.socket() as sk:
                    sk.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    return rank, world, local


def assign_videos(costs, world_size):
    """Longest-first greedy assignment.  ``costs[v]`` ~ N_frames * n_objects of video v.
    Returns a list ``owner[v]`` of ranks; deterministic, identical on every rank."""
    order = sorted(range(len(costs)), key=lambda v: (-costs[v], v))
    load = [0] * world_size
    owner = [0] * len(costs)
    for v in order:
        r = min(range(world_size), key=lambda i: (load[i], i))
        owner[v] = r
        load[r] += costs[v]
    return owner


def my_videos(costs, rank, world_size):
    return [v for v, r in enumerate(assign_videos(costs, world_size)) if r == rank]


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX all-reduce of a Python float (used for the step timing of bench.py)."""
    if not dist.is_initialized():
        return float(value)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' \
            else torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' \
            else torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_label_maps(results, n_videos, dst=0):
    """Gather per-video label maps on rank ``dst``.

    ``results``: dict video_id -> uint8 tensor [N,H,W] (this rank's videos; on the rank's device for
    nccl, CPU for gloo).  Two phases: (1) all_gather of a fixed-size int64 header table
    [n_videos, 4] = (owner_has_it, N, H, W) -- 32 bytes per video, every rank learns every size (one host
    read of the whole table, not one per entry); (2) ONE ``gather`` of the ranks' payloads to ``dst``,
    padded to the largest rank payload (flat uint8): RCCL has no variable-size gather; each peer sends
    its block over its own xGMI link to ``dst`` and nothing is sent to the other ranks (an all_gather
    would move world_size times the bytes).  Returns {video_id: tensor} on ``dst`` and {} elsewhere."""
    if not dist.is_initialized():
        return dict(results)
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dist.get_backend() == 'nccl'
    dev = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')      # where the collectives run
    # where the payload is PACKED: on the GPU whenever the label maps live there (one flat buffer, one D2H copy for gloo,
    # none for RCCL) -- the same packing code under both backends
    pack_dev = next((t.device for t in results.values() if t.is_cuda), dev)
    header = torch.zeros(n_videos, 4, dtype=torch.int64)
    for v, t in results.items():
        header[v] = torch.tensor([1, t.shape[0], t.shape[1], t.shape[2]], dtype=torch.int64)
    header = header.to(dev)
    headers = [torch.zeros_like(header) for _ in range(world)]
    dist.all_gather(headers, header)
    table = torch.stack(headers).cpu()                               # [world, n_videos, 4], one D2H copy
    sizes = (table[:, :, 0] * table[:, :, 1] * table[:, :, 2] * table[:, :, 3]).sum(dim=1).tolist()
    cap = max(max(sizes), 1)
    payload = torch.zeros(cap, dtype=torch.uint8, device=pack_dev)
    at = 0
    for v in sorted(results):
        flat = results[v].to(pack_dev).reshape(-1)
        payload[at:at + flat.numel()] = flat
        at += flat.numel()
    payload = payload.to(dev)
    payloads = [torch.zeros_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, payloads, dst=dst)
    out = {}
    if rank == dst:
        for r in range(world):
            at = 0
            for v in range(n_videos):
                has, n, hh, ww = (int(x) for x in table[r, v])
                if has:
                    out[v] = payloads[r][at:at + n * hh * ww].reshape(n, hh, ww).clone()
                    at += n * hh * ww
    return out
