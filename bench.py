# -*- coding: utf-8 -*-
"""bench.py -- frames/sec of RMNet's per-frame inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], synthetic): 480x854 clips, 1 object each (K = 2 mask channels),
memory pinned at T = 5 frames (4 committed + the tentative previous frame); ``--clips-per-gpu`` (default
4) independent clips are batched on every GPU -- videos share nothing, and a single 480p stream cannot
fill 256 CUs (measured: 147 frames/s with 1 clip, 176 / 196 / 209 with 2 / 4 / 8).  One "step" = one frame of the
reference's loop (models/rmnet.py:410-450, utils/helpers.py:55): TinyFlowNet on the frame pair,
memorise frame t-1 (ResNet-50 memory encoder + KV head + region boxes + bank write), regional query
boxes from the flow-warped previous mask, query encoder + KV head, fused regional memory read,
decoder, soft aggregation, soft-max.  fp32 everywhere (the reference's dtype).  Inputs are resident
in HBM before the timed region.  With N ranks every rank runs its own clips (videos are independent;
weak scaling, no data-path collective) and ``value`` = N * clips * K / max-over-ranks time.

The JSON line also carries
  roofline     -- the dominant hand-written kernel (bk_main, the regional memory read) timed live
                  with HIP events recorded on its own stream around every launch of the timed
                  region: achieved = algorithmic bytes per launch / mean duration, vs 8 TB/s HBM;
  cpu_baseline -- the oracle's CPU restatement of the same path timed on this box's host cores
                  (rank 0, N = 1 only; bounded sample).
"""

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, K_CH, T_MEM = 480, 854, 2, 5
DE, DO = 128, 512
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(no, T, h, w):
    """4*[(De+Do)*T*h*w + (De+Do)*h*w + 2*Do*h*w] per object-frame."""
    hw = h * w
    return no * 4 * ((DE + DO) * T * hw + (DE + DO) * hw + 2 * DO * hw)


class HipEvents:
    """Raw hipEvent_t handles (the C ABI records them on the kernel's own stream)."""

    def __init__(self, n):
        self.hip = ctypes.CDLL('libamdhip64.so')
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.ev = []
        for _ in range(n):
            e = ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
            self.ev.append(e.value)

    def floor_us(self, stream, reps=50):
        """Elapsed time between two event records with NOTHING in between on ``stream``: the fixed
        cost of the event bracket itself (barrier packets + timestamps), to be subtracted from a
        bracketed kernel."""
        import torch
        vals = []
        for _ in range(reps):
            assert self.hip.hipEventRecord(ctypes.c_void_p(self.ev[0]), ctypes.c_void_p(stream)) == 0
            assert self.hip.hipEventRecord(ctypes.c_void_p(self.ev[1]), ctypes.c_void_p(stream)) == 0
            torch.cuda.synchronize()
            vals.append(self.elapsed_ms(self.ev[0], self.ev[1]) * 1e3)
        vals.sort()
        return vals[len(vals) // 2]

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), ctypes.c_void_p(a), ctypes.c_void_p(b))
        assert rc == 0, rc
        return float(ms.value)


def cpu_baseline(n_frames=5):
    """Oracle CPU path (plain torch + C ops) on a bounded sample: one 480x854 clip, 1 object,
    memorize_every = 1 so the memory grows 1..n_frames (mean T = 3 for 5 frames -- slightly cheaper
    than the pinned T = 5, i.e. generous to the CPU)."""
    from oracle import oracle
    from rmnet_amd import networks
    from rmnet_amd.synthetic import synthetic_clip
    from rmnet_amd.tiny_flownet import TinyFlowNet
    torch.set_grad_enabled(False)
    net = networks.procedural_init_(oracle.OracleRMNet(reader='torch')).eval()
    tfn = networks.procedural_init_(TinyFlowNet(None)).eval()
    frames, masks, _, n_objects = synthetic_clip(n_frames + 1, K_CH, H, W, seed=0, size=2.1)
    warm = frames[:, :2]
    net(warm, masks[:, :2], tfn(warm), n_objects[:, :2], 1)            # warm-up (1 frame)
    t0 = time.perf_counter()
    flows = tfn(frames)
    net(frames, masks, flows, n_objects, 1)
    dt = time.perf_counter() - t0
    return {'value': round(n_frames / dt, 4), 'unit': 'frames/s', 'cores': torch.get_num_threads(),
            'kind': 'port',
            'sample': '%d frames of one 480x854 clip, 1 object, memorize_every=1 (T=1..%d), TinyFlowNet '
                      'included, fp32, torch %d threads; %.1f s' % (n_frames, n_frames, torch.get_num_threads(), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-miopen-find', action='store_true',
                    help='leave torch.backends.cudnn.benchmark off (MIOpen immediate mode; ~5 %% slower convs)')
    ap.add_argument('--no-fuse-epilogue', action='store_true',
                    help='keep BatchNorm / bias / skip add / ReLU as separate torch kernels '
                         '(default: one rmnet_channel_affine_f32 pass per convolution)')
    ap.add_argument('--fold-bn', action='store_true',
                    help='fold eval-mode BatchNorm into the trunk convolutions (measured: no gain at 4 clips/GPU)')
    ap.add_argument('--channels-last', action='store_true', help='conv stacks in NHWC memory format (experiment)')
    ap.add_argument('--clips-per-gpu', type=int, default=8,
                    help='independent clips batched on every GPU (one 480p clip cannot fill 256 CUs; measured '
                         'on MI355X: 164 / 194 / 227 / 248 / 247 frames/s at 1 / 2 / 4 / 8 / 10 clips)')
    ap.add_argument('--graph', action='store_true', help='replay the frame step as one captured HIP graph')
    ap.add_argument('--dist-backend', default=None, help="override the process-group backend ('gloo' lets several "
                    "ranks share one GPU when testing the N>1 path on a 1-GPU box)")
    args = ap.parse_args()

    from rmnet_amd import dist as rd
    from rmnet_amd import networks
    from rmnet_amd.rmnet import RMNet
    from rmnet_amd.synthetic import synthetic_clip
    from rmnet_amd.tiny_flownet import TinyFlowNet

    rank, world, local = rd.init_from_env(args.dist_backend)
    if args.dist_backend == 'gloo':
        local = local % max(torch.cuda.device_count(), 1)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    torch.backends.cudnn.benchmark = not args.no_miopen_find

    net = networks.procedural_init_(RMNet(None)).to(dev).eval()
    tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev).eval()
    if args.fold_bn:
        net.fuse_for_inference()
    elif not args.no_fuse_epilogue and not args.channels_last:
        net.fuse_epilogues()
        tfn.fuse_epilogues()
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)
        tfn = tfn.to(memory_format=torch.channels_last)
    n_clip = 12
    B = max(1, args.clips_per_gpu)
    # every rank its own clip(s); size=2.1: object ~18 % of the frame, regional boxes ~46 % of the cells
    # 
    clips = [synthetic_clip(n_clip, K_CH, H, W, seed=rank * 64 + c, size=2.1) for c in range(B)]
    frames = torch.cat([c[0] for c in clips]).to(dev)
    masks = torch.cat([c[1] for c in clips]).to(dev).float()

    ctx = net._ClipContext(net, B, K_CH, H, W, [K_CH - 1] * B, dev)
    bank = net.new_bank(ctx, T_MEM)
    for t in range(1, T_MEM):                                             # fill 4 committed frames
        flow = tfn._forward(frames[:, t], frames[:, t - 1])
        net.frame_step(ctx, bank, frames[:, t - 1], masks[:, t - 1], frames[:, t], flow, commit=True)
    assert bank.committed == T_MEM - 1

    events = HipEvents(3 * args.steps)
    ev_floor_us = events.floor_us(torch.cuda.current_stream(dev).cuda_stream)

    def frame_body(prev_frame, prev_mask, cur_frame):
        # one frame of the loop; all of the step's work, incl. the final soft-max, is done
        flow = tfn._forward(cur_frame, prev_frame)
        out = net.frame_step(ctx, bank, prev_frame, prev_mask, cur_frame, flow, commit=False)
        if isinstance(out, tuple):          # fused decoder tail: (logits, soft-max of the logits)
            return out[1]
        return torch.softmax(out, dim=1)

    def eager_step(i, ev=None):
        # frames cycle through the clip; the mask fed back is the synthetic blob of frame t-1 (with
        # random-init weights the prediction itself is meaningless and would drive the boxes to
        # degenerate sizes)
        t = T_MEM + (i % (n_clip - T_MEM))
        net._profile_events = ev
        return frame_body(frames[:, t - 1], masks[:, t - 1], frames[:, t])

    step = eager_step
    if args.graph:
        # The frame step has no host synchronisation (boxes, rectangles, split plan all stay on the
        # device), so it is captured once and replayed; inputs go through static buffers.
        s_prev, s_mask, s_cur = frames[:, T_MEM - 1].clone(), masks[:, T_MEM - 1].clone(), frames[:, T_MEM].clone()
        net._profile_events = None
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                frame_body(s_prev, s_mask, s_cur)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            s_out = frame_body(s_prev, s_mask, s_cur)

        def step(i, ev=None):
            t = T_MEM + (i % (n_clip - T_MEM))
            s_prev.copy_(frames[:, t - 1])
            s_mask.copy_(masks[:, t - 1])
            s_cur.copy_(frames[:, t])
            graph.replay()
            return s_out

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    rd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i, tuple(events.ev[3 * i:3 * i + 3]))
    torch.cuda.synchronize()
    rd.barrier()
    torch.cuda.synchronize()
    elapsed = rd.max_over_ranks(time.perf_counter() - t0)
    assert bool(torch.isfinite(out).all())
    if args.graph:
        # events cannot be read back from inside a replayed graph: time the kernel on the same state
        # with an eager pass right after the timed region
        for i in range(args.steps):
            eager_step(i, tuple(events.ev[3 * i:3 * i + 3]))
        torch.cuda.synchronize()
    net._profile_events = None

    main_ms = [events.elapsed_ms(events.ev[3 * i], events.ev[3 * i + 1]) for i in range(args.steps)]
    comb_ms = [events.elapsed_ms(events.ev[3 * i + 1], events.ev[3 * i + 2]) for i in range(args.steps)]
    main_raw = sum(main_ms) / len(main_ms)
    main_avg = max(main_raw - ev_floor_us * 1e-3, 1e-6)     # kernel time = bracket - empty-bracket floor
    abytes = algorithmic_bytes(B * (K_CH - 1), T_MEM, ctx.h, ctx.w)
    achieved = abytes / (main_avg * 1e-3) / 1e9
    # executed MFMA work of the same launches: 3 split-fp16 terms over the COMPACTED tiles
    areas = bank.areas()[:, :T_MEM].cpu()
    njt = ((areas + 31) // 32).sum(dim=1)                              # 32-cell tiles per object
    lw_, _, lh_, _ = __import__('rmnet_amd.helpers', fromlist=['pad_amounts']).pad_amounts(H, W, 16)
    t_last = T_MEM + ((args.steps - 1) % (n_clip - T_MEM))
    from rmnet_amd import ops as _ops
    _, _, qr = _ops.region_map(net.warp(masks[:, t_last - 1], tfn._forward(frames[:, t_last], frames[:, t_last - 1]))[0].contiguous(),
                               want_map=False, cell_grid=(lw_, lh_, 16, ctx.h, ctx.w))
    qr = qr[:, 1].cpu()
    mq = (qr[:, 1] - qr[:, 0] + 1).clamp(min=0) * (qr[:, 3] - qr[:, 2] + 1).clamp(min=0)
    nqt = (mq + 1 + 63) // 64
    mfma_flops = float((nqt * 64 * njt * 32).sum()) * (DE + DO) * 2 * 3
    mfma_tflops = mfma_flops / (main_avg * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'mr_main_hbm_traffic.json')   # from a separate --pmc pass
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if int(tj.get('algorithmic_bytes_per_launch', -1)) == int(abytes):   # measured for this very workload
            traffic = tj.get('hbm_bytes_per_launch')

    if rank == 0:
        line = {
            'metric': 'frames/sec at 480p, 1 object, T=5 memory; memory-read HBM GB/s vs peak',
            'value': round(world * B * args.steps / elapsed, 3), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (convs fp32; memory read = split-fp16 MFMA hi*hi+hi*lo+lo*hi with fp32 accumulate, fp32-class accuracy)',
            'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: 480x854 synthetic clips, 1 object each (K=2), memory pinned '
                                   'at T=5, TinyFlowNet + memorize + regional read + decoder per frame; '
                                   'clips_per_gpu independent clips batched per GPU',
                       'weights': 'procedural random-init (no checkpoint offline)',
                       'prev_mask': 'synthetic blob mask of frame t-1 (object ~18 % of the frame, boxes ~46 % of the cells)',
                       'sharding': 'one clip per rank',
                       'miopen_find': not args.no_miopen_find, 'channels_last': bool(args.channels_last),
                       'hip_graph': bool(args.graph), 'clips_per_gpu': B,
                       'batchnorm_folded': bool(args.fold_bn),
                       'fused_epilogues': bool(not args.fold_bn and not args.no_fuse_epilogue and not args.channels_last)},
            'roofline': {'bound': 'hbm', 'kernel': 'bk_main (fused regional memory read, split-fp16 MFMA bank kernel)',
                         'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'algorithmic_bytes_per_launch': abytes, 'launches': args.steps,
                         'avg_us': round(main_avg * 1e3, 2), 'avg_us_event_bracket': round(main_raw * 1e3, 2),
                         'event_floor_us': round(ev_floor_us, 2), 'min_us_event_bracket': round(min(main_ms) * 1e3, 2),
                         'combine_avg_us_event_bracket': round(sum(comb_ms) / len(comb_ms) * 1e3, 2),
                         'mfma': {'executed_tflops': round(mfma_tflops, 1), 'peak_f16_dense_tflops': 2500.0,
                                  'frac': round(mfma_tflops / 2500.0, 4),
                                  'note': 'v_mfma_f32_16x16x32_f16, three split-fp16 terms (hi*hi+hi*lo+lo*hi) over the '
                                          'compacted 64-query x 32-cell tiles actually executed'},
                         'timing': 'hipEventRecord on the launch stream around every bk_main of the timed region; '
                                   'avg_us = bracket mean minus the empty-bracket floor measured the same way'},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == '__main__':
    main()
