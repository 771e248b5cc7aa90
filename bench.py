# -*- coding: utf-8 -*-
"""bench.py -- frames/sec of RMNet's per-frame inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: started from a plain shell (WORLD_SIZE unset) bench.py re-executes itself as N ranks under torch.distributed.run (one
process per GPU, RCCL over xGMI, rendezvous on 127.0.0.1 at a free port) and rank 0 prints the one JSON line; started BY
torch.distributed.run (the driver's form) it is one of those ranks.

Workload (BASELINE.json configs[1], synthetic): 480x854 clips, 1 object each (K = 2 mask channels), memory pinned at T = 5
frames (4 committed + the tentative previous frame); ``--clips-per-gpu`` (default 16; 8 until round 4) independent clips are batched on
every GPU -- videos share nothing, and a single 480p stream cannot fill 256 CUs (measured on MI355X: 164 / 227 / 254 / 261 / 261 / 267
frames/s at 1 / 4 / 8 / 12 / 16 / 32 clips; at 16 the read's launch is 208 workgroups of one (object, query tile) pair each: no merge).  One "step" = one frame of the reference's loop (models/rmnet.py:410-450, utils/helpers.py:55):
TinyFlowNet on the frame pair, memorise frame t-1 (ResNet-50 memory encoder + KV head + region boxes + bank write), regional
query boxes from the flow-warped previous mask, query encoder + KV head, fused regional memory read, decoder, soft
aggregation, soft-max.  Convolutions fp32 (the reference's dtype); the arithmetic of the memory read is ``--read-precision``
('auto' = RMNet's default: fp16 operands for one object per clip, i.e. this workload -- `dtype` says what ran and what its
parity evidence is).  Inputs are resident in HBM before the timed region.  The frames/s figure is governed by the fp32 MIOpen
convolutions (85 % of a step's GPU time; the hand-written kernels 9 %): it moves with them, not with the read kernel.  With N
ranks every rank runs its own clips (videos are independent; weak scaling, no data-path collective) and ``value`` =
N * clips * K / max-over-ranks time.

The JSON line also carries
  roofline     -- the dominant hand-written kernel, bk_main = the WHOLE regional memory read in one launch (soft-max read of
                  the bank, merge of the partial results, masked cells, the q_val half of the cat), timed live with HIP
                  events recorded on its own stream around every launch of the timed region.  achieved / peak / frac are
                  SURVEY.md section 8d's figure: algorithmic bytes per launch / mean duration vs 8 TB/s HBM.  What bounds
                  the kernel is latency (fp16-operand modes: matrix pipe ~20 % busy) resp. the matrix pipe under the power cap
                  (split mode); roofline.mfma has the pipe figures, roofline.modes the same launches in
                  the other arithmetic modes (split = fp32-class; qx; f16);
  extras       -- single-stream (eager and HIP-graph replay) and free-running (memorize_every = 5, N = 67, fed-back masks)
                  fps; whole-loop frames/s for BASELINE configs[2] (5 objects), configs[4] (720p, 3 objects, T = 20) and the
                  loader's K = 11 channel count; kernel figures for the hand-written kernels, each against SURVEY.md section
                  8d's byte formulas (rank 0, N = 1 only; taken in a child process);
  cpu_baseline -- the oracle's CPU restatement of the same path timed on this box's host cores (rank 0, N = 1 only; bounded
                  sample): BASELINE configs[0] (T = 3, N = 4) over a sweep of thread counts, value = the best one; the T = 5
                  pinned shape and the three native ops alone at the best count and at 8 threads.
"""

import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, K_CH, T_MEM = 480, 854, 2, 5
DEFAULT_CLIPS = 16            # clips batched per GPU in the default run (--clips-per-gpu)
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


def _median_time(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_baseline(n_frames=3):
    """Oracle CPU path (plain torch + C ops, oracle/) timed on this box's host cores, bounded sample.

    ``value`` = the TWIN of the GPU line's workload (BASELINE configs[1]'s shape): ONE 480x854 stream, 1 object, memory PINNED at
    T = 5 (4 committed frames pre-filled untimed, every timed frame = TinyFlowNet + memorize + cat -> T = 5 + warp / boxes + segment +
    soft-max), at the BEST thread count of a sweep over {8, 16, 32, 64, 128} (those that the box has).  The sweep itself runs
    BASELINE.md section 3's CPU leg = BASELINE configs[0] (one clip, N = 4 frames, memorize_every = 1: the memory grows to T = 3,
    TinyFlowNet + ``OracleRMNet.forward``) and is kept in ``configs0``; also: the three native ops alone."""
    import numpy as np
    import torch.nn.functional as F
    from oracle import oracle
    from rmnet_amd import networks
    from rmnet_amd.synthetic import synthetic_clip
    from rmnet_amd.tiny_flownet import TinyFlowNet
    torch.set_grad_enabled(False)
    all_threads = torch.get_num_threads()
    net = networks.procedural_init_(oracle.OracleRMNet(reader='torch')).eval()
    tfn = networks.procedural_init_(TinyFlowNet(None)).eval()

    def set_threads(nt):
        torch.set_num_threads(nt)
        oracle.set_num_threads(nt)

    # ---- BASELINE configs[0]: N = 4, memorize_every = 1, T reaches 3 (3 segmented frames per pass)
    f0, m0, _, n0 = synthetic_clip(4, K_CH, H, W, seed=0, size=2.1)

    def cfg0_pass():
        t0 = time.perf_counter()
        est = net(f0, m0, tfn(f0), n0, 1)
        dt = time.perf_counter() - t0
        assert bool(torch.isfinite(est).all())
        return dt
    counts = sorted({n for n in (8, 16, 32, 64, 128) if n <= all_threads} | {min(8, all_threads)})
    sweep = {}
    for nt in counts:
        set_threads(nt)
        net(f0[:, :2], m0[:, :2], tfn(f0[:, :2]), n0[:, :2], 1)    # warm-up: one segmented frame at this thread count
        sweep[nt] = round(3.0 / min(cfg0_pass(), cfg0_pass()), 4)
    best = max(sweep, key=lambda n: sweep[n])

    # ---- the GPU workload's own shape: memory pinned at T = 5
    frames, masks, _, n_objects = synthetic_clip(T_MEM + n_frames, K_CH, H, W, seed=0, size=2.1)
    masks = masks.float()
    keys = values = None
    set_threads(best)
    for t in range(T_MEM - 1):                                  # untimed: the 4 committed frames
        pk, pv, _ = net.memorize(frames[:, t], masks[:, t], [K_CH - 1])
        keys = pk if keys is None else torch.cat([keys, pk], dim=3)
        values = pv if values is None else torch.cat([values, pv], dim=3)

    def one_frame(t):
        flow = tfn._forward(frames[:, t], frames[:, t - 1])
        pk, pv, _ = net.memorize(frames[:, t - 1], masks[:, t - 1], [K_CH - 1])
        tk, tv = torch.cat([keys, pk], dim=3), torch.cat([values, pv], dim=3)      # T = 5
        att, _ = net.get_att_map(masks[:, t - 1], flow)
        return F.softmax(net.segment(frames[:, t], att, tk, tv, [K_CH - 1]), dim=1)

    def timed(nt):
        set_threads(nt)
        one_frame(T_MEM - 1)                                    # warm-up
        t0 = time.perf_counter()
        for i in range(n_frames):
            one_frame(T_MEM + i)
        return time.perf_counter() - t0
    dt_best = timed(best)
    dt8 = timed(min(8, all_threads)) if best != min(8, all_threads) else dt_best

    # ---- the native ops alone
    g = torch.Generator().manual_seed(0)
    h, w = 30, 54
    mk = torch.randn(1, DE, T_MEM, h, w, generator=g) * 0.6
    mv = torch.randn(1, DO, T_MEM, h, w, generator=g)
    qk = torch.randn(1, DE, h, w, generator=g) * 0.6
    qv = torch.randn(1, DO, h, w, generator=g)
    soft = np.zeros((1, K_CH, 480, 864), np.float32)
    soft[0, 1, 120:330, 250:600] = 0.9
    rng = np.random.RandomState(0)
    flow = ((rng.rand(H, W, 2) - 0.5) * 20).astype(np.float32)
    m1 = (np.eye(2, 3) + (rng.rand(2, 3) - 0.5) * 0.1).astype(np.float32)
    m2 = (np.eye(2, 3) + (rng.rand(2, 3) - 0.5) * 0.1).astype(np.float32)
    ops = {}
    for label, nt in (('best_threads', best), ('8_threads', min(8, all_threads))):
        set_threads(nt)
        oracle.torch_memory_read(mk, mv, qk, qv)
        ops[label] = {
            'threads': nt,
            'memory_read_T5_480p_s': round(_median_time(lambda: oracle.torch_memory_read(mk, mv, qk, qv), 3), 4),
            'region_map_K2_480x864_s': round(_median_time(lambda: oracle.region_map(soft), 5), 5),
        }
    set_threads(all_threads)
    ops['flow_affine_480x854_1thread_s'] = round(_median_time(lambda: oracle.flow_affine(flow, m1, m2), 5), 5)
    ops['note'] = ('memory_read = models/rmnet.py:147-165 with torch CPU ops (the survey measured 0.28 s for the reference '
                   'itself on 8 vCPUs); region_map / flow_affine = oracle/rmnet_oracle.c (the reference runs flow_affine '
                   'single-threaded in DataLoader workers)')
    # `value` is the TWIN of the GPU line's workload (round-5 verdict): one 480x854 stream, 1 object, memory pinned at T = 5 -- the
    # same frame the GPU times, on the host cores -- at the best thread count of the sweep.  BASELINE configs[0] (the reference's own
    # CPU-runnable case: T grows to 3, N = 4) stays beside it.
    return {'value': round(n_frames / dt_best, 4), 'unit': 'frames/s', 'cores': best, 'threads_best': best, 'kind': 'port',
            'host_threads_available': all_threads,
            'value_8_threads': round(n_frames / dt8, 4),
            'sample': 'the GPU workload\'s own shape on the CPU (BASELINE configs[1]): one synthetic 480x854 stream, 1 object, memory PINNED at '
                      'T=%d (4 committed frames pre-filled untimed), %d timed frames of TinyFlowNet + memorize + cat + warp / boxes + segment + '
                      'soft-max (oracle.OracleRMNet, fp32), one warm-up frame; thread count = the best of the configs[0] sweep below' % (T_MEM, n_frames),
            'configs0': {'fps_best_threads': sweep[best], 'fps_8_threads': sweep[min(8, all_threads)],
                         'sweep': {str(n): v for n, v in sweep.items()},
                         'note': 'BASELINE configs[0] (BASELINE.md section 3): one synthetic 480x854 clip, 1 object, N=4 frames, memorize_every=1 '
                                 '(memory grows to T=3), TinyFlowNet + OracleRMNet.forward; best of two passes (3 segmented frames each) per thread '
                                 'count, warm-up of one frame per count'},
            'ops': ops}


def source_hash():
    """sha256 over the read kernels' sources: stamps profiles/*_hbm_traffic.json so that a PMC figure taken
    with an older kernel is never reported for a newer one."""
    hsh = hashlib.sha256()
    for fn in ('bank.hip', 'memory_read.hip', 'common.h'):     # (whitespace-insensitive since round 5: indentation is not the kernel)
        for ln in open(os.path.join(ROOT, 'rmnet_amd', 'csrc', fn), 'rb').read().split(b'\n'):
            if ln.strip():
                hsh.update(ln.strip() + b'\n')
    return hsh.hexdigest()[:16]


def kernel_figures(dev, events):
    """Hand-written kernels alone (HIP events on the launch stream, empty-bracket floor subtracted), each
    with SURVEY.md section 8d's algorithmic bytes: BASELINE configs[2] (5 objects) and configs[4] (720p,
    3 objects, T = 20) through the bank read, the region map, the flow update, the bank append."""
    import numpy as np
    from rmnet_amd import ops
    hip, ev = events.hip, events.ev
    st = torch.cuda.current_stream(dev).cuda_stream
    floor = events.floor_us(st)

    def bracket(fn, reps=12, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            hip.hipEventRecord(ctypes.c_void_p(ev[0]), ctypes.c_void_p(st))
            fn()
            hip.hipEventRecord(ctypes.c_void_p(ev[1]), ctypes.c_void_p(st))
            torch.cuda.synchronize()
            ts.append(events.elapsed_ms(ev[0], ev[1]) * 1e3 - floor)
        return float(np.mean(ts))

    def gbs(nbytes, us):
        return {'us': round(us, 2), 'algorithmic_bytes': int(nbytes), 'GBps': round(nbytes / us / 1e3, 1),
                'hbm_frac': round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}

    out = {}
    g = torch.Generator().manual_seed(1)
    rng = np.random.RandomState(2)
    for name, (no, T, h, w, dense) in (('cfg3_5obj_T5_480p_boxes46', (5, 5, 30, 54, False)),
                                       ('cfg5_3obj_T20_720p_boxes46', (3, 20, 45, 80, False)),
                                       ('cfg5_3obj_T20_720p_dense', (3, 20, 45, 80, True))):
        k = (torch.randn(no, DE, h, w, generator=g) * 0.6).to(dev)
        v = torch.randn(no, DO, h, w, generator=g).to(dev)
        rects = []
        for _ in range(no):
            rh, rw = int(h * 0.68), int(w * 0.68)
            y0, x0 = rng.randint(0, h - rh + 1), rng.randint(0, w - rw + 1)
            rects.append((0, w - 1, 0, h - 1) if dense else (x0, x0 + rw - 1, y0, y0 + rh - 1))
        r = torch.tensor(rects, dtype=torch.int32, device=dev)
        bank = ops.MemoryBank(no, T, h, w, dev)
        for t in range(T):
            bank.append(t, k, v, r)
        e3 = (ev[2], ev[3], ev[4])
        ab = algorithmic_bytes(no, T, h, w)
        for mode in ('split', 'f16', 'qx'):
            bank.precision = mode
            for _ in range(3):
                bank.read(T, k, v, r)
            torch.cuda.synchronize()
            mm, cc = [], []
            for _ in range(8):
                bank.read(T, k, v, r, events=e3)
                torch.cuda.synchronize()
                mm.append(events.elapsed_ms(e3[0], e3[1]) * 1e3 - floor)
                cc.append(events.elapsed_ms(e3[1], e3[2]) * 1e3 - floor)
            row = gbs(ab, float(np.mean(mm) + max(np.mean(cc), 0.0)))       # (one kernel = the whole op)
            if mode == 'split':
                out[name] = row
            else:
                out[name][mode + '_mode'] = {'us': row['us'], 'GBps': row['GBps'], 'hbm_frac': row['hbm_frac']}
        bank.precision = 'split'
        if name.startswith('cfg3'):
            out['bk_append_5obj_480p'] = gbs(2 * 4 * (DE + DO) * h * w * no, bracket(lambda: bank.append(T - 1, k, v, r)))
        del bank
    # region map: 2 * 4 * B * K * H * W + 16 * B * K bytes (mask read once, map written once)
    for K_ in (2, 11):
        m = torch.zeros(1, K_, 480, 864, device=dev)
        m[0, 1:, 120:330, 250:600] = 0.9
        out['region_map_K%d_480x864' % K_] = gbs(2 * 4 * K_ * 480 * 864 + 16 * K_, bracket(lambda: ops.region_map(m)))
        out['region_boxes_only_K%d_480x864' % K_] = gbs(4 * K_ * 480 * 864 + 16 * K_,
                                                        bracket(lambda: ops.region_map(m, want_map=False, cell_grid=(0, 0, 16, 30, 54))))
    m = torch.zeros(8, 2, 480, 854, device=dev)
    m[:, 1, 120:330, 250:600] = 0.9
    fl = torch.full((8, 2, 480, 854), -2.5, device=dev)
    out['region_boxes_warped_8x2_480x854'] = gbs(4 * 8 * (2 + 2) * 480 * 854 + 16 * 16,
                                                 bracket(lambda: ops.region_map(m, want_map=False, flow=fl, cell_grid=(5, 0, 16, 30, 54))))
    f = ((torch.rand(480, 854, 2, generator=g) - 0.5) * 20).to(dev)
    m1 = torch.tensor([[1.02, 0.01, 1.5], [-0.01, 0.98, -2.0]], device=dev)
    out['flow_affine_480x854'] = gbs(16 * 480 * 854 + 48, bracket(lambda: ops.flow_affine(f, m1, m1)))
    out['event_floor_us'] = round(floor, 2)
    return out


def loop_fps(net, tfn, dev, B, K, n_obj, H, W, T_mem, steps, seed0=100):
    """Frames/s of the whole per-frame loop (TinyFlowNet + memorize + regional boxes + read + decoder + soft-max) for
    another configuration: B clips of n_obj objects each in K mask channels, memory pinned at T_mem frames."""
    from rmnet_amd.synthetic import synthetic_clip
    n_clip = T_mem + 4
    clips = [synthetic_clip(n_clip, n_obj + 1, H, W, seed=seed0 + c, size=2.1 if n_obj == 1 else 1.1) for c in range(B)]
    frames = torch.cat([c[0] for c in clips]).to(dev)
    masks = torch.cat([c[1] for c in clips]).to(dev).float()
    if K > n_obj + 1:                                            # the reference's test loader: N_MAX_OBJECTS + 1 = 11 channels
        masks = torch.cat([masks, torch.zeros(B, n_clip, K - n_obj - 1, H, W, device=dev)], dim=2)
    ctx = net._ClipContext(net, B, K, H, W, [n_obj] * B, dev)
    bank = net.new_bank(ctx, T_mem)
    net._profile_events = None
    for t in range(1, T_mem):
        net.frame_step(ctx, bank, frames[:, t - 1], masks[:, t - 1], frames[:, t], tfn._forward(frames[:, t], frames[:, t - 1]), commit=True)

    def step(i):
        t = T_mem + (i % (n_clip - T_mem))
        out = net.frame_step(ctx, bank, frames[:, t - 1], masks[:, t - 1], frames[:, t], tfn._forward(frames[:, t], frames[:, t - 1]), commit=False)
        return out[1] if isinstance(out, tuple) else torch.softmax(out, dim=1)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    return {'fps': round(B * steps / dt, 2), 'ms_per_step': round(1e3 * dt / steps, 3), 'clips': B, 'objects_per_clip': n_obj,
            'mask_channels': K, 'frame': '%dx%d' % (H, W), 'memory_frames': T_mem, 'steps': steps}


_T_START = time.perf_counter()


def _phase(name):
    """Wall-clock log of the run's phases on stderr (the JSON line on stdout stays alone)."""
    if os.environ.get('RANK', '0') == '0':
        print('[bench %7.1f s] %s' % (time.perf_counter() - _T_START, name), file=sys.stderr, flush=True)


def _self_launch(n):
    """``python bench.py --gpus N`` without a launcher: re-execute this command line as N ranks under
    ``torch.distributed.run`` (rendezvous on 127.0.0.1 at a free port; HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only has
    dmabuf IPC, RCCL needs it) and pass the ranks' output through -- rank 0 prints the one JSON line.  Returns the exit code."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the single-stream / free-running / per-kernel figures')
    ap.add_argument('--extras-child', action='store_true',
                    help='(internal) compute ONLY the extras block and print it: the default run takes its extras from a child process '
                         'started this way, so that a fault in one of the many secondary configurations cannot take the headline line with it')
    ap.add_argument('--no-miopen-find', action='store_true',
                    help='leave torch.backends.cudnn.benchmark off (MIOpen immediate mode; ~5 %% slower convs)')
    ap.add_argument('--no-fuse-epilogue', action='store_true',
                    help='keep BatchNorm / bias / skip add / ReLU as separate torch kernels '
                         '(default: one rmnet_channel_affine_f32 pass per convolution)')
    ap.add_argument('--fold-bn', action='store_true',
                    help='fold eval-mode BatchNorm into the trunk convolutions (measured: no gain at 4 clips/GPU)')
    ap.add_argument('--nchw', dest='channels_last', action='store_false',
                    help='keep both networks in NCHW memory format (the layout of rounds 1-5).  Default since round 6: channels_last -- MIOpen NHWC '
                         'kernels without the layout transposes that wrap them on NCHW tensors, channels-last glue kernels: +7 %% frames/s '
                         '(profiles/r06_conv_layout.md)')
    ap.add_argument('--channels-last', dest='channels_last', action='store_true', help='(the default; kept for scripts that pass it)')
    ap.set_defaults(channels_last=True)
    ap.add_argument('--clips-per-gpu', type=int, default=DEFAULT_CLIPS,
                    help='independent clips batched on every GPU (one 480p clip cannot fill 256 CUs; measured on MI355X in round 5, '
                         'profiles/r05_c_plan_and_tail_experiments.md: 164 / 227 / 253.7 / 261.2 / 260.6 / 267.4 frames/s at 1 / 4 / 8 / 12 / 16 / 32 clips). '
                         '16 (default since round 5; 8 before) = 16 x 13 (object, query tile) pairs = 208 workgroups of the read, each walking its '
                         'pair\'s whole memory: no split, no partial results, no merge')
    ap.add_argument('--graph', action='store_true', help='replay the frame step as one captured HIP graph')
    ap.add_argument('--read-precision', choices=('auto', 'split', 'qx', 'f16'), default='auto',
                    help="arithmetic of the bank read in the timed region: 'auto' (default, = RMNet's default) picks 'f16' for clips with one "
                         "object -- this workload -- while the clip's largest logit stays small, 'split' otherwise and for clips with several (profiles/r06_iou_temperature.md); 'split' = fp16 hi/lo "
                         "pairs, three MFMA terms, fp32-class; 'f16' = fp16 operands, one term; 'qx' = 'f16' with an exact query (two terms for "
                         "the logits).  The other modes' kernels are timed on the same launches after the timed region (roofline.modes)")
    ap.add_argument('--dist-backend', default=None, help="override the process-group backend ('gloo' lets several "
                    "ranks share one GPU when testing the N>1 path on a 1-GPU box)")
    args = ap.parse_args()
    if args.channels_last:          # (read by ATen when the first convolution runs: channels_last tensors then go to MIOpen as NHWC)
        os.environ.setdefault('PYTORCH_MIOPEN_SUGGEST_NHWC', '1')

    from rmnet_amd import dist as rd
    from rmnet_amd import networks
    from rmnet_amd.rmnet import RMNet
    from rmnet_amd.synthetic import synthetic_clip
    from rmnet_amd.tiny_flownet import TinyFlowNet

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started from a plain shell (`python bench.py --gpus N`): become the launcher of N ranks of this same command line
        # (one process per GPU, RCCL over xGMI) and relay rank 0's JSON line
        sys.exit(_self_launch(args.gpus))
    # (an explicit --dist-backend with --gpus 1 creates a ONE-rank process group: the collectives of the N > 1 path then run through
    #  that backend -- RCCL for 'nccl' -- on the one GPU a test box has)
    rank, world, local = rd.init_from_env(args.dist_backend, force_group=args.dist_backend is not None)
    import torch.distributed as tdist
    grouped = tdist.is_initialized()
    if args.dist_backend == 'gloo':
        local = local % max(torch.cuda.device_count(), 1)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    torch.backends.cudnn.benchmark = not args.no_miopen_find

    _phase('imports done')
    net = networks.procedural_init_(RMNet(None, read_precision=args.read_precision)).to(dev).eval()
    requested_precision = args.read_precision
    args.read_precision = net.resolve_read_precision([K_CH - 1] * max(1, args.clips_per_gpu))   # what 'auto' means for one object per clip
    tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev).eval()
    if args.fold_bn:
        net.fuse_for_inference()
    elif not args.no_fuse_epilogue:
        net.fuse_epilogues()
        tfn.fuse_epilogues()
    if args.channels_last:          # [r6] the fused glue kernels have channels-last variants (csrc/epilogue.hip): the layout no longer costs the fusion
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

    events = HipEvents(max(3 * args.steps, 8))
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

    _phase('memory filled (MIOpen find of the headline shapes done)')
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
    elapsed_local = time.perf_counter() - t0
    elapsed = rd.max_over_ranks(elapsed_local)
    assert bool(torch.isfinite(out).all())
    _phase('timed region done')
    multi = None
    if grouped:
        # Outside the timed region: every rank's own rate, and the path's ONE collective step -- the gather of the per-clip
        # label maps to rank 0 (rmnet_amd.dist.gather_label_maps: header all_gather + one padded gather) -- timed
        # on the last frame's label maps (uint8 [1, H, W] per clip), payloads resident on the GPU.
        cdev = dev if tdist.get_backend() == 'nccl' else torch.device('cpu')
        mine_t = torch.tensor([float(elapsed_local)], dtype=torch.float64, device=cdev)
        all_t = [torch.zeros_like(mine_t) for _ in range(world)]
        tdist.all_gather(all_t, mine_t)
        times = [float(t.item()) for t in all_t]
        gather_ms, gather_err = None, None
        try:          # (the collective is the same call on every rank: an unsupported-op error is raised on all of them, before any traffic)
            labels = out.argmax(dim=1).to(torch.uint8)
            mine = {rank * B + c: labels[c:c + 1] for c in range(B)}
            rd.gather_label_maps(mine, world * B)                           # warm-up (connection set-up)
            torch.cuda.synchronize()
            rd.barrier()
            tg = time.perf_counter()
            got = rd.gather_label_maps(mine, world * B)
            torch.cuda.synchronize()
            gather_ms = round(rd.max_over_ranks(1e3 * (time.perf_counter() - tg)), 3)
            if rank == 0:
                assert sorted(got) == list(range(world * B)) and tuple(got[0].shape) == (1, H, W)
        except (RuntimeError, AssertionError) as exc:
            gather_err = repr(exc)[:300]
        multi = {'per_rank_fps': [round(B * args.steps / t, 2) for t in times],
                 'gather_ms': gather_ms, 'gather_error': gather_err, 'gather_bytes_per_rank': int(B * H * W),
                 'backend': tdist.get_backend(),
                 'note': 'per_rank_fps = clips x steps / that rank\'s own time of the timed region (value uses the max over ranks); '
                         'gather_ms = header all_gather + one padded gather of the last frame\'s uint8 label maps to rank 0, '
                         'outside the timed region (a whole clip moves N x as many bytes, once per clip)'}
    if args.graph:
        # events cannot be read back from inside a replayed graph: time the kernel on the same state
        # with an eager pass right after the timed region
        for i in range(args.steps):
            eager_step(i, tuple(events.ev[3 * i:3 * i + 3]))
        torch.cuda.synchronize()
    net._profile_events = None

    main_ms = [events.elapsed_ms(events.ev[3 * i], events.ev[3 * i + 1]) for i in range(args.steps)]
    comb_ms = [events.elapsed_ms(events.ev[3 * i + 1], events.ev[3 * i + 2]) for i in range(args.steps)]
    # the OTHER arithmetic mode of the same kernel on the same launches (same bank, same boxes), outside the timed region
    other_modes = [m for m in ('split', 'qx', 'f16') if m != args.read_precision]
    other_ms = None
    small_ms = {}
    if not args.no_extras and not args.extras_child and world == 1:     # (secondary figures: the one-GPU line only)
        other_ms = {}
        for om in other_modes:
            bank.precision = om
            for i in range(3):
                eager_step(i)
            for i in range(args.steps):
                eager_step(i, tuple(events.ev[3 * i:3 * i + 3]))
            torch.cuda.synchronize()
            net._profile_events = None
            other_ms[om] = [events.elapsed_ms(events.ev[3 * i], events.ev[3 * i + 2]) - 2e-3 * ev_floor_us for i in range(args.steps)]
        bank.precision = args.read_precision
        # ... and the timed arithmetic at OTHER launch sizes (object-frames per launch = clips per GPU here), each inside the same frame
        # loop on its own clips (MIOpen immediate mode for these extra shapes: the read does not depend on the convolutions'
        # algorithms, only on what they leave in the caches).  1 = the reference's operating point (core/inference.py:22-28: one clip
        # at a time); <= 8: every (object, query tile) pair is cut into column blocks and merged by its last arriver; 16: one workgroup
        # per pair; > 19: more pairs than workgroups -- the launch runs in ROUNDS (csrc/common.h: bank_round_chunk_len).
        find = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = False
        k_small = min(args.steps, 10)
        for n in (1, 4, 8, 12, 16, 20, 24, 32):
            if n == B:
                continue
            try:
                if n <= B:
                    fn, mn = frames[:n], masks[:n]
                else:
                    extra = [synthetic_clip(n_clip, K_CH, H, W, seed=rank * 64 + c, size=2.1) for c in range(B, n)]
                    fn = torch.cat([frames] + [c[0].to(dev) for c in extra])
                    mn = torch.cat([masks] + [c[1].to(dev).float() for c in extra])
                ctxn = net._ClipContext(net, n, K_CH, H, W, [K_CH - 1] * n, dev)
                bankn = net.new_bank(ctxn, T_MEM, precision=args.read_precision)
                for t in range(1, T_MEM):
                    net.frame_step(ctxn, bankn, fn[:, t - 1], mn[:, t - 1], fn[:, t], tfn._forward(fn[:, t], fn[:, t - 1]), commit=True)
                for i in range(3 + k_small):
                    t = T_MEM + (i % (n_clip - T_MEM))
                    net._profile_events = tuple(events.ev[3 * (i - 3):3 * (i - 3) + 3]) if i >= 3 else None
                    net.frame_step(ctxn, bankn, fn[:, t - 1], mn[:, t - 1], fn[:, t], tfn._forward(fn[:, t], fn[:, t - 1]), commit=False)
                torch.cuda.synchronize()
                net._profile_events = None
                small_ms[n] = [events.elapsed_ms(events.ev[3 * i], events.ev[3 * i + 2]) - 2e-3 * ev_floor_us for i in range(k_small)]
                del bankn, ctxn, fn, mn
            except RuntimeError as exc:      # (a secondary figure must not take the driver line down)
                small_ms[n] = repr(exc)[:200]
                torch.cuda.synchronize()
        torch.backends.cudnn.benchmark = find
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
    nqt = (mq + 63) // 64
    n_terms = {'split': 3, 'qx': 2, 'f16': 1}[args.read_precision]         # = the template argument of bk_main
    s_terms, pv_terms = {'split': (3, 3), 'qx': (2, 1), 'f16': (1, 1)}[args.read_precision]
    mfma_flops = float((nqt * 64 * njt * 32).sum()) * (DE * s_terms + DO * pv_terms) * 2   # executed: the mode's MFMA terms over the padded tiles
    mfma_tflops = mfma_flops / (main_avg * 1e-3) / 1e12
    useful_flops = float((mq * areas.sum(dim=1)).sum()) * (DE + DO) * 2          # one term over the un-padded regional cells
    useful_tflops = useful_flops / (main_avg * 1e-3) / 1e12
    comb_avg = max(sum(comb_ms) / len(comb_ms) - ev_floor_us * 1e-3, 1e-6)
    op_achieved = abytes / ((main_avg + comb_avg) * 1e-3) / 1e9
    traffic, traffic_note = None, 'no PMC file for this kernel version'
    tname = {'split': 'bk_main_hbm_traffic.json', 'f16': 'bk_main_f16_hbm_traffic.json', 'qx': 'bk_main_qx_hbm_traffic.json'}[args.read_precision]
    tpath = os.path.join(ROOT, 'profiles', tname)                        # from separate --pmc passes (tools/pmc_traffic.sh)
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get('source_hash') == source_hash() and int(tj.get('algorithmic_bytes_per_launch', -1)) == int(abytes):
            traffic = tj.get('hbm_bytes_per_launch')
            traffic_note = ('profiles/' + tname + ' (same kernel sources, same workload): FETCH_SIZE x 2 + WRITE_SIZE x 1, the '
                            'factors calibrated on known byte counts in profiles/r03_power_ceiling.md')
        else:
            traffic_note = 'profiles/' + tname + ' is from other kernel sources or another workload: not reported'

    from rmnet_amd import rmnet as _rmnet_mod
    AUTO_BOUND = _rmnet_mod.AUTO_LOGIT_BOUND
    logit_max = bank.logit_max()             # (one host sync, outside the timed region) the quantity 'auto' decides on
    extras = None
    if rank == 0 and world == 1 and not args.no_extras and not args.extras_child:
        # The extras (single stream, HIP-graph replay, free-running clip, other configurations, every kernel alone, profiler shares) run
        # in a CHILD process: they launch a dozen secondary configurations, and a GPU fault in any of them -- round 4 met one in the
        # runtime's memset node under graph replay -- would otherwise kill this process before the headline line is printed.
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), '--extras-child', '--steps', '3', '--warmup', '2', '--no-cpu-baseline',
               '--clips-per-gpu', str(args.clips_per_gpu), '--read-precision', requested_precision]
        for flag, on in (('--no-miopen-find', args.no_miopen_find), ('--no-fuse-epilogue', args.no_fuse_epilogue), ('--fold-bn', args.fold_bn),
                         ('--nchw', not args.channels_last)):
            if on:
                cmd.append(flag)
        _phase('extras: child process')
        try:
            env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
            lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"extras"')]
            if res.returncode == 0 and lines:
                extras = json.loads(lines[-1])['extras']
            else:
                extras = {'error': 'extras child rc=%s: %s' % (res.returncode, (res.stderr or '')[-400:])}
        except Exception as exc:                           # (timeout, spawn failure)
            extras = {'error': 'extras child: %r' % (exc,)}
        _phase('extras: child done')
    elif rank == 0 and world == 1 and args.extras_child:
        extras = {}
        # ---- who owns the timed region's GPU time: hand-written kernels (namespace rmnet) vs everything else (MIOpen / rocBLAS /
        #      torch element-wise), from torch.profiler's device-side kernel records of three steps of the same loop
        try:
            if os.environ.get('BENCH_NO_PROFILER') == '1':
                raise RuntimeError('skipped (BENCH_NO_PROFILER=1)')
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for i in range(3):
                    eager_step(i)
                torch.cuda.synchronize()
            own = other = conv = 0.0
            top = {}
            for evt in prof.key_averages():
                dt_us = float(getattr(evt, 'device_time_total', 0.0) or getattr(evt, 'cuda_time_total', 0.0))
                if dt_us <= 0.0:
                    continue
                name = evt.key
                if 'rmnet' in name:
                    own += dt_us
                else:
                    other += dt_us
                    if any(k in name.lower() for k in ('conv', 'winograd', 'gemm', 'igemm', 'miopen', 'cijk', 'sp3asm', 'naive_conv')):
                        conv += dt_us
                top[name] = top.get(name, 0.0) + dt_us
            tot = own + other
            if tot > 0:
                big = sorted(top.items(), key=lambda kv: -kv[1])[:3]
                extras['timed_region_gpu_time_share'] = {
                    'hand_written_rmnet_kernels': round(own / tot, 4), 'convolution_gemm_kernels': round(conv / tot, 4),
                    'other_torch_kernels': round((other - conv) / tot, 4), 'gpu_busy_ms_per_step': round(tot / 3e3, 3),
                    'largest_kernels': [{'name': k[:80], 'share': round(v / tot, 4)} for k, v in big],
                    'note': 'torch.profiler device records of 3 steps (%d clips per GPU); box-to-box variance of the headline value '
                            'follows the convolution share (MIOpen solver choice), not the hand-written kernels' % B}
        except Exception as exc:                          # (profiler support depends on the torch / ROCm build)
            extras['timed_region_gpu_time_share'] = {'error': repr(exc)[:200]}
        _phase('extras: single stream')
        # ---- one clip alone (single stream): same step, B = 1
        ctx1 = net._ClipContext(net, 1, K_CH, H, W, [K_CH - 1], dev)
        bank1 = net.new_bank(ctx1, T_MEM)
        f1, m1 = frames[:1], masks[:1]
        for t in range(1, T_MEM):
            net.frame_step(ctx1, bank1, f1[:, t - 1], m1[:, t - 1], f1[:, t], tfn._forward(f1[:, t], f1[:, t - 1]), commit=True)
        net._profile_events = None

        def step1(i):
            t = T_MEM + (i % (n_clip - T_MEM))
            out1 = net.frame_step(ctx1, bank1, f1[:, t - 1], m1[:, t - 1], f1[:, t], tfn._forward(f1[:, t], f1[:, t - 1]), commit=False)
            return out1[1] if isinstance(out1, tuple) else torch.softmax(out1, dim=1)
        for i in range(5):
            step1(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(20):
            step1(i)
        torch.cuda.synchronize()
        extras['single_stream_fps'] = round(20 / (time.perf_counter() - t1), 2)
        # ... and what a single-stream frame is made of: GPU-busy time (sum of the kernels' device time) against wall time per frame.
        # busy ~ wall: bound by the batch-1 kernels themselves (a HIP graph cannot help); busy << wall: launch gaps
        try:
            if os.environ.get('BENCH_NO_PROFILER') == '1':
                raise RuntimeError('skipped (BENCH_NO_PROFILER=1)')
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof1:
                for i in range(4):
                    step1(i)
                torch.cuda.synchronize()
            own1 = conv1 = tot1 = 0.0
            nk = 0
            for evt in prof1.key_averages():
                dt_us = float(getattr(evt, 'device_time_total', 0.0) or getattr(evt, 'cuda_time_total', 0.0))
                if dt_us <= 0.0:
                    continue
                tot1 += dt_us
                nk += int(getattr(evt, 'count', 0))
                if 'rmnet' in evt.key:
                    own1 += dt_us
                elif any(k in evt.key.lower() for k in ('conv', 'winograd', 'gemm', 'igemm', 'miopen', 'cijk', 'sp3asm', 'naive_conv')):
                    conv1 += dt_us
            wall_ms = 1e3 / extras['single_stream_fps']
            extras['single_stream'] = {
                'fps': extras['single_stream_fps'], 'wall_ms_per_frame': round(wall_ms, 3), 'gpu_busy_ms_per_frame': round(tot1 / 4e3, 3),
                'busy_over_wall': round(tot1 / 4e3 / wall_ms, 3), 'kernel_launches_per_frame': round(nk / 4.0, 1),
                'convolution_gemm_share_of_busy': round(conv1 / max(tot1, 1e-9), 4), 'hand_written_share_of_busy': round(own1 / max(tot1, 1e-9), 4),
                'note': 'the reference\'s operating point (core/inference.py:22-28: one clip at a time): torch.profiler device records of 4 frames; '
                        'busy_over_wall near 1 = the batch-1 kernels themselves fill the frame, not launch gaps'}
        except Exception as exc:
            extras['single_stream'] = {'error': repr(exc)[:200]}
        # ---- the same single stream with the step (TinyFlowNet + frame_step) captured ONCE as a HIP graph and replayed
        #      (SURVEY 8f-3): ~340 launches per frame become one graph launch + three input copies
        try:
            sb = [f1[:, T_MEM - 1].clone(), m1[:, T_MEM - 1].clone(), f1[:, T_MEM].clone()]

            def body1():
                o1 = net.frame_step(ctx1, bank1, sb[0], sb[1], sb[2], tfn._forward(sb[2], sb[0]), commit=False)
                return o1[1] if isinstance(o1, tuple) else torch.softmax(o1, dim=1)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body1()
            torch.cuda.current_stream(dev).wait_stream(side)
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                o_static = body1()

            def gstep(i):
                t = T_MEM + (i % (n_clip - T_MEM))
                sb[0].copy_(f1[:, t - 1]); sb[1].copy_(m1[:, t - 1]); sb[2].copy_(f1[:, t])
                g1.replay()
                return o_static
            for i in range(5):
                gstep(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(40):
                gstep(i)
            torch.cuda.synchronize()
            extras['single_stream_graph_fps'] = round(40 / (time.perf_counter() - t1), 2)
            assert bool(torch.isfinite(o_static).all())
            del g1
        except Exception as exc:                      # (capture support depends on the torch / ROCm build)
            extras['single_stream_graph_fps'] = None
            extras['single_stream_graph_error'] = repr(exc)[:300]
        # ---- free-running loop: RMNet.forward on one 67-frame clip (DAVIS-val mean length), memorize_every = 5,
        #      estimated masks fed back (the real feedback edge), TinyFlowNet inside the timed region
        _phase('extras: free-running clip')
        from rmnet_amd.synthetic import synthetic_clip as _clip
        N_FREE = 67
        ff, fm, _, fn_obj = _clip(N_FREE, K_CH, H, W, seed=7, size=2.1)
        ff = ff.to(dev)
        net(ff[:, :6], fm[:, :6], tfn(ff[:, :6]), fn_obj[:, :6], 5)          # warm-up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        est = net(ff, fm, tfn(ff), fn_obj, 5, graph=False)
        torch.cuda.synchronize()
        dt_free = time.perf_counter() - t1
        t1 = time.perf_counter()
        est_g = net(ff, fm, tfn(ff), fn_obj, 5, graph=True)                  # opt-in graph replay (RMNet.forward's default is eager)
        torch.cuda.synchronize()
        dt_free_g = time.perf_counter() - t1
        saved_prec = net.read_precision
        net.read_precision = 'f16' if args.read_precision == 'split' else 'split'     # (the timed mode against split, or split against f16)
        net(ff[:, :6], fm[:, :6], tfn(ff[:, :6]), fn_obj[:, :6], 5)          # warm-up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        est_o = net(ff, fm, tfn(ff), fn_obj, 5, graph=False)
        torch.cuda.synchronize()
        dt_free_o = time.perf_counter() - t1
        net.read_precision = saved_prec
        la_, lb_ = est.argmax(2), est_o.argmax(2)
        inter = float(((la_ == 1) & (lb_ == 1)).sum())
        union = float(((la_ == 1) | (lb_ == 1)).sum())
        cover = float((est[0, 1:, 1] > 0.5).float().mean())
        extras['free_running'] = {'fps': round((N_FREE - 1) / dt_free, 2), 'fps_graph': round((N_FREE - 1) / dt_free_g, 2),
                                  'graph_vs_eager_max_prob_diff': round(float((est - est_g).abs().max()), 6),
                                  'f16_vs_split': {'fps_other_mode': round((N_FREE - 1) / dt_free_o, 2),
                                                   'label_iou': round(inter / union, 6) if union else 1.0,
                                                   'max_prob_diff': round(float((est - est_o).abs().max()), 6),
                                                   'note': 'the same clip through both arithmetic modes of the bank read (masks fed back for 66 frames)'},
                                  'frames': N_FREE, 'memorize_every': 5,
                                  'memory_frames_at_end': 14, 'clips': 1,
                                  'note': 'RMNet.forward + TinyFlowNet on one clip, masks fed back; with random-init weights the '
                                          'estimated object covers %.0f %% of the frame on average (boxes follow it)' % (100 * cover)}
        del ff, est, est_g, est_o
        _phase('extras: loops at the other configurations')
        # ---- whole-loop frames/s at the other BASELINE configurations: same step as the headline, other shapes
        # (MIOpen immediate mode here: a find pass for each of these shape sets costs ~90 s of wall time per configuration
        #  and the default run has to stay within minutes; measured with find: 60.2 / 65.5 / 44.0 / 250.7 frames/s)
        torch.backends.cudnn.benchmark = False
        extras['loops'] = {
            'cfg2_5obj_T5_480p_1clip': loop_fps(net, tfn, dev, 1, 6, 5, H, W, 5, 10),
            'cfg2_5obj_T5_480p_4clips': loop_fps(net, tfn, dev, 4, 6, 5, H, W, 5, 6),
            'cfg4_3obj_T20_720p_1clip': loop_fps(net, tfn, dev, 1, 4, 3, 720, 1280, 20, 6),
            'loader_K11_1obj_T5_480p_8clips': loop_fps(net, tfn, dev, 8, 11, 1, H, W, 5, 10),
            'note': 'memory pinned at T (T - 1 committed frames + the tentative previous frame); prev-frame masks = the synthetic '
                    'blobs; K = 11 mirrors the reference test loader (config.py:137): 10 of the 11 channels are empty for a 1-object clip.  '
                    'MIOpen immediate mode for these four (no find pass: it costs ~90 s of wall time per shape set); with find '
                    'the same loops measured 60.2 / 65.5 / 44.0 / 250.7 frames/s (profiles/r03_a_bench_line.json history)',
            'miopen_find': False}
        torch.backends.cudnn.benchmark = not args.no_miopen_find
        _phase('extras: kernels one by one')
        extras['kernels'] = kernel_figures(dev, events)
        _phase('extras done')

    if args.extras_child:
        print(json.dumps({'extras': extras}), flush=True)
        return
    if rank == 0:
        line = {
            'metric': 'frames/sec at 480p, 1 object, T=5 memory; memory-read HBM GB/s vs peak',
            'value': round(world * B * args.steps / elapsed, 3), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'split': 'f32 (convs fp32; memory read = split-fp16 MFMA hi*hi+hi*lo+lo*hi with fp32 accumulate, fp32-class accuracy)',
                      'qx': 'f32 convs; memory read = fp16 operands with an exact query (q as a hi/lo pair: two MFMA terms for the logits; K, P, V rounded '
                            'to 11 bits), fp32 accumulate (RMNET_BANK_QX) -- on one-object clips with live mask boundaries: mask IoU vs the CPU path '
                            '0.99999-1.00000, logits within 1e-4 = the exact-fp32 GPU loop\'s own distance (profiles/r05_iou_calibration.md)',
                      'f16': 'f32 convs; memory read = fp16 operands (K, V, q, P rounded to 11 bits), fp32 accumulate (RMNET_BANK_F16): the frame '
                             'loop\'s default for clips with one object -- on one-object 480x854 clips whose masks HAVE a boundary (tests/live_fixture.py) '
                             'mask IoU vs the CPU path 0.99993-0.99997 and foreground logits within 1.3e-3 (an IoU loss of 1e-3 ~ 2e-2); the same comparison '
                             'FAILS (0.9969) when the read-out is noised by 1 % (profiles/r05_iou_calibration.md, tests/test_gpu_parity.py).  [r6] \'auto\' keeps '
                             'it only while the largest affinity logit the bank has MEASURED on the clip stays below ' + ('%.0f (this run: %.1f)' % (AUTO_BOUND, logit_max)) +
                             ': with the key convolutions scaled until the soft-max is peaked (top-1 mass 0.5, logits ~150) f16 and qx fall to 0.9985-0.9990 and '
                             'the clip is re-read in split (profiles/r06_iou_temperature.md)'}[args.read_precision],
            'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: 480x854 synthetic clips, 1 object each (K=2), memory pinned '
                                   'at T=5, TinyFlowNet + memorize + regional read + decoder per frame; '
                                   'clips_per_gpu independent clips batched per GPU',
                       'weights': 'procedural random-init (no checkpoint offline)',
                       'prev_mask': 'synthetic blob mask of frame t-1 (object ~18 % of the frame, boxes ~46 % of the cells); '
                                    'extras.free_running feeds the estimated masks back instead',
                       'sharding': 'one clip per rank',
                       'miopen_find': not args.no_miopen_find, 'channels_last': bool(args.channels_last),
                       'hip_graph': bool(args.graph), 'clips_per_gpu': B,
                       'batchnorm_folded': bool(args.fold_bn),
                       'fused_epilogues': bool(not args.fold_bn and not args.no_fuse_epilogue)},
            'roofline': {'bound': 'hbm',        # the roof achieved / peak / frac are quoted against (BASELINE.json's metric: HBM GB/s vs peak)
                         'limited_by': 'not HBM: matrix pipe, L1 / TA and LDS of a CU at ~50 % each inside the tile walk (DESIGN.md section 4; '
                                       'the mfma object below prices the same launch against the fp16 MFMA peak)',
                         'kernel': 'bk_main<%d> = the whole regional memory read in ONE launch (%s MFMA read of the bank, merge of the '
                                   'partial results by the last workgroup of every query tile, masked cells, q_val half of the cat)'
                                   % (n_terms, {3: 'split-fp16 (3-term)', 2: 'fp16-operand, exact-query (2 + 1 terms)', 1: 'fp16-operand (1-term)'}[n_terms]),
                         'read_precision': args.read_precision, 'read_precision_requested': requested_precision,
                         'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'definition': 'SURVEY.md 8d / BASELINE.json: algorithmic bytes per launch (%d B = %d object-frames x 31,518,720 B) / '
                                       'mean kernel duration, against 8 TB/s HBM' % (abytes, B * (K_CH - 1)),
                         'traffic': traffic, 'traffic_source': traffic_note,
                         'algorithmic_bytes_per_launch': abytes,
                         'mfma': {'executed_%dterm_tflops' % n_terms: round(mfma_tflops, 1), 'useful_1term_tflops': round(useful_tflops, 1),
                                  'dense_1term_tflops': round(B * (K_CH - 1) * 2.0 * T_MEM * ctx.h * ctx.w * ctx.h * ctx.w * (DE + DO) / (main_avg * 1e-3) / 1e12, 1),
                                  'peak': 2500.0, 'unit': 'TFLOP/s', 'executed_frac_of_peak': round(mfma_tflops / 2500.0, 4),
                                  'note': 'executed = the MFMA terms of the mode (split: hi*hi + hi*lo + lo*hi; f16: one) over the compacted 64-query x 32-cell tiles incl. '
                                          'padding; useful = one term over the un-padded regional cells; dense = SURVEY 8d '
                                          '2*THW*hw*(De+Do) per object-frame as if nothing were masked.  The kernel is power-capped on the '
                                          'matrix pipe (profiles/r03_power_ceiling.md): a pure random-data MFMA loop sustains 1.7-1.9 PFLOP/s.  SQ counters of the same kernel '
                                          '(profiles/r03_d_mfma_pmc.md): SQ_VALU_MFMA_BUSY_CYCLES = 87,336 cycles per SIMD per launch of the default mode = 47-50 % of '
                                          'the un-profiled launch (68 % during the tile walks), 26-27 % for the fp16-operand mode'},
                         'launches': args.steps,
                         'auto_rule': {'requested': requested_precision, 'ran': args.read_precision, 'largest_logit_measured_by_the_bank': round(logit_max, 2),
                                       'bound': AUTO_BOUND, 'several_objects': 'split',
                                       'note': "read_precision='auto': one object per clip -> f16 while the bank's logit word (largest soft-max reference of "
                                               "its reads, natural units) stays <= bound, else the clip is re-read in split; several objects -> split "
                                               "(profiles/r06_iou_temperature.md, rmnet_amd/rmnet.py)"},
                         'avg_us': round(main_avg * 1e3, 2), 'avg_us_event_bracket': round(main_raw * 1e3, 2),
                         'event_floor_us': round(ev_floor_us, 2), 'min_us_event_bracket': round(min(main_ms) * 1e3, 2),
                         'op_avg_us': round((main_avg + comb_avg) * 1e3, 2), 'op_frac': round(op_achieved / HBM_PEAK_GBS, 4),
                         'timing': 'hipEventRecord on the launch stream around every bk_main of the timed region; avg_us = bracket mean '
                                   'minus the empty-bracket floor measured the same way (op_* adds the now empty second bracket where '
                                   'round 2 had its combine kernel)'},
        }
        if other_ms is not None:
            modes = {args.read_precision: {'avg_us': round((main_avg + comb_avg) * 1e3, 2), 'GBps': round(op_achieved, 1),
                                           'frac': round(op_achieved / HBM_PEAK_GBS, 4)}}
            for om, ms in other_ms.items():
                o_us = 1e3 * sum(ms) / len(ms)
                modes[om] = {'avg_us': round(o_us, 2), 'GBps': round(abytes / o_us / 1e3, 1), 'frac': round(abytes / o_us / 1e3 / HBM_PEAK_GBS, 4)}
            modes['note'] = ("the same launches (same bank, same boxes) in the three arithmetic modes of the kernel, whole read, HIP events on the "
                             "launch stream; the top-level figures are the timed region's mode ('%s').  split = fp32-class (error 1e-7); f16 = RMNET_BANK_F16: "
                             "fp16 operands, fp32 accumulate, the loop's default for one object per clip with small logits (this workload); qx = RMNET_BANK_QX: f16 with an exact "
                             "query (opt-in); clips with several objects and clips with peaked soft-maxes are read in split.  Whole-clip mask IoU vs the CPU path (profiles/r05_iou_calibration.md): "
                             "one object, live mask boundaries: f16 0.99993-0.99997, qx / split / exact fp32 0.99999-1.00000; 3 / 5 objects: exact fp32 "
                             ">= 0.9997, qx >= 0.9993, f16 0.9986-0.9995" % args.read_precision)
            line['roofline']['modes'] = modes
            if small_ms:
                sizes = {str(B): {'object_frames': B, 'avg_us': round((main_avg + comb_avg) * 1e3, 2), 'frac': round(op_achieved / HBM_PEAK_GBS, 4),
                                  'timed_region': True}}
                for n, ms in small_ms.items():
                    if isinstance(ms, str):
                        sizes[str(n)] = {'object_frames': n, 'error': ms}
                        continue
                    n_us = 1e3 * sum(ms) / len(ms)
                    abn = algorithmic_bytes(n * (K_CH - 1), T_MEM, ctx.h, ctx.w)
                    sizes[str(n)] = {'object_frames': n, 'avg_us': round(n_us, 2), 'GBps': round(abn / n_us / 1e3, 1),
                                     'frac': round(abn / n_us / 1e3 / HBM_PEAK_GBS, 4)}
                sizes = {k: sizes[k] for k in sorted(sizes, key=int)}
                sizes['note'] = ('the timed arithmetic (%s) at 1 ... 32 object-frames per launch (= clips per GPU at one object per clip), each inside the '
                                 'frame loop on its own clips (%d launches; MIOpen immediate mode).  13 (object, query tile) pairs per object-frame on ~210 '
                                 'computing workgroups: <= 12 object-frames: pairs cut into column blocks, merged by their last arriver; 16: one workgroup '
                                 'per pair, no merge; >= 20: ROUNDS of aligned chunks -- whole objects first, the rest cut to fill one more round '
                                 '(DESIGN.md section 4).  1 = the reference\'s operating point: one clip at a time' % (args.read_precision, min(args.steps, 10)))
                line['roofline']['launch_sizes'] = sizes
        if extras is not None and extras.get('single_stream_fps'):
            line['config']['workload'] += ' -- value = %d clips batched per GPU; ONE 480p stream alone: %.1f frames/s' % (B, extras['single_stream_fps'])
        else:
            line['config']['workload'] += ' -- value = %d clips batched per GPU' % B
        if multi is not None:
            line['multi_gpu'] = multi
        if extras is not None:
            line['extras'] = extras
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
            _phase('cpu baseline done')
        print(json.dumps(line), flush=True)
    if grouped:
        tdist.destroy_process_group()


if __name__ == '__main__':
    main()
