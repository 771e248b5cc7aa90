# -*- coding: utf-8 -*-
"""What 16-bit convolutions would give (round-5 verdict, "next" item 4: "a bf16/fp16 conv run goes in extras only, with its
live-fixture IoU beside it").  NOT the driver line: bench.py's convolutions stay fp32, as the reference's.  MEASUREMENT TOOL.

The frame loop of bench.py (BASELINE configs[1]: 480x854, 1 object, T = 5 pinned, 16 clips per GPU, channels_last) with both
convolution stacks under ``torch.autocast`` (MIOpen's fp16 / bf16 MFMA kernels; the module graph, i.e. torch's own BatchNorm / ReLU:
the fused glue kernels of csrc/epilogue.hip are fp32), the hand-written read path unchanged (keys / values / read-out enter and leave it
as fp32).  Per dtype: frames/s of the loop, and the live-boundary fixture (tests/live_fixture.py) through ``RMNet.forward`` under the
same autocast against (a) the fp32 GPU loop and (b) the CPU path (oracle.OracleRMNet): label IoU, largest live foreground-logit gap.

    python tools/conv16_probe.py [--steps 10] [--clips 16] [--fixture live480-a] [--no-cpu]
"""
import argparse, contextlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('PYTORCH_MIOPEN_SUGGEST_NHWC', '1')
import torch
from rmnet_amd import networks
from rmnet_amd.rmnet import RMNet
from rmnet_amd.tiny_flownet import TinyFlowNet
from rmnet_amd.synthetic import synthetic_clip
import live_fixture as lf

torch.set_grad_enabled(False)
H, W, K_CH, T_MEM = 480, 854, 2, 5


def fp32_boundaries(net, tfn=None):
    """The read path's C ABI takes fp32: the KeyValue heads hand fp32 keys / values over whatever the stacks computed in, the flow
    leaves TinyFlowNet as fp32.  (Wrappers live HERE, not in the product: the product's stacks are fp32.)"""
    for kv in (net.kv_memory, net.kv_query):
        inner = kv.forward
        kv.forward = (lambda f: (lambda x: tuple(t.float() for t in f(x))))(inner)
    if tfn is not None:
        inner_t = tfn._forward
        tfn._forward = lambda a, b: inner_t(a, b).float()
    return net, tfn


def _to_f32(x):
    if torch.is_tensor(x):
        return x.float() if x.is_floating_point() else x
    if isinstance(x, (tuple, list)):
        return type(x)(_to_f32(t) for t in x)
    return x


PARTS = {'enc': ('encoder_memory', 'encoder_query'), 'kv': ('kv_memory', 'kv_query'), 'dec': ('decoder',)}


def autocast_parts(net, parts, dtype):
    """Only the named stacks of ``net`` run under autocast (their outputs leave as fp32): which stack pays how much of the IoU."""
    for part in parts:
        for name in PARTS[part]:
            mod = getattr(net, name)
            inner = mod.forward

            def wrapped(*a, _f=inner):
                with torch.autocast('cuda', dtype=dtype):
                    return _to_f32(_f(*a))
            mod.forward = wrapped
    return net


def cast_ctx(dtype):
    return contextlib.nullcontext() if dtype is None else torch.autocast('cuda', dtype=dtype)


def loop_fps(dtype, dev, clips, steps, warmup, parts=None):
    net = networks.procedural_init_(RMNet(None, read_precision='auto')).to(dev).eval()
    tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev).eval()
    net = net.to(memory_format=torch.channels_last)
    tfn = tfn.to(memory_format=torch.channels_last)
    fp32_boundaries(net, tfn)
    if parts:
        autocast_parts(net, parts, dtype)
        dtype = None
    n_clip = 12
    cl = [synthetic_clip(n_clip, K_CH, H, W, seed=c, size=2.1) for c in range(clips)]
    frames = torch.cat([c[0] for c in cl]).to(dev)
    masks = torch.cat([c[1] for c in cl]).to(dev).float()
    ctx = net._ClipContext(net, clips, K_CH, H, W, [K_CH - 1] * clips, dev)
    bank = net.new_bank(ctx, T_MEM)
    with cast_ctx(dtype):
        for t in range(1, T_MEM):
            flow = tfn._forward(frames[:, t], frames[:, t - 1])
            net.frame_step(ctx, bank, frames[:, t - 1], masks[:, t - 1], frames[:, t], flow, commit=True)

        def step(i):
            t = T_MEM + (i % (n_clip - T_MEM))
            flow = tfn._forward(frames[:, t], frames[:, t - 1])
            out = net.frame_step(ctx, bank, frames[:, t - 1], masks[:, t - 1], frames[:, t], flow, commit=False)
            return torch.softmax(out.float(), dim=1)
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    return clips * steps / dt, 1e3 * dt / steps, bank.logit_max()


def fixture_rows(name, dev, with_cpu, selective=()):
    frames, masks, flows, n_objects, every, delta = lf.make_clip(name)
    rows, ref_gpu = {}, None
    cpu = None
    if with_cpu:
        from oracle import oracle
        ref = networks.procedural_init_(oracle.OracleRMNet()).eval()
        lf.shift_foreground_bias(ref, delta)
        nt = torch.get_num_threads()
        torch.set_num_threads(min(16, nt))
        t0 = time.time()
        cpu = ref(frames, masks, flows, n_objects, every, return_logits=True)
        torch.set_num_threads(nt)
        rows['cpu_path_s'] = round(time.time() - t0, 1)
        lf.assert_live(cpu[0], name)
    cases = [('fp32', None, None)] + ([('fp16', torch.float16, None), ('bf16', torch.bfloat16, None)] if not selective else
                                     [('fp16:' + '+'.join(ps), torch.float16, ps) for ps in selective])
    for tag, dtype, parts in cases:
        net = networks.procedural_init_(RMNet(None, read_precision='auto')).to(dev).eval()
        lf.shift_foreground_bias(net, delta)
        net = net.to(memory_format=torch.channels_last)
        fp32_boundaries(net)
        if parts:
            autocast_parts(net, parts, dtype)
            dtype = None
        with cast_ctx(dtype):
            est, logits = net(frames, masks, flows, n_objects, every, return_logits=True)
        est, logits = est.float().cpu(), logits.float().cpu()
        if ref_gpu is None:
            ref_gpu = (est, logits)
        r = {'read_precision': net.last_clip['read_precision'], 'reread': net.last_clip['reread'],
             'iou_vs_fp32_gpu_loop': round(lf.label_iou(est, ref_gpu[0]), 5), 'live_fg_logit_gap_vs_fp32_gpu_loop': float('%.3g' % lf.logit_gap(logits, ref_gpu[1]))}
        if cpu is not None:
            r['iou_vs_cpu_path'] = round(lf.label_iou(est, cpu[0]), 5)
            r['live_fg_logit_gap_vs_cpu_path'] = float('%.3g' % lf.logit_gap(logits, cpu[1]))
        rows[tag] = r
        print(name, tag, json.dumps(r), flush=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--clips', type=int, default=16)
    ap.add_argument('--fixture', default='live480-a')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-loop', action='store_true', help='fixtures only')
    ap.add_argument('--no-fixture', action='store_true', help='loops only')
    ap.add_argument('--dtypes', default='fp32,fp16,bf16', help='loop dtypes of the all-stacks run')
    ap.add_argument('--selective', default='', help="fp16 for some stacks only, e.g. 'enc,dec,enc+kv' (enc = both ResNet-50 encoders, kv = the KeyValue heads, dec = the decoder)")
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    assert torch.cuda.is_available(), 'needs the GPU box'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    res = {'workload': '480x854, 1 object, T = 5 pinned, %d clips per GPU, channels_last, module graph (no fused glue), %d steps' % (args.clips, args.steps),
           'loop': {}, 'fixture': {}}
    torch.backends.cudnn.benchmark = True          # MIOpen find, as bench.py
    selective = [tuple(c.split('+')) for c in args.selective.split(',') if c]
    loops = [l for l in (('fp32', None, None), ('fp16', torch.float16, None), ('bf16', torch.bfloat16, None)) if l[0] in args.dtypes.split(',')] if not selective else \
        [('fp16:' + '+'.join(ps), torch.float16, ps) for ps in selective]
    for tag, dtype, parts in ([] if args.no_loop else loops):
        try:
            fps, ms, lmax = loop_fps(dtype, dev, args.clips, args.steps, args.warmup, parts)
            res['loop'][tag] = {'frames_per_s': round(fps, 1), 'ms_per_step': round(ms, 2), 'largest_logit': round(lmax, 2)}
        except Exception as exc:      # (a dtype MIOpen cannot serve on this build is a finding, not a crash)
            res['loop'][tag] = {'error': repr(exc)[:300]}
        print('loop', tag, json.dumps(res['loop'][tag]), flush=True)
        torch.cuda.empty_cache()
    torch.backends.cudnn.benchmark = False         # (batch-1 fixture shapes: no second find per dtype)
    for fx in ([] if args.no_fixture else args.fixture.split(',')):
        try:
            res['fixture'][fx] = fixture_rows(fx, dev, not args.no_cpu, selective)
        except Exception as exc:
            res['fixture'][fx] = {'error': repr(exc)[:300]}
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, 'w') as fh:
            fh.write(line + '\n')


if __name__ == '__main__':
    main()
