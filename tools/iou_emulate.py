# -*- coding: utf-8 -*-
"""CPU-only pricing of the bank read's arithmetic modes on whole clips (no GPU needed): the CPU path (oracle.OracleRMNet) with
MemoryReader.forward replaced by tests/live_fixture.rounded_reader -- the same function with K / q / P / V rounded to fp16 the way
csrc/bank.hip's modes round them -- against the unrounded CPU path.  Both sides run the same convolutions on the same cores, so the
IoU loss printed is the rounding's alone (the GPU-vs-CPU rows of tools/iou_calib.py add MIOpen-vs-CPU convolution rounding).
Also: mutation rows (read-out zeroed / noised) and the bias scan that chose the live fixtures' shifts.

    python tools/iou_emulate.py live <fixture> [modes]            e.g.  live live480-a f16,mixed,qx,K,q,P,V,zero,noise0.01,noise0.001
    python tools/iou_emulate.py multi <n_obj> <N> <every> <seed> <size> [modes]     (the round-4 calibration clips: several objects)
    python tools/iou_emulate.py scan <fixture> <delta,delta,...>  cover / near-threshold fraction per frame for candidate shifts
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle import oracle
from rmnet_amd import networks
from rmnet_amd.synthetic import synthetic_clip
import live_fixture as lf
torch.set_grad_enabled(False)
nt = int(os.environ.get('THREADS', '8'))
torch.set_num_threads(nt)


def net(reader='torch', delta=0.0):
    return lf.shift_foreground_bias(networks.procedural_init_(oracle.OracleRMNet(reader=reader)).eval(), delta)


def reader_of(mode):
    if mode == 'zero':
        return lf.rounded_reader('exact', mutate='zero')
    if mode.startswith('noise'):
        return lf.rounded_reader('exact', mutate=('noise', float(mode[5:])))
    return lf.rounded_reader(mode)


def rows(frames, masks, flows, n_objects, every, delta, modes, n_obj):
    t0 = time.time()
    ref, ref_l = net('torch', delta)(frames, masks, flows, n_objects, every, return_logits=True)
    print('  CPU path %.0f s at %d threads; cover / near-threshold per frame: %s' % (
        time.time() - t0, nt, ' '.join('%.3f/%.3f' % cn for cn in lf.liveness(ref)) if n_obj == 1 else
        ' '.join('%.3f' % float((ref[0, -1].argmax(0) == k).float().mean()) for k in range(1, n_obj + 1))), flush=True)
    for mode in modes:
        est, lg = net(reader_of(mode), delta)(frames, masks, flows, n_objects, every, return_logits=True)
        ious = [lf.label_iou(est, ref, k) for k in range(1, n_obj + 1)]
        print('  %-10s clip IoU %s  min %.5f | max prob diff %.2e | max fg-logit diff (unclamped) %.3e' % (
            mode, ' '.join('%.5f' % i for i in ious), min(ious), float((est - ref).abs().max()), lf.logit_gap(lg, ref_l)), flush=True)


what = sys.argv[1]
if what == 'live':
    name = sys.argv[2]
    modes = (sys.argv[3] if len(sys.argv) > 3 else 'f16,mixed,qx,zero,noise0.01,noise0.001').split(',')
    frames, masks, flows, n_objects, every, delta = lf.make_clip(name)
    print('%s: %s, bias shift %.2f' % (name, lf.LIVE_CLIPS[name], delta))
    rows(frames, masks, flows, n_objects, every, delta, modes, 1)
elif what == 'multi':
    n_obj, N, every, seed, size = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
    modes = (sys.argv[7] if len(sys.argv) > 7 else 'f16,mixed,qx').split(',')
    frames, masks, flows, n_objects = synthetic_clip(N, n_obj + 1, 480, 854, seed=seed, size=size)
    print('%d objects 480x854, %d frames, memorize_every %d, seed %d, blob size %.1f' % (n_obj, N, every, seed, size))
    rows(frames, masks, flows, n_objects, every, 0.0, modes, n_obj)
elif what == 'scan':
    name = sys.argv[2]
    frames, masks, flows, n_objects, every, _ = lf.make_clip(name)
    for d in [float(x) for x in sys.argv[3].split(',')]:
        est = net('torch', d)(frames, masks, flows, n_objects, every)
        print('%s delta %6.2f  cover/near per frame: %s' % (name, d, ' '.join('%.3f/%.3f' % cn for cn in lf.liveness(est))), flush=True)
