# -*- coding: utf-8 -*-
"""Generate tests/golden/*.npz by running the REFERENCE (hzxie/RMNet at /root/reference).

Runs only in the build container (the GPU box has no /root/reference).  Nothing from the
reference is copied: the script imports its Python modules in place, feeds them seeded inputs
and stores inputs + outputs as data.  Two imports the reference needs do not exist here and
are pre-seeded in ``sys.modules``:

* ``torchvision.models.resnet50``  -> ``rmnet_amd.networks.resnet50`` (torch-only trunk with
  torchvision's parameter names; weights are procedural either way -- no network access);
* the compiled CUDA module ``reg_att_map_generator`` (cannot be built: nvcc absent)
  -> ``oracle.region_map`` on CPU.  Consequence: fixtures that pass through the region map pin
  the *rest* of the path against the reference; the region map itself is pinned by the
  hand-computed known-answer file ``region_map_kat.json`` (written by hand, not by this script).

``region_fuzz`` does the same for the kernel's loosen / clamp / point-count branches (.cu:55-77): the reference's box finder
gives the tight boxes of 40 seeded masks whose box edges sit at, just before and just after every switching distance; the
tests apply the four clamp expressions themselves (cases.boxes_after_loosen) for every (n_bbox_loose_pixels,
n_pts_threshold) of cases.REGION_FUZZ_LOOSE x REGION_FUZZ_NPTS, independently of oracle/rmnet_oracle.c.

``flow_affine`` vectors come from the reference's own C++ compiled by oracle/Makefile
(oracle/_ref/flow_affine_transformation.so).

``region_boxes`` additionally pins the region map's box finder against the reference's own Python
box finder (utils/helpers.py:93-102 get_bounding_boxes): with n_pts_threshold = 1 and
n_bbox_loose_pixels = 0 the CUDA kernel reduces to it (reg_att_map_generator.cu:63-74 with L = 0).

    python tests/golden/make_golden.py                   # rewrites every fixture next to this script
    python tests/golden/make_golden.py region_boxes msi  # only the named sections
"""

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from rmnet_amd import networks  # noqa: E402
from rmnet_amd.synthetic import synthetic_clip  # noqa: E402


def _install_stand_ins():
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvm.resnet50 = networks.resnet50
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm

    ext = types.ModuleType('reg_att_map_generator')

    def forward(mask, prob_threshold, n_pts_threshold, n_bbox_loose_pixels):
        att, bb = oracle.region_map(mask.detach().cpu().numpy(), prob_threshold, n_pts_threshold,
                                    n_bbox_loose_pixels)
        return torch.from_numpy(att), torch.from_numpy(bb)

    ext.forward = forward
    sys.modules['reg_att_map_generator'] = ext
    sys.path.insert(0, REF)


def _load_ref_flow_ext():
    so = os.path.join(ROOT, 'oracle', '_ref', 'flow_affine_transformation.so')
    spec = importlib.util.spec_from_file_location('flow_affine_transformation', so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def section_memory_reader(ref_rmnet, ref_tfn, ref_helpers, rng):
    # ---------------------------------------------------------------- MemoryReader (M1)
    reader = ref_rmnet.MemoryReader()
    cases = {}

    def mr_case(name, no, De, Do, T, h, w, zero_mem=None, zero_qry=None, keep_p=False, scale=1.0):
        mk = (rng.randn(no, De, T, h, w) * scale).astype(np.float32)
        mv = rng.randn(no, Do, T, h, w).astype(np.float32)
        qk = (rng.randn(no, De, h, w) * scale).astype(np.float32)
        qv = rng.randn(no, Do, h, w).astype(np.float32)
        if zero_mem is not None:
            for o in range(no):
                for t in range(T):
                    cx0, cx1, cy0, cy1 = zero_mem[o][t]
                    keep = np.zeros((h, w), np.float32)
                    keep[cy0:cy1 + 1, cx0:cx1 + 1] = 1
                    mk[o, :, t] *= keep
                    mv[o, :, t] *= keep
        if zero_qry is not None:
            for o in range(no):
                cx0, cx1, cy0, cy1 = zero_qry[o]
                keep = np.zeros((h, w), np.float32)
                keep[cy0:cy1 + 1, cx0:cx1 + 1] = 1
                qk[o] *= keep
                qv[o] *= keep
        out, p = reader(torch.from_numpy(mk), torch.from_numpy(mv), torch.from_numpy(qk),
                        torch.from_numpy(qv))
        cases[name + '.m_key'], cases[name + '.m_val'] = mk, mv
        cases[name + '.q_key'], cases[name + '.q_val'] = qk, qv
        cases[name + '.mem_val'] = out.numpy()
        if zero_mem is not None:
            cases[name + '.mem_rects'] = np.asarray(zero_mem, np.int32)
        if zero_qry is not None:
            cases[name + '.qry_rects'] = np.asarray(zero_qry, np.int32)
        if keep_p:
            cases[name + '.p'] = p.numpy()

    mr_case('dense', 1, 128, 512, 2, 6, 8, scale=0.5)
    mr_case('regional', 2, 128, 512, 2, 6, 8, scale=0.5,
            zero_mem=[[(1, 5, 0, 3), (2, 7, 2, 5)], [(0, 7, 0, 5), (3, 3, 1, 4)]],
            zero_qry=[(2, 6, 1, 4), (0, 4, 0, 5)])
    mr_case('allmasked_query', 1, 128, 512, 3, 5, 7, scale=0.5,
            zero_mem=[[(0, 6, 0, 4), (1, 4, 1, 3), (1, 0, 1, 0)]], zero_qry=[(1, 0, 1, 0)])
    mr_case('tiny_p', 1, 16, 32, 1, 3, 4, keep_p=True)
    mr_case('peaky', 1, 128, 512, 2, 4, 5, scale=3.0)      # large logits: exercises the running max
    if 'memory_reader' in WRITE:
        np.savez_compressed(os.path.join(HERE, 'memory_reader.npz'), **cases)



def section_flow_affine(ref_rmnet, ref_tfn, ref_helpers, rng):
    # ---------------------------------------------------------------- flow affine (F1)
    ext = _load_ref_flow_ext()
    fa = {}

    def fa_case(name, H, W, m1, m2, amp):
        flow = (rng.rand(H, W, 2).astype(np.float32) - 0.5) * amp
        m1 = np.asarray(m1, np.float32).reshape(2, 3)
        m2 = np.asarray(m2, np.float32).reshape(2, 3)
        fa[name + '.flow'], fa[name + '.m1'], fa[name + '.m2'] = flow, m1, m2
        fa[name + '.out'] = ext.update_optical_flow(flow, m1, m2)

    ident = [1, 0, 0, 0, 1, 0]
    th = 0.3
    rot = [1.1 * np.cos(th), -1.1 * np.sin(th), 3.5, 1.1 * np.sin(th), 1.1 * np.cos(th), -2.25]
    fa_case('identity', 48, 64, ident, ident, 6.0)
    fa_case('rand01', 48, 64, rng.rand(6), rng.rand(6), 1.0)        # what the reference's test.py feeds
    fa_case('rot_scale', 48, 64, rot, [0.9, 0.05, -1.0, -0.05, 0.9, 2.0], 10.0)
    fa_case('out_of_range', 37, 53, rot, ident, 400.0)              # clamps on every side
    fa_case('half_ties', 16, 24, [1, 0, 0.5, 0, 1, -0.5], [1, 0, 1.5, 0, 1, 2.5], 0.0)  # round half away
    np.savez_compressed(os.path.join(HERE, 'flow_affine.npz'), **fa)



def section_pad(ref_rmnet, ref_tfn, ref_helpers, rng):
    # ---------------------------------------------------------------- pad_divide_by (H1)
    pads = {}
    for (h, w, d) in [(480, 854, 16), (480, 910, 16), (720, 1280, 16), (150, 250, 16), (480, 854, 64),
                      (33, 47, 16)]:
        (x,), pad = ref_helpers.pad_divide_by([torch.zeros(1, 1, h, w)], d, (h, w))
        pads['%dx%d/%d' % (h, w, d)] = {'pad': [int(v) for v in pad], 'shape': list(x.shape[2:])}
    with open(os.path.join(HERE, 'pad_divide_by.json'), 'w') as f:
        json.dump(pads, f, indent=1, sort_keys=True)



def section_clip(ref_rmnet, ref_tfn, ref_helpers, rng):
    # ---------------------------------------------------------------- RMNet glue (P1-P5, A1)
    net = ref_rmnet.RMNet(None)
    networks.procedural_init_(net)
    net.eval()
    N, K, H, W = 4, 3, 150, 250
    frames, masks, flows, n_objects = synthetic_clip(N, K, H, W, seed=3)
    g = {}
    g['state_keys'] = np.array(sorted(net.state_dict().keys()))
    g['weights_checksum'] = np.float64(sum(float(v.double().abs().sum()) for v in net.state_dict().values()))
    # A1: warp + region map
    soft = torch.rand(1, K, H, W, generator=torch.Generator().manual_seed(5))
    warped, valid = net.warp(soft, flows[:, 1] * 2.5)
    g['warp.in'], g['warp.flow'] = soft.numpy(), (flows[:, 1] * 2.5).numpy()
    g['warp.out'], g['warp.valid'] = warped.numpy(), valid.numpy()
    # P1: memorize frame 0
    k4, v4, bb = net.memorize(frames[:, 0], masks[:, 0].float(), [K - 1])
    g['memorize.k4'], g['memorize.bboxes'] = k4.numpy(), bb.numpy()
    g['memorize.v4_every4'] = v4[:, :, ::4].contiguous().numpy()   # keeps the fixture small
    # P4: soft aggregation
    ps = torch.rand(K - 1, 20, 30, generator=torch.Generator().manual_seed(6))
    g['softagg.ps'], g['softagg.logit'] = ps.numpy(), net.soft_aggregation(ps, K, [K - 1]).numpy()
    # P5: whole clip, memorize_every = 2 so that both the committed and the tentative slot occur
    est = net(frames, masks, flows, n_objects, 2)
    g['clip.N'], g['clip.K'], g['clip.H'], g['clip.W'], g['clip.seed'] = N, K, H, W, 3
    g['clip.memorize_every'] = 2
    g['clip.est_argmax'] = est.argmax(dim=2).numpy().astype(np.uint8)
    g['clip.est_last_obj1'] = est[:, -1, 1].numpy()
    g['clip.est_t1'] = est[:, 1].numpy()
    np.savez_compressed(os.path.join(HERE, 'rmnet_clip.npz'), **g)



def section_tiny_flownet(ref_rmnet, ref_tfn, ref_helpers, rng):
    # ---------------------------------------------------------------- TinyFlowNet (C2)
    tfn = ref_tfn.TinyFlowNet(None)
    networks.procedural_init_(tfn)
    tfn.eval()
    frames = synthetic_clip(4, 3, 150, 250, seed=3)[0]
    small = frames[:, :2, :, :70, :100].contiguous()
    fl = tfn(small)
    np.savez_compressed(os.path.join(HERE, 'tiny_flownet.npz'), frames=small.numpy(), flows=fl.numpy(),
                        state_keys=np.array(sorted(tfn.state_dict().keys())))


def section_region_boxes(ref_rmnet, ref_tfn, ref_helpers, rng):
    # ---------------------------------------------------------------- region map boxes (G1)
    # The reference's own box finder on thresholded soft masks (tests/golden/cases.py builds the masks);
    # -1 rows = it returned None (empty channel).
    import cases
    out = {}
    for i, (B, K, H, W) in enumerate(cases.REGION_BOX_SHAPES):
        m = cases.region_box_case(i)
        tight = np.full((B * K, 4), -1, np.int32)
        for b in range(B):
            for k in range(K):
                bb = ref_helpers.get_bounding_boxes(m[b, k] >= 0.5)
                if bb[0] is not None:
                    tight[b * K + k] = [int(v) for v in bb]
        out['case%02d.tight' % i] = tight
        out['case%02d.checksum' % i] = np.float64(m.astype(np.float64).sum())   # guards the shared generator
    np.savez_compressed(os.path.join(HERE, 'region_boxes.npz'), **out)


def section_region_fuzz(ref_rmnet, ref_tfn, ref_helpers, rng):
    # ---------------------------------------------------------------- region map loosen / clamp / count branches (G1)
    # Reference side of the fuzz (tests/golden/cases.py: region_fuzz_case): the REFERENCE's own box finder on the
    # thresholded masks + the number of above-threshold pixels (what .cu:41-42 counts); the clamp expressions of
    # .cu:55-77 are evaluated by the tests on these (cases.boxes_after_loosen) -- rmnet_oracle.c is not involved.
    import cases
    out = {}
    for i, (B, K, H, W) in enumerate(cases.REGION_FUZZ_SHAPES):
        m = cases.region_fuzz_case(i)
        tight = np.full((B * K, 4), -1, np.int32)
        npts = np.zeros(B * K, np.int32)
        for b in range(B):
            for k in range(K):
                hit = m[b, k] >= np.float32(0.5)
                npts[b * K + k] = int(hit.sum())
                bb = ref_helpers.get_bounding_boxes(hit)
                if bb[0] is not None:
                    tight[b * K + k] = [int(v) for v in bb]
        out['case%02d.tight' % i] = tight
        out['case%02d.npts' % i] = npts
        out['case%02d.checksum' % i] = np.float64(m.astype(np.float64).sum())
    np.savez_compressed(os.path.join(HERE, 'region_fuzz.npz'), **out)


def section_msi(ref_rmnet, ref_tfn, ref_helpers, rng):
    # ---------------------------------------------------------------- multi_scale_inference (H1)
    import cases
    from types import SimpleNamespace
    net = ref_rmnet.RMNet(None)
    networks.procedural_init_(net)
    net.eval()
    tfn = ref_tfn.TinyFlowNet(None)
    networks.procedural_init_(tfn)
    tfn.eval()
    c = cases.MSI_CLIP
    frames, masks, _, n_objects = synthetic_clip(c['N'], c['K'], c['H'], c['W'], seed=c['seed'], size=c['size'])
    out = {}
    for name, scales, flip in cases.MSI_CASES:
        cfg = SimpleNamespace(TEST=SimpleNamespace(FRAME_SCALES=scales, FLIP_LR=flip,
                                                   MEMORIZE_EVERY=c['memorize_every']))
        flows, probs = ref_helpers.multi_scale_inference(cfg, tfn, net, frames, masks, n_objects)
        out[name + '.flows'] = flows.numpy().astype(np.float16)            # (compact: compared at 2e-3)
        out[name + '.probs'] = probs.numpy().astype(np.float16)
        out[name + '.argmax'] = probs.argmax(dim=2).numpy().astype(np.uint8)
    # var_or_cuda on the CPU-only container: contiguous copy, device unchanged (utils/helpers.py:16-24)
    x = torch.arange(24.).view(2, 3, 4).transpose(1, 2)
    y = ref_helpers.var_or_cuda(x)
    out['var_or_cuda.contiguous'] = np.array([bool(y.is_contiguous()), bool(torch.equal(x, y))])
    np.savez_compressed(os.path.join(HERE, 'multi_scale_inference.npz'), **out)


WRITE = set()      # sections whose files are rewritten by this run

SECTIONS = [('memory_reader', section_memory_reader), ('flow_affine', section_flow_affine), ('pad', section_pad),
            ('clip', section_clip), ('tiny_flownet', section_tiny_flownet), ('region_boxes', section_region_boxes),
            ('region_fuzz', section_region_fuzz), ('msi', section_msi)]


def main(argv):
    _install_stand_ins()
    sys.path.insert(0, HERE)
    import models.rmnet as ref_rmnet          # the reference, imported in place
    import models.tiny_flownet as ref_tfn
    import utils.helpers as ref_helpers
    torch.set_grad_enabled(False)
    want = set(argv) or {n for n, _ in SECTIONS}
    unknown = want - {n for n, _ in SECTIONS}
    assert not unknown, 'unknown sections: %s' % sorted(unknown)
    WRITE.update(want)
    if 'flow_affine' in want:
        want.add('memory_reader')     # the two share one random stream (memory_reader draws first); replayed, not rewritten
    shared = np.random.RandomState(1234)
    for name, fn in SECTIONS:
        if name in want:
            torch.manual_seed(0)
            fn(ref_rmnet, ref_tfn, ref_helpers, shared if name in ('memory_reader', 'flow_affine') else np.random.RandomState(99))
            print('section', name, 'done' if name in WRITE else 'replayed (not written)')
    print('golden fixtures written to', HERE)
    for fn in sorted(os.listdir(HERE)):
        print('  %-28s %8d B' % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == '__main__':
    main(sys.argv[1:])
