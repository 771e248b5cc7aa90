# -*- coding: utf-8 -*-
"""Host-side logic that needs no GPU: product code never touches the oracle, the state-dict keys
match the reference's, video sharding is balanced, and the N>1 gather works (gloo, world_size 2)."""

import os
import re
import socket

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'rmnet_amd')):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(base, fn), errors='ignore').read()
                if re.search(r'^\s*(from|import)\s+oracle|liboracle|oracle/_ref', text, flags=re.M):
                    bad.append(fn)
    assert not bad, bad


def test_state_dict_keys_match_reference(golden_dir):
    from rmnet_amd.rmnet import RMNet
    g = np.load(os.path.join(golden_dir, 'rmnet_clip.npz'))
    net = RMNet(None)
    assert sorted(net.state_dict().keys()) == list(g['state_keys'])
    # DataParallel-prefixed checkpoints load too (core/inference.py:43)
    sd = {'module.' + k: v for k, v in net.state_dict().items()}
    net.load_reference_state_dict(sd)
    from rmnet_amd.tiny_flownet import TinyFlowNet
    tfn = TinyFlowNet(None)
    tfn.load_reference_state_dict({'module.' + k: v for k, v in tfn.state_dict().items()})


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from rmnet_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.RMNetHipError, match='missing'):
        _lib.load()


def test_assign_videos_is_balanced_and_deterministic():
    from rmnet_amd.dist import assign_videos, my_videos
    costs = [67, 34, 104, 50, 80, 91, 40, 75, 69, 36, 99, 58, 84, 43, 77, 66]
    owner = assign_videos(costs, 8)
    assert owner == assign_videos(costs, 8)
    loads = [sum(c for c, r in zip(costs, owner) if r == k) for k in range(8)]
    assert max(loads) - min(loads) <= max(costs)
    assert sorted(v for r in range(8) for v in my_videos(costs, r, 8)) == list(range(len(costs)))
    assert assign_videos(costs, 1) == [0] * len(costs)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    from rmnet_amd import dist as rd
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    rd.init_from_env(backend='gloo')
    costs = [5, 3, 4, 2, 6]
    mine = rd.my_videos(costs, rank, world)
    res = {v: torch.full((costs[v], 4 + v, 6), 10 * v + 1, dtype=torch.uint8) for v in mine}
    rd.barrier()
    out = rd.gather_label_maps(res, len(costs))
    mx = rd.max_over_ranks(1.0 + rank)
    sm = rd.sum_over_ranks(2.0)
    if rank == 0:
        q.put((sorted(out.keys()), [tuple(out[v].shape) for v in sorted(out)],
               [int(out[v].float().mean()) for v in sorted(out)], mx, sm))
    else:
        assert out == {}
    dist.destroy_process_group()


def test_two_rank_gather_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    keys, shapes, means, mx, sm = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert keys == [0, 1, 2, 3, 4]
    assert shapes == [(5, 4, 6), (3, 5, 6), (4, 6, 6), (2, 7, 6), (6, 8, 6)]
    assert means == [1, 11, 21, 31, 41]
    assert mx == 2.0 and sm == 4.0


def test_device_side_jaccard_matches_the_reference_definition(oracle_mod):
    """rmnet_amd.metrics (utils/metrics.py:84-102 as tensor ops) vs the oracle's scalar restatement,
    incl. the both-empty -> 1 rule."""
    from rmnet_amd import metrics
    rng = np.random.RandomState(5)
    N, H, W, n = 4, 17, 23, 3
    pred = rng.randint(0, n + 1, size=(N, H, W))
    gt = rng.randint(0, n + 1, size=(N, H, W))
    pred[2][pred[2] == 2] = 0
    gt[2][gt[2] == 2] = 0                    # object 2 absent in both at frame 2
    j = metrics.jaccard_per_object(torch.from_numpy(pred), torch.from_numpy(gt), n).numpy()
    for t in range(N):
        for o in range(1, n + 1):
            assert abs(j[t, o - 1] - oracle_mod.iou(pred[t] == o, gt[t] == o)) < 1e-12
    assert j[2, 1] == 1.0
    assert abs(float(metrics.mean_jaccard(torch.from_numpy(pred), torch.from_numpy(gt), n)) - j[1:-1].mean()) < 1e-12


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rmnet_amd import inference
    g = torch.Generator().manual_seed(0)
    videos = []
    for v, (n, no) in enumerate([(5, 1), (9, 2), (3, 1), (7, 3), (4, 1)]):
        labels = torch.randint(0, no + 1, (n, 6, 8), generator=g, dtype=torch.uint8)
        videos.append({'frames': torch.zeros(n, 3, 6, 8), 'n_objects': no, 'labels': labels, 'id': v})
    seen = []

    def stub(video):                       # "prediction" = ground truth with object 1 erased on frame 1
        seen.append(video['id'])
        out = video['labels'].clone()
        out[1][out[1] == 1] = 0
        return out
    maps = inference.segment_videos(videos, stub)
    score = inference.evaluate_videos(videos, stub)
    q.put((rank, sorted(seen), {k: v.numpy().copy() for k, v in maps.items()}, score))   # (by value: the sender may exit before the parent reads)
    dist.destroy_process_group()


def test_sharded_video_inference_two_ranks_gloo():
    """rmnet_amd.inference over 2 gloo ranks: every clip runs on exactly one rank, rank 0 receives all
    label maps unchanged, and the all-reduced J equals the single-process value."""
    import torch.multiprocessing as mp
    from rmnet_amd import inference
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, seen0, maps0, j0), (r1, seen1, maps1, j1) = out
    assert sorted(set(seen0) | set(seen1)) == [0, 1, 2, 3, 4] and not (set(seen0) & set(seen1))
    assert sorted(maps0) == [0, 1, 2, 3, 4] and maps1 == {}
    g = torch.Generator().manual_seed(0)
    want_j_num = want_j_den = 0.0
    from rmnet_amd import metrics
    for v, (n, no) in enumerate([(5, 1), (9, 2), (3, 1), (7, 3), (4, 1)]):
        labels = torch.randint(0, no + 1, (n, 6, 8), generator=g, dtype=torch.uint8)
        pred = labels.clone()
        pred[1][pred[1] == 1] = 0
        assert np.array_equal(maps0[v], pred.numpy())
        j = metrics.jaccard_per_object(pred.long(), labels.long(), no)[1:-1]
        want_j_num += float(j.sum()); want_j_den += j.numel()
    assert abs(j0 - want_j_num / want_j_den) < 1e-12 and abs(j1 - j0) < 1e-12


def test_trunk_is_torchvision_resnet50(golden_dir):
    """The ResNet-50 stand-in shared by the product, the oracle and the golden generator against an
    EXTERNAL statement of torchvision's architecture (tests/golden/make_resnet50_kat.py): state-dict keys and
    shapes of conv1..layer3, per-stage parameter counts (9,536 / 215,808 / 1,219,584 / 7,098,368), where
    the stride sits, and the feature strides RMNet relies on (models/rmnet.py:57-64, 86-94)."""
    import json
    from rmnet_amd import networks
    kat = json.load(open(os.path.join(golden_dir, 'resnet50_trunk_kat.json')))
    trunk = networks.resnet50(pretrained=True)
    sd = trunk.state_dict()
    assert sorted(sd.keys()) == sorted(kat['state_dict'].keys())
    for k, shp in kat['state_dict'].items():
        assert list(sd[k].shape) == shp, k
    mods = dict(trunk.named_modules())
    for k, (stride, pad) in kat['conv_stride_padding'].items():
        c = mods[k]
        assert isinstance(c, torch.nn.Conv2d) and c.stride == (stride, stride) and c.padding == (pad, pad), k
        assert c.bias is None and c.dilation == (1, 1) and c.groups == 1
    count = lambda pre: sum(p.numel() for n, p in trunk.named_parameters() if n.split('.')[0] in pre)
    assert count(('conv1', 'bn1')) == kat['trainable_parameters']['stem']
    for st in ('layer1', 'layer2', 'layer3'):
        assert count((st,)) == kat['trainable_parameters'][st]
    mp = trunk.maxpool
    assert (mp.kernel_size, mp.stride, mp.padding) == (kat['maxpool']['kernel'], kat['maxpool']['stride'], kat['maxpool']['padding'])
    # feature strides / channels on a real input, through the encoder that uses the trunk
    from rmnet_amd.rmnet import RMNet
    net = RMNet(None).eval()
    with torch.no_grad():
        r4, r3, r2, c1, _ = net.encoder_query(torch.zeros(1, 3, 64, 96))
    for t, name in ((c1, 'stem'), (r2, 'layer1'), (r3, 'layer2'), (r4, 'layer3')):
        assert t.shape[2] == 64 // kat['output_stride'][name] and t.shape[3] == 96 // kat['output_stride'][name]
    assert (r2.shape[1], r3.shape[1], r4.shape[1]) == tuple(kat['output_channels'][k] for k in ('layer1', 'layer2', 'layer3'))
    # both encoders of RMNet carry exactly these tensors under the reference's names
    rs = net.state_dict()
    for k, shp in kat['state_dict'].items():
        ref_k = k.replace('layer1', 'res2').replace('layer2', 'res3').replace('layer3', 'res4')
        for enc in ('encoder_query', 'encoder_memory'):
            assert list(rs[enc + '.' + ref_k].shape) == shp


def test_compat_registers_the_reference_module_names():
    """``import rmnet_amd.compat`` puts the two compiled-module names the
    reference imports into sys.modules, with the reference's call signatures
    (reg_att_map_generator_cuda.cpp:26-38: forward(mask, float, int, int) -> [att, bboxes];
    flow_affine_transformation.cpp:87-90: update_optical_flow(flow, m1, m2))."""
    import inspect
    import sys
    for name in ('reg_att_map_generator', 'flow_affine_transformation'):
        sys.modules.pop(name, None)
    from rmnet_amd import compat
    compat.install()
    ram, fat = sys.modules['reg_att_map_generator'], sys.modules['flow_affine_transformation']
    assert list(inspect.signature(ram.forward).parameters) == ['mask', 'prob_threshold', 'n_pts_threshold', 'n_bbox_loose_pixels']
    assert list(inspect.signature(fat.update_optical_flow).parameters)[:3] == ['optical_flow', 'tr_matrix1', 'tr_matrix2']
    gen = ram.RegionalAttentionMapGenerator()
    sig = inspect.signature(gen.forward)
    assert [(p.name, p.default) for p in sig.parameters.values()][1:] == [('prob_threshold', 0.5), ('n_pts_threshold', 10), ('n_bbox_loose_pixels', 64)]
    sentinel = object()
    sys.modules['reg_att_map_generator'] = sentinel
    compat.install()                                   # does not clobber a module somebody else registered ...
    assert sys.modules['reg_att_map_generator'] is sentinel
    compat.install(force=True)                         # ... unless asked to
    assert sys.modules['reg_att_map_generator'] is ram
    with pytest.raises(RuntimeError, match='CUDA'):    # CHECK_CUDA of the pybind layer (.cpp:14)
        ram.forward(torch.zeros(1, 2, 8, 8), 0.5, 10, 64)
    # the reference's own wrapper package binds to it unchanged (build container only)
    ref = '/root/reference'
    if os.path.isdir(ref):
        import importlib
        sys.path.insert(0, ref)
        try:
            for k in [k for k in sys.modules if k == 'extensions' or k.startswith('extensions.')]:
                del sys.modules[k]
            pkg = importlib.import_module('extensions.reg_att_map_generator')
            assert pkg.reg_att_map_generator is ram
            assert hasattr(pkg, 'RegionalAttentionMapGenerator')
        finally:
            sys.path.remove(ref)
            for k in [k for k in sys.modules if k == 'extensions' or k.startswith('extensions.') or k == 'utils' or k.startswith('utils.')]:
                del sys.modules[k]


# YouTube-VOS-shaped variety (BASELINE configs[3]): mixed 480p / 720p label maps, 1-5 objects, 5-9 frames; only 5 videos
# for 8 ranks, so three ranks own nothing and must still take part in both collectives.
_YTVOS_VIDEOS = [(7, 480, 854, 1), (5, 720, 1280, 3), (9, 480, 854, 5), (6, 720, 1280, 2), (8, 480, 854, 4)]


def _ytvos_labels(v):
    n, H, W, no = _YTVOS_VIDEOS[v]
    g = torch.Generator().manual_seed(900 + v)
    return torch.randint(0, no + 1, (n, H // 8, W // 8), generator=g, dtype=torch.uint8)   # (1/8 size: the test moves ~1 MB)


def _eight_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    from rmnet_amd import dist as rd
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    rd.init_from_env(backend='gloo')
    costs = [n * no for n, _, _, no in _YTVOS_VIDEOS]
    mine = rd.my_videos(costs, rank, world)
    res = {v: _ytvos_labels(v) for v in mine}
    rd.barrier()
    out = rd.gather_label_maps(res, len(costs))
    n_owners = rd.sum_over_ranks(1.0 if mine else 0.0)
    slowest = rd.max_over_ranks(float(sum(costs[v] for v in mine)))
    if rank == 0:
        q.put((mine, {v: t.numpy().copy() for v, t in out.items()}, n_owners, slowest))
    else:
        assert out == {}
        q.put((mine, None, n_owners, slowest))
    dist.destroy_process_group()


def test_eight_rank_sharding_and_gather_with_idle_ranks_gloo():
    """BASELINE configs[3] without the hardware: 8 gloo ranks, 5 videos of mixed resolution / object count / length.
    assign_videos gives every video to exactly one rank (three ranks stay idle), every rank joins the header
    all_gather and the padded gather, rank 0 gets every label map bit for bit, and the reductions agree on all ranks."""
    import torch.multiprocessing as mp
    from rmnet_amd.dist import assign_videos
    world = 8
    costs = [n * no for n, _, _, no in _YTVOS_VIDEOS]
    owner = assign_videos(costs, world)
    assert sorted(owner) == sorted(set(owner)) and len(set(owner)) == 5      # 5 videos -> 5 distinct ranks
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    seen = sorted(v for mine, _, _, _ in got for v in mine)
    assert seen == [0, 1, 2, 3, 4]
    assert sum(1 for mine, _, _, _ in got if not mine) == 3                  # idle ranks took part and returned
    maps = [m for _, m, _, _ in got if m is not None]
    assert len(maps) == 1 and sorted(maps[0]) == [0, 1, 2, 3, 4]
    for v in range(5):
        assert np.array_equal(maps[0][v], _ytvos_labels(v).numpy())
    assert all(n == 5.0 for _, _, n, _ in got)
    assert len({s for _, _, _, s in got}) == 1 and got[0][3] == float(max(costs))


def test_fused_epilogue_snapshots_follow_submodule_checkpoint_loads():
    """Round-2 advisor finding: the (scale, shift) snapshots of fuse_epilogues() were refreshed only when the TOP module
    loaded a checkpoint.  Loading into a sub-module must refresh them too, and fuse_epilogues(False) removes the hooks."""
    from rmnet_amd import networks
    from rmnet_amd.rmnet import RMNet
    net = networks.procedural_init_(RMNet(None)).eval()
    net.fuse_epilogues()
    blk = [m for m in net.encoder_query.modules() if isinstance(m, networks._Bottleneck)][0]
    before = blk._s1.clone()
    sd = net.encoder_query.state_dict()
    for k in list(sd):
        if k.endswith('running_var'):
            sd[k] = sd[k] * 4.0
    net.encoder_query.load_state_dict(sd)
    assert torch.allclose(blk._s1, before * 0.5, rtol=1e-4)            # 1 / sqrt(4 var): the snapshot followed the load
    stem = net.encoder_query
    assert torch.allclose(stem._s1, networks._bn_scale_shift(stem.bn1)[0])
    net.fuse_epilogues(False)
    assert all(getattr(m, '_fuse_hook', None) is None for m in net.modules())


def test_launch_plan_invariants_on_the_host(tmp_path):
    """tests/native/plan_check.cpp: the real bank_chunks() of csrc/common.h under the chunk walk and the pair slot lists of
    bk_main (restated there), over a dense sweep of single objects and ~400 whole launches x 5 workgroup budgets x both
    arithmetic modes: every (query tile, memory tile) walked once, slots unique and inside the budget, a pair's segments at
    the positions its closed-form slot list names, the search for the chunk length ending inside the workgroup budget.
    (Both plan bugs of round 3 -- a negative chunk count for objects without query tiles, own-column-block plans that did
    not fit the launch -- are of the kind this catches without a GPU.)"""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    exe = str(tmp_path / 'plan_check')
    cc = subprocess.run([hipcc, '-O1', '-std=c++17', '-w', '--offload-arch=gfx950', os.path.join(ROOT, 'tests', 'native', 'plan_check.cpp'), '-o', exe],
                        capture_output=True, text=True, timeout=600)
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and 'plan_check ok' in run.stdout, run.stdout[-3000:]


def test_bench_becomes_its_own_launcher_for_several_gpus(monkeypatch):
    """`python bench.py --gpus N` from a plain shell (WORLD_SIZE unset): bench.py re-executes its own command line as N ranks under
    torch.distributed.run on 127.0.0.1 at a free port, with dmabuf IPC for RCCL, and hands the exit code back.  (The GPU suite runs
    the real thing with two gloo ranks on one GPU: test_bench_launches_its_own_ranks.)"""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    monkeypatch.delenv('HSA_ENABLE_IPC_MODE_LEGACY', raising=False)
    assert bench._self_launch(4) == 7
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert 1024 <= int(cmd[cmd.index('--master-port') + 1]) <= 65535
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert bench.DEFAULT_CLIPS == 16


def test_auto_read_precision_start_is_sticky_per_network():
    """``read_precision='auto'`` (rmnet_amd/rmnet.py): one-object clips start in 'f16', clips with several objects in 'split'; once a
    one-object clip of this network measured logits beyond AUTO_LOGIT_BOUND (the flag ``forward`` sets after the re-read) its later
    one-object clips start in 'split' -- no clip is processed twice again -- until new reference weights are loaded / the rule is
    reset.  A forced arithmetic ignores the flag.  (The measured half of the rule runs on the GPU: test_key_temperature_sweep.)"""
    from rmnet_amd.rmnet import RMNet
    net = RMNet(None)
    assert net.resolve_read_precision([1]) == 'f16' and net.resolve_read_precision([1, 1]) == 'f16'
    assert net.resolve_read_precision([1, 3]) == 'split'
    net._auto_peaked = True                       # what forward() leaves behind after re-reading a peaked clip in 'split'
    assert net.resolve_read_precision([1]) == 'split' and net.resolve_read_precision([3]) == 'split'
    net.reset_auto_precision()
    assert net.resolve_read_precision([1]) == 'f16'
    net._auto_peaked = True
    net.load_reference_state_dict({'module.' + k: v for k, v in net.state_dict().items()})      # new weights: a new logit scale
    assert net.resolve_read_precision([1]) == 'f16'
    forced = RMNet(None, read_precision='f16')
    forced._auto_peaked = True
    assert forced.resolve_read_precision([1]) == 'f16' and forced.resolve_read_precision([5]) == 'f16'
