# -*- coding: utf-8 -*-
"""Parity of the gfx950 kernels (through the C ABI) with the oracle and the golden vectors.
Every test here needs a real MI355X: ``pytest -m gpu``.

Bars (BASELINE.json north_star): integer / index outputs (boxes, cell rectangles, 0/1 maps, the
integer-valued flow update) bit-exact; the fp32 memory read within MR_ATOL of the oracle (fp32
rounding only -- the kernel accumulates in fp32 MFMA, the oracle in double); end-to-end masks:
probabilities within 1e-3 and label IoU >= 0.999 of the CPU path.
"""

import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MR_ATOL = 3e-5     # absolute, for O(1) inputs / outputs
MR_RTOL = 2e-5


def dev():
    assert torch.cuda.is_available(), 'these tests need the GPU box'
    return torch.device('cuda', 0)


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


def test_native_library_is_what_runs():
    from rmnet_amd import _lib, ops
    ops.region_map(torch.zeros(1, 2, 16, 16, device=dev()))
    maps = open('/proc/self/maps').read()
    assert 'librmnet_hip.so' in maps
    assert os.path.samefile(_lib.LIB_PATH, os.path.join(os.path.dirname(_lib.__file__), 'librmnet_hip.so'))


# ----------------------------------------------------------------------------- region map (G1)
def _kat_mask(c):
    m = np.zeros((c['B'], c['K'], c['H'], c['W']), np.float32)
    for f in c['fills']:
        m[f['b'], f['k'], f['y0']:f['y1'] + 1, f['x0']:f['x1'] + 1] = f['v']
    return m


def test_region_map_known_answers(golden_dir, oracle_mod):
    from rmnet_amd.reg_att_map_generator import RegionalAttentionMapGenerator
    gen = RegionalAttentionMapGenerator()
    kat = json.load(open(os.path.join(golden_dir, 'region_map_kat.json')))
    for c in kat['cases']:
        m = _kat_mask(c)
        att, bb = gen(cu(m), c['thr'], c['npts'], c['loose'])
        assert bb.dtype == torch.int32 and att.dtype == torch.float32
        assert (bb.cpu().numpy() == np.array(c['bboxes'], np.int32)).all(), c['name']
        o_att, o_bb = oracle_mod.region_map(m, c['thr'], c['npts'], c['loose'])
        assert np.array_equal(att.cpu().numpy(), o_att), c['name']


@pytest.mark.parametrize('B,K,H,W', [(1, 2, 480, 864), (1, 11, 480, 854), (2, 3, 150, 250), (1, 4, 33, 47),
                                     (1, 2, 720, 1280), (3, 2, 1, 5)])
def test_region_map_random_soft_masks_bitexact(B, K, H, W, oracle_mod):
    from rmnet_amd import ops
    rng = np.random.RandomState(H * 7 + W)
    m = np.zeros((B, K, H, W), np.float32)
    for b in range(B):
        for k in range(K):
            if rng.rand() < 0.2:
                continue                      # empty channel -> full-frame fallback
            y0, x0 = rng.randint(0, H), rng.randint(0, W)
            y1, x1 = rng.randint(y0, H) + 1, rng.randint(x0, W) + 1
            m[b, k, y0:y1, x0:x1] = rng.rand(y1 - y0, x1 - x0) * 1.2
    m[m > 1] = 0.5                            # exercise the inclusive threshold
    o_att, o_bb = oracle_mod.region_map(m)
    att, bb, _ = ops.region_map(cu(m))
    assert np.array_equal(bb.cpu().numpy(), o_bb)
    assert np.array_equal(att.cpu().numpy(), o_att)
    # boxes-only form + cell rectangles (what the frame loop uses)
    lw, lh = ((16 - W % 16) % 16) // 2, ((16 - H % 16) % 16) // 2
    h, w = (H + 15) // 16, (W + 15) // 16
    none_att, bb2, rects = ops.region_map(cu(m), want_map=False, cell_grid=(lw, lh, 16, h, w))
    assert none_att is None and np.array_equal(bb2.cpu().numpy(), o_bb)
    assert np.array_equal(rects.cpu().numpy(), oracle_mod.cell_rects(o_bb, lw, lh, h, w))
    r2 = ops.boxes_to_cell_rects(bb2, lw, lh, 16, h, w, k_per_batch=K)
    assert np.array_equal(r2.cpu().numpy(), rects.cpu().numpy())


def test_region_map_rejects_bad_input():
    from rmnet_amd import ops
    x = torch.zeros(1, 2, 8, 8, device=dev())
    with pytest.raises(RuntimeError, match='contiguous'):
        ops.region_map(x.transpose(2, 3))
    with pytest.raises(RuntimeError, match='float32'):
        ops.region_map(x.double())


# ----------------------------------------------------------------------------- flow affine (F1)
def test_flow_affine_golden_bitexact(golden_dir):
    from rmnet_amd import flow_affine_transformation as fat
    g = np.load(os.path.join(golden_dir, 'flow_affine.npz'))
    for n in sorted({k.split('.')[0] for k in g.files}):
        out = fat.update_optical_flow(g[n + '.flow'], g[n + '.m1'], g[n + '.m2'])      # NumPy convention
        assert isinstance(out, np.ndarray) and out.dtype == np.float32
        assert np.array_equal(out.view(np.uint32), g[n + '.out'].view(np.uint32)), n
        out2 = fat.update_optical_flow_cuda(cu(g[n + '.flow']), cu(g[n + '.m1']), cu(g[n + '.m2']))
        assert np.array_equal(out2.cpu().numpy().view(np.uint32), g[n + '.out'].view(np.uint32)), n


@pytest.mark.parametrize('H,W', [(480, 854), (720, 1280), (37, 53), (1, 1)])
def test_flow_affine_random_bitexact(H, W, oracle_mod):
    from rmnet_amd import ops
    rng = np.random.RandomState(H + W)
    for amp, jitter in [(3.0, 0.05), (60.0, 0.5), (2000.0, 3.0)]:
        flow = ((rng.rand(H, W, 2) - 0.5) * amp).astype(np.float32)
        m1 = (np.eye(2, 3) + (rng.rand(2, 3) - 0.5) * jitter).astype(np.float32)
        m2 = (np.eye(2, 3) + (rng.rand(2, 3) - 0.5) * jitter).astype(np.float32)
        want = oracle_mod.flow_affine(flow, m1, m2)
        got = ops.flow_affine(cu(flow), cu(m1), cu(m2)).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (amp, jitter)


# ----------------------------------------------------------------------------- memory read (M1-M3)
def _mr(g, name, **kw):
    from rmnet_amd import ops
    args = [cu(g[name + '.m_key']), cu(g[name + '.m_val']), cu(g[name + '.q_key']), cu(g[name + '.q_val'])]
    return ops.memory_read(*args, **kw)


def test_memory_read_golden_dense(golden_dir):
    g = np.load(os.path.join(golden_dir, 'memory_reader.npz'))
    from rmnet_amd import ops
    for name in ['dense', 'regional', 'allmasked_query', 'peaky']:
        out, p = _mr(g, name)                      # default: transient split-fp16 bank inside the call
        assert p is None
        np.testing.assert_allclose(out.cpu().numpy(), g[name + '.mem_val'], atol=MR_ATOL, rtol=MR_RTOL,
                                   err_msg=name)
        out_x, _ = _mr(g, name, flags=ops.MR_EXACT_FP32)   # exact-fp32 MFMA kernel (mr_main)
        np.testing.assert_allclose(out_x.cpu().numpy(), g[name + '.mem_val'], atol=MR_ATOL, rtol=MR_RTOL,
                                   err_msg=name + ' exact')
        out_g, _ = _mr(g, name, flags=1)          # generic kernels, same answer
        np.testing.assert_allclose(out_g.cpu().numpy(), g[name + '.mem_val'], atol=MR_ATOL, rtol=MR_RTOL)


def test_memory_read_golden_regional_fused_mask(golden_dir):
    """Rectangles + (already masked OR un-masked) inputs == the reference on masked inputs."""
    g = np.load(os.path.join(golden_dir, 'memory_reader.npz'))
    rng = np.random.RandomState(0)
    for name in ['regional', 'allmasked_query']:
        mr, qr = cu(g[name + '.mem_rects']), cu(g[name + '.qry_rects'])
        out, _ = _mr(g, name, mem_rects=mr, qry_rects=qr)
        np.testing.assert_allclose(out.cpu().numpy(), g[name + '.mem_val'], atol=MR_ATOL, rtol=MR_RTOL)
        # put garbage where the masks are zero: the fused kernel must ignore it
        dirty = {}
        for k in ('m_key', 'm_val', 'q_key', 'q_val'):
            a = g[name + '.' + k].copy()
            a[a == 0] = rng.randn(int((a == 0).sum())).astype(np.float32) * 3
            dirty[k] = cu(a)
        from rmnet_amd import ops
        out2, _ = ops.memory_read(dirty['m_key'], dirty['m_val'], dirty['q_key'], dirty['q_val'], mr, qr)
        np.testing.assert_allclose(out2.cpu().numpy(), g[name + '.mem_val'], atol=MR_ATOL, rtol=MR_RTOL)
        out3, _ = ops.memory_read(dirty['m_key'], dirty['m_val'], dirty['q_key'], dirty['q_val'], mr, qr, flags=1)
        np.testing.assert_allclose(out3.cpu().numpy(), g[name + '.mem_val'], atol=MR_ATOL, rtol=MR_RTOL)


def test_memory_read_affinity_output(golden_dir):
    from rmnet_amd.rmnet import MemoryReader
    g = np.load(os.path.join(golden_dir, 'memory_reader.npz'))
    out, p = _mr(g, 'tiny_p', want_p=True)        # De=16/Do=32 -> generic path
    np.testing.assert_allclose(out.cpu().numpy(), g['tiny_p.mem_val'], atol=MR_ATOL, rtol=MR_RTOL)
    np.testing.assert_allclose(p.cpu().numpy(), g['tiny_p.p'], atol=1e-6, rtol=1e-5)
    reader = MemoryReader(return_affinity=True)   # fast path + p on request
    out, p = reader(cu(g['dense.m_key']), cu(g['dense.m_val']), cu(g['dense.q_key']), cu(g['dense.q_val']))
    np.testing.assert_allclose(out.cpu().numpy(), g['dense.mem_val'], atol=MR_ATOL, rtol=MR_RTOL)
    assert p.shape == (1, 2 * 6 * 8, 6 * 8)
    np.testing.assert_allclose(p.sum(1).cpu().numpy(), 1.0, atol=1e-5)


def _random_case(rng, no, T, h, w, scale=0.6, regional=True):
    mk = (rng.randn(no, 128, T, h, w) * scale).astype(np.float32)
    mv = rng.randn(no, 512, T, h, w).astype(np.float32)
    qk = (rng.randn(no, 128, h, w) * scale).astype(np.float32)
    qv = rng.randn(no, 512, h, w).astype(np.float32)
    mr = qr = None
    if regional:
        def rect():
            if rng.rand() < 0.15:
                return (1, 0, 1, 0)
            x0, y0 = rng.randint(0, w), rng.randint(0, h)
            return (x0, rng.randint(x0, w), y0, rng.randint(y0, h))
        mr = np.array([[rect() for _ in range(T)] for _ in range(no)], np.int32)
        qr = np.array([rect() for _ in range(no)], np.int32)
    return mk, mv, qk, qv, mr, qr


@pytest.mark.parametrize('no,T,h,w,regional', [
    (1, 1, 4, 5, False), (2, 3, 9, 13, True), (3, 2, 12, 20, True), (1, 5, 30, 54, True),
    (1, 4, 8, 8, True), (1, 7, 16, 24, False)])
def test_memory_read_random_vs_oracle(no, T, h, w, regional, oracle_mod):
    from rmnet_amd import ops
    rng = np.random.RandomState(no * 1000 + T * 100 + h)
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=regional)
    for flags in (0, ops.MR_EXACT_FP32):
        if regional:
            want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
            got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr), flags=flags)
        else:
            want, _ = oracle_mod.memory_read(mk, mv, qk, qv)
            got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), flags=flags)
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)


def test_memory_read_edge_rectangles(oracle_mod):
    """Empty memory, empty query, full rectangles, single cells, a query count that is an exact
    multiple of the 64-query tile (the mean slot then needs its own tile)."""
    from rmnet_amd import ops
    rng = np.random.RandomState(11)
    h, w, T = 8, 16, 2
    mk, mv, qk, qv, _, _ = _random_case(rng, 1, T, h, w, regional=False)
    full, empty = (0, w - 1, 0, h - 1), (1, 0, 1, 0)
    cases = [([empty, empty], full), ([full, full], empty), ([full, full], full), ([empty, (3, 3, 2, 2)], (5, 5, 7, 7)),
             ([(0, 15, 0, 3), full], (0, 15, 0, 3)),      # 64 unmasked queries exactly
             ([(-3, 40, -2, 90), empty], (-1, 99, 2, 5))]  # rectangles sticking out are clamped
    for mrect, qrect in cases:
        mr, qr = np.array([mrect], np.int32), np.array([qrect], np.int32)
        clamp = lambda r: (max(r[0], 0), min(r[1], w - 1), max(r[2], 0), min(r[3], h - 1))
        want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, np.array([[clamp(r) for r in mrect]], np.int32),
                                                  np.array([clamp(qrect)], np.int32))
        for flags in (0, ops.MR_EXACT_FP32):
            got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr), flags=flags)
            np.testing.assert_allclose(got.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL, err_msg=str((mrect, qrect)))


def test_memory_read_bank_strides_and_running_max(oracle_mod):
    """A bank with capacity > T is read through the channel stride; a late, much larger logit
    forces the deferred soft-max reference to be bumped (rescale branch)."""
    from rmnet_amd import ops
    rng = np.random.RandomState(5)
    no, T, Tcap, h, w = 2, 3, 5, 6, 10
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, Tcap, h, w, regional=True)
    mk[:, :, 2, h - 1, w - 1] = qk[:, :, 2, 3] * 9.0          # spike late in the memory: S jumps by >> 30
    mr[:, 2] = (0, w - 1, 0, h - 1)
    qr[:] = (0, w - 1, 0, h - 1)
    want, _ = oracle_mod.regional_memory_read(mk[:, :, :T], mv[:, :, :T], qk, qv, mr[:, :T], qr)
    want_d, _ = oracle_mod.memory_read(mk[:, :, :T], mv[:, :, :T], qk, qv)
    for flags in (0, ops.MR_EXACT_FP32):
        got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr[:, :T]), cu(qr), T=T, flags=flags)
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)
        got_d, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), T=T, flags=flags)
        np.testing.assert_allclose(got_d.cpu().numpy(), want_d, atol=MR_ATOL, rtol=MR_RTOL)


@pytest.mark.parametrize('no,T,h,w', [(1, 5, 30, 54), (3, 20, 45, 80)])
def test_memory_read_full_size_properties(no, T, h, w):
    """BASELINE.json sizes (480p T=5; 720p T=20, 3 objects), checked through size-independent
    properties instead of the (slow) oracle: regional == dense-on-premasked, soft-max rows sum to one
    (V = 1 reads back 1), and linearity in the values."""
    from rmnet_amd import ops
    rng = np.random.RandomState(T)
    g = torch.Generator(device='cpu').manual_seed(T)
    mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6).to(dev())
    mv = torch.randn(no, 512, T, h, w, generator=g).to(dev())
    qk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev())
    qv = torch.randn(no, 512, h, w, generator=g).to(dev())
    mr = np.zeros((no, T, 4), np.int32)
    qr = np.zeros((no, 4), np.int32)
    for o in range(no):
        for t in range(T):
            x0, y0 = rng.randint(0, w // 2), rng.randint(0, h // 2)
            mr[o, t] = (x0, x0 + w // 2, y0, y0 + h // 2)
        x0, y0 = rng.randint(0, w // 3), rng.randint(0, h // 3)
        qr[o] = (x0, x0 + w // 2, y0, y0 + h // 2)
    mr_t, qr_t = cu(mr), cu(qr)
    out_r, _ = ops.memory_read(mk, mv, qk, qv, mr_t, qr_t)
    mk_m, mv_m = ops.rect_mask(mk, mr_t), ops.rect_mask(mv, mr_t)
    qk_m = ops.rect_mask(qk.unsqueeze(2).contiguous(), qr_t.view(no, 1, 4)).squeeze(2)
    qv_m = ops.rect_mask(qv.unsqueeze(2).contiguous(), qr_t.view(no, 1, 4)).squeeze(2)
    out_d, _ = ops.memory_read(mk_m, mv_m, qk_m, qv_m)
    assert torch.allclose(out_r, out_d, atol=MR_ATOL, rtol=MR_RTOL)
    assert torch.equal(out_r[:, 512:], qv_m)                       # the cat half is q_val * box, exactly
    ones, _ = ops.memory_read(mk, torch.ones_like(mv), qk, qv)
    assert torch.allclose(ones[:, :512], torch.ones_like(ones[:, :512]), atol=1e-5)
    a, _ = ops.memory_read(mk, mv, qk, qv)
    b, _ = ops.memory_read(mk, 2.0 * mv + 1.0, qk, qv)
    assert torch.allclose(b[:, :512], 2.0 * a[:, :512] + 1.0, atol=1e-4, rtol=1e-4)


def _fill_bank(ops, mk, mv, mr, capacity=None):
    no, _, T, h, w = mk.shape
    bank = ops.MemoryBank(no, capacity or T, h, w, dev())
    for t in range(T):
        rects = None if mr is None else cu(mr[:, t])
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), rects)
    return bank


@pytest.mark.parametrize('no,T,h,w,regional', [
    (1, 1, 4, 5, False), (2, 3, 9, 13, True), (3, 2, 12, 20, True), (1, 5, 30, 54, True),
    (1, 4, 8, 8, True), (1, 7, 16, 24, False), (2, 9, 10, 7, True),
    (14, 2, 6, 9, True),      # > 12 objects: the launch plan's LDS-atomic path
    (70, 1, 4, 5, True),      # > 64 objects: two launch groups
    (5, 3, 30, 54, True),     # boxes of very different sizes planned together
    (5, 5, 30, 54, True),     # BASELINE configs[2] at its exact shape: 5 objects, T = 5, 480p grid
    (50, 2, 12, 20, False),   # 200 equal pairs of 16 tiles on ~230 workgroups: every object a column block of its own (C > njt)
    (24, 3, 12, 20, True),    # ragged version of the same regime
    (64, 2, 16, 24, False),   # [r6] 384 equal pairs on ~230 workgroups: the launch runs in ROUNDS (whole objects first, the rest cut in blocks)
    (60, 3, 16, 24, True),    # [r6] ragged version: pairs of 1-6 query tiles, more pairs than workgroups for most seeds
    (20, 1, 30, 54, False),   # [r6] 20 dense 480p objects = 520 pairs: two whole rounds + a cut round (the launch size where round 5 fell to 0.39)
    (1, 3, 30, 54, True)])    # BASELINE configs[0]: 1 object, 3 memory frames
def test_bank_read_vs_oracle(no, T, h, w, regional, oracle_mod):
    """The split-fp16 bank path (what the frame loop uses) against the oracle, same tolerance as the
    fp32-MFMA path."""
    from rmnet_amd import ops
    rng = np.random.RandomState(no * 1000 + T * 100 + h + 1)
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=regional)
    bank = _fill_bank(ops, mk, mv, mr, capacity=T + 2)
    if regional:
        want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
        got = bank.read(T, cu(qk), cu(qv), cu(qr))
    else:
        want, _ = oracle_mod.memory_read(mk, mv, qk, qv)
        got = bank.read(T, cu(qk), cu(qv))
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)


def test_bank_edge_rectangles_overwrite_and_running_max(oracle_mod):
    from rmnet_amd import ops
    rng = np.random.RandomState(21)
    h, w, T = 8, 16, 3
    mk, mv, qk, qv, _, _ = _random_case(rng, 1, T, h, w, regional=False)
    full, empty = (0, w - 1, 0, h - 1), (1, 0, 1, 0)
    mk[:, :, 2, h - 1, w - 1] = qk[:, :, 2, 3] * 9.0       # late spike: forces the deferred max to bump
    for mrect, qrect in [([empty] * 3, full), ([full] * 3, empty), ([full] * 3, full),
                         ([empty, (3, 3, 2, 2), empty], (5, 5, 7, 7)), ([(0, 15, 0, 3), full, full], (0, 15, 0, 3)),
                         ([(2, 9, 1, 6), empty, full], (1, 14, 0, 7))]:
        mr, qr = np.array([mrect], np.int32), np.array([qrect], np.int32)
        bank = _fill_bank(ops, mk, mv, mr)
        want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
        got = bank.read(T, cu(qk), cu(qv), cu(qr))
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL, err_msg=str((mrect, qrect)))
        # reading fewer frames ignores the later slots; overwriting a slot replaces it (tentative frame)
        want2, _ = oracle_mod.regional_memory_read(mk[:, :, :2], mv[:, :, :2], qk, qv, mr[:, :2], qr)
        np.testing.assert_allclose(bank.read(2, cu(qk), cu(qv), cu(qr)).cpu().numpy(), want2, atol=MR_ATOL, rtol=MR_RTOL)
        bank.append(1, cu(mk[:, :, 0]), cu(mv[:, :, 0]), cu(mr[:, 0]))
        mk3, mv3, mr3 = mk.copy(), mv.copy(), mr.copy()
        mk3[:, :, 1], mv3[:, :, 1], mr3[:, 1] = mk[:, :, 0], mv[:, :, 0], mr[:, 0]
        want3, _ = oracle_mod.regional_memory_read(mk3, mv3, qk, qv, mr3, qr)
        np.testing.assert_allclose(bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy(), want3, atol=MR_ATOL, rtol=MR_RTOL)


@pytest.mark.parametrize('no,T,h,w', [(1, 5, 30, 54), (3, 20, 45, 80)])
def test_bank_full_size_agrees_with_fp32_kernel(no, T, h, w):
    """BASELINE sizes: the split-fp16 bank read == the exact-fp32 MFMA read (two independent kernels,
    different number formats and memory layouts) to fp32 rounding."""
    from rmnet_amd import ops
    rng = np.random.RandomState(T + 3)
    g = torch.Generator(device='cpu').manual_seed(T)
    mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6).to(dev())
    mv = torch.randn(no, 512, T, h, w, generator=g).to(dev())
    qk = (torch.randn(no, 128, h, w, generator=g) * 0.6).to(dev())
    qv = torch.randn(no, 512, h, w, generator=g).to(dev())
    mr = np.zeros((no, T, 4), np.int32)
    qr = np.zeros((no, 4), np.int32)
    for o in range(no):
        for t in range(T):
            x0, y0 = rng.randint(0, w // 2), rng.randint(0, h // 2)
            mr[o, t] = (x0, x0 + w // 2, y0, y0 + h // 2)
        x0, y0 = rng.randint(0, w // 3), rng.randint(0, h // 3)
        qr[o] = (x0, x0 + w // 2, y0, y0 + h // 2)
    bank = ops.MemoryBank(no, T, h, w, dev())
    for t in range(T):
        bank.append(t, mk[:, :, t].contiguous(), mv[:, :, t].contiguous(), cu(mr[:, t]))
    got = bank.read(T, qk, qv, cu(qr))
    want, _ = ops.memory_read(mk, mv, qk, qv, cu(mr), cu(qr), flags=ops.MR_EXACT_FP32)
    assert torch.allclose(got, want, atol=MR_ATOL, rtol=MR_RTOL)
    via, _ = ops.memory_read(mk, mv, qk, qv, cu(mr), cu(qr))     # drop-in entry = staging + the same bank read
    assert torch.allclose(via, want, atol=MR_ATOL, rtol=MR_RTOL)
    full = ops.MemoryBank(no, T, h, w, dev())
    for t in range(T):
        full.append(t, mk[:, :, t].contiguous(), torch.ones_like(mv[:, :, t]).contiguous())
    ones = full.read(T, qk, qv)
    assert torch.allclose(ones[:, :512], torch.ones_like(ones[:, :512]), atol=1e-5)   # soft-max rows sum to 1


def test_rect_mask_vs_oracle(oracle_mod):
    from rmnet_amd import ops
    rng = np.random.RandomState(2)
    x = rng.randn(3, 7, 2, 6, 9).astype(np.float32)
    r = np.array([[(0, 8, 0, 5), (1, 0, 1, 0)], [(2, 4, 1, 3), (8, 8, 5, 5)], [(0, 0, 0, 0), (3, 7, 2, 2)]], np.int32)
    assert np.array_equal(ops.rect_mask(cu(x), cu(r)).cpu().numpy(), oracle_mod.rect_mask(x, r))


# ----------------------------------------------------------------------------- the frame loop (P1-P5)
def _nets(oracle_mod, read_precision=None):
    """The product (frame loop in its DEFAULT arithmetic unless ``read_precision`` is given) and the CPU oracle, same weights."""
    from rmnet_amd import networks
    from rmnet_amd.rmnet import RMNet
    prod = RMNet(None) if read_precision is None else RMNet(None, read_precision=read_precision)
    networks.procedural_init_(prod)
    ref = oracle_mod.OracleRMNet()
    ref.load_state_dict(prod.state_dict())
    return prod.to(dev()).eval(), ref.eval()


def test_rmnet_clip_matches_cpu_path_and_reference(golden_dir, oracle_mod):
    from rmnet_amd.synthetic import synthetic_clip
    g = np.load(os.path.join(golden_dir, 'rmnet_clip.npz'))
    prod, ref = _nets(oracle_mod)
    N, K, H, W = int(g['clip.N']), int(g['clip.K']), int(g['clip.H']), int(g['clip.W'])
    frames, masks, flows, n_objects = synthetic_clip(N, K, H, W, seed=int(g['clip.seed']))
    with torch.no_grad():
        est = prod(frames, masks, flows, n_objects, int(g['clip.memorize_every']))
        assert est.is_cuda
        est = est.cpu()
        est_cpu = ref(frames, masks, flows, n_objects, int(g['clip.memorize_every']))
    # vs the oracle's CPU path (same weights, same inputs)
    assert float((est - est_cpu).abs().max()) < 1e-3
    lab, lab_cpu = est.argmax(2).numpy(), est_cpu.argmax(2).numpy()
    for k in range(1, K):
        assert oracle_mod.iou(lab[:, 1:] == k, lab_cpu[:, 1:] == k) >= 0.999
    # vs the reference itself (golden)
    np.testing.assert_allclose(est[:, 1].numpy(), g['clip.est_t1'], atol=1e-3)
    assert (lab == g['clip.est_argmax']).mean() > 0.999


def test_rmnet_pieces_keep_reference_contract(golden_dir, oracle_mod):
    from rmnet_amd.synthetic import synthetic_clip
    g = np.load(os.path.join(golden_dir, 'rmnet_clip.npz'))
    prod, ref = _nets(oracle_mod)
    N, K, H, W = int(g['clip.N']), int(g['clip.K']), int(g['clip.H']), int(g['clip.W'])
    frames, masks, flows, n_objects = synthetic_clip(N, K, H, W, seed=int(g['clip.seed']))
    d = dev()
    with torch.no_grad():
        k4, v4, bb = prod.memorize(frames[:, 0].to(d), masks[:, 0].float().to(d), [K - 1])
        assert (bb.cpu().numpy() == g['memorize.bboxes']).all()
        np.testing.assert_allclose(k4.cpu().numpy(), g['memorize.k4'], atol=2e-4, rtol=1e-3)
        np.testing.assert_allclose(v4[:, :, ::4].cpu().numpy(), g['memorize.v4_every4'], atol=2e-4, rtol=1e-3)
        warped, valid = prod.warp(cu(g['warp.in']), cu(g['warp.flow']))
        np.testing.assert_allclose(warped.cpu().numpy(), g['warp.out'], atol=1e-4)   # torch grid_sample, GPU vs CPU
        att, box = prod.get_att_map(cu(g['warp.in']), cu(g['warp.flow']))
        o_att, o_box = ref.get_att_map(torch.from_numpy(g['warp.in']), torch.from_numpy(g['warp.flow']))
        assert np.array_equal(box.cpu().numpy(), o_box.numpy()) and np.array_equal(att.cpu().numpy(), o_att.numpy())
        logit = prod.soft_aggregation(cu(g['softagg.ps']), K, [K - 1])
        np.testing.assert_allclose(logit.cpu().numpy(), g['softagg.logit'], atol=1e-5)
        # public segment(): reference argument list, fused kernels inside
        k4_cpu, v4_cpu, bb_cpu = ref.memorize(frames[:, 0], masks[:, 0].float(), [K - 1])
        att_cpu, cbb_cpu = ref.get_att_map(masks[:, 0].float(), flows[:, 1])
        want = ref.segment(frames[:, 1], att_cpu, k4_cpu, v4_cpu, [K - 1])
        got = prod.segment(frames[:, 1].to(d), att_cpu.to(d), k4, v4, bb.unsqueeze(2), cbb_cpu.to(d), [K - 1])
        assert float((got.cpu() - want).abs().max()) < 2e-3


def test_tiny_flownet_on_gpu(golden_dir):
    from rmnet_amd import networks
    from rmnet_amd.tiny_flownet import TinyFlowNet
    g = np.load(os.path.join(golden_dir, 'tiny_flownet.npz'))
    net = networks.procedural_init_(TinyFlowNet(None)).to(dev()).eval()
    with torch.no_grad():
        fl = net(cu(g['frames']))
    np.testing.assert_allclose(fl.cpu().numpy(), g['flows'], atol=2e-3, rtol=1e-3)


# ----------------------------------------------------------------------------- wider loop coverage
def _clip_with_late_object(N, H, W, seed):
    """K = 3 clip in which object 2 only appears at frame 2 (n_objects changes mid-clip): exercises
    the new-object injection and the 'non-existing object' logits of models/rmnet.py:436-448."""
    from rmnet_amd.synthetic import synthetic_clip
    frames, masks, flows, _ = synthetic_clip(N, 3, H, W, seed=seed, size=1.3)
    masks = masks.clone()
    for t in range(2):                       # object 2 is background in frames 0 and 1
        masks[0, t, 0] = masks[0, t, 0] | masks[0, t, 2]
        masks[0, t, 2] = 0
    n_objects = torch.tensor([[1, 1] + [2] * (N - 2)], dtype=torch.long)
    return frames, masks, flows, n_objects


@pytest.mark.parametrize('fused', [False, True])
def test_rmnet_new_object_and_batch_of_clips(fused, oracle_mod):
    """Two clips batched, one gaining an object mid-clip (models/rmnet.py:436-448): the module graph
    and the fused path (epilogue kernels, warp-fused boxes, one-kernel decoder tail -- whose soft-max
    must be bypassed on the frames where the logits are edited) against the CPU path."""
    prod, ref = _nets(oracle_mod)
    if fused:
        prod.fuse_epilogues()
    f1, m1, fl1, n1 = _clip_with_late_object(5, 96, 160, seed=11)
    f2, m2, fl2, n2 = _clip_with_late_object(5, 96, 160, seed=12)
    n2 = torch.full_like(n2, 2)
    from rmnet_amd.synthetic import synthetic_clip
    f2, m2, fl2, _ = synthetic_clip(5, 3, 96, 160, seed=12, size=1.3)
    frames, masks = torch.cat([f1, f2]), torch.cat([m1, m2])
    flows, n_objects = torch.cat([fl1, fl2]), torch.cat([n1, n2])
    with torch.no_grad():
        est = prod(frames, masks, flows, n_objects, 2).cpu()
        est_cpu = ref(frames, masks, flows, n_objects, 2)
    assert float((est - est_cpu).abs().max()) < 1e-3
    lab, lab_cpu = est.argmax(2).numpy(), est_cpu.argmax(2).numpy()
    assert (lab == lab_cpu).mean() > 0.999
    # before it appears, object 2 of clip 0 has (numerically) zero probability
    assert float(est[0, 1, 2].max()) < 1e-6


def test_frame_step_is_graph_capturable(oracle_mod):
    """No host synchronisation inside a frame step: it can be captured into a HIP graph and replayed
    (boxes, rectangles and the split plan all stay on the device)."""
    from rmnet_amd.synthetic import synthetic_clip
    prod, _ = _nets(oracle_mod)
    d = dev()
    frames, masks, flows, _ = synthetic_clip(4, 2, 96, 160, seed=5, size=1.5)
    frames, masks, flows = frames.to(d), masks.to(d).float(), flows.to(d)
    ctx = prod._ClipContext(prod, 1, 2, 96, 160, [1], d)
    bank = prod.new_bank(ctx, 3)
    with torch.no_grad():
        prod.frame_step(ctx, bank, frames[:, 0], masks[:, 0], frames[:, 1], flows[:, 1], commit=True)
        eager = prod.frame_step(ctx, bank, frames[:, 1], masks[:, 1], frames[:, 2], flows[:, 2], commit=False).clone()
        side = torch.cuda.Stream(d)
        side.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(side):
            prod.frame_step(ctx, bank, frames[:, 1], masks[:, 1], frames[:, 2], flows[:, 2], commit=False)
        torch.cuda.current_stream(d).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = prod.frame_step(ctx, bank, frames[:, 1], masks[:, 1], frames[:, 2], flows[:, 2], commit=False)
        g.replay()
        torch.cuda.synchronize()
    diff = float((out - eager).abs().max())
    assert diff < 1e-3, diff     # same kernels; MIOpen may pick another algorithm under capture


def test_long_memory_bank_and_fp32_kernel(oracle_mod):
    """T = 70 memorised frames (> one wave of the prefix scans), many empty boxes."""
    from rmnet_amd import ops
    rng = np.random.RandomState(70)
    no, T, h, w = 1, 70, 5, 6
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=True)
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr))
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)
    bank = _fill_bank(ops, mk, mv, mr)
    np.testing.assert_allclose(bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)


def test_batchnorm_folding_keeps_the_masks(golden_dir, oracle_mod):
    """fuse_for_inference() (BN folded into the trunk convolutions) == the un-folded network."""
    import copy
    from rmnet_amd.synthetic import synthetic_clip
    prod, _ = _nets(oracle_mod)
    for m in prod.modules():                      # non-trivial statistics, otherwise folding is a no-op
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1)
            m.running_var.uniform_(0.8, 1.25)
    fused = copy.deepcopy(prod).fuse_for_inference()
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in fused.modules())
    frames, masks, flows, n_objects = synthetic_clip(3, 3, 96, 160, seed=9, size=1.3)
    with torch.no_grad():
        a = prod(frames, masks, flows, n_objects, 1)
        b = fused(frames, masks, flows, n_objects, 1)
    assert float((a - b).abs().max()) < 1e-3
    assert (a.argmax(2) == b.argmax(2)).float().mean() > 0.999


@pytest.mark.parametrize('N,C,H,W', [(2, 5, 7, 9), (1, 64, 240, 432), (3, 16, 30, 54), (1, 3, 1, 1)])
def test_channel_affine_is_the_torch_expression(N, C, H, W):
    """rmnet_channel_affine_f32 == act(x*scale + shift + (res*rscale + rshift)) evaluated by torch op
    by op (bit-exact: the kernel is compiled without FMA contraction), every optional operand,
    in place, and the unaligned scalar path."""
    from rmnet_amd import ops
    g = torch.Generator().manual_seed(N * 100 + C)
    x = torch.randn(N, C, H, W, generator=g).to(dev())
    r = torch.randn(N, C, H, W, generator=g).to(dev())
    sc, sh, rs, rh = [torch.randn(C, generator=g).to(dev()) for _ in range(4)]
    v = lambda t: t.view(1, C, 1, 1)
    assert torch.equal(ops.channel_affine(x, sc, sh), x * v(sc) + v(sh))
    assert torch.equal(ops.channel_affine(x, sc, sh, relu=True), torch.relu(x * v(sc) + v(sh)))
    assert torch.equal(ops.channel_affine(x, None, sh, res=r), (x + v(sh)) + r)
    assert torch.equal(ops.channel_affine(x, sc, sh, res=r, res_scale=rs, res_shift=rh, relu=True),
                       torch.relu((x * v(sc) + v(sh)) + (r * v(rs) + v(rh))))
    y = x.clone()
    assert ops.channel_affine(y, sc, None, res=r, res_shift=rh, out=y) is y      # in place on x
    assert torch.equal(y, x * v(sc) + (r + v(rh)))
    y = r.clone()
    ops.channel_affine(x, None, None, res=y, relu=True, out=y)                 # in place on the residual
    assert torch.equal(y, torch.relu(x + r))
    if H * W > 4:                                                              # 4-byte aligned views only
        xs, rs_ = x.flatten()[1:1 + (N * C * H * W - C * H * W)], r.flatten()[1:1 + (N * C * H * W - C * H * W)]
        if N > 1:
            xs, rs_ = xs.view(N - 1, C, H, W), rs_.view(N - 1, C, H, W)
            assert torch.equal(ops.channel_affine(xs, sc, sh, res=rs_, relu=True), torch.relu((xs * v(sc) + v(sh)) + rs_))
    x[0, 0, 0, 0] = float('nan')
    assert torch.isnan(ops.channel_affine(x, sc, sh, relu=True)[0, 0, 0, 0])    # torch.relu keeps NaN
    with pytest.raises(RuntimeError):
        ops.channel_affine(x.cpu(), sc, sh)
    with pytest.raises(RuntimeError):
        ops.channel_affine(x, sc[:-1].contiguous(), sh)


def test_fused_epilogues_match_the_module_graph(oracle_mod):
    """fuse_epilogues(): ResBlocks and Bottleneck trunks within fp32 rounding of the module graph
    (BatchNorm re-expressed as scale/shift, bias added after the convolution), parameters and state
    dict untouched, whole clip within the mask bar."""
    import copy
    from rmnet_amd import networks
    from rmnet_amd.synthetic import synthetic_clip
    prod, _ = _nets(oracle_mod)
    for m in prod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1)
            m.running_var.uniform_(0.8, 1.25)
            m.weight.data.uniform_(0.8, 1.2)
            m.bias.data.uniform_(-0.1, 0.1)
    fused = copy.deepcopy(prod).fuse_epilogues()
    assert list(fused.state_dict().keys()) == list(prod.state_dict().keys())
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for blk_p, blk_f in zip(prod.modules(), fused.modules()):
            if isinstance(blk_p, networks.ResBlock):
                cin = blk_p.conv1.in_channels
                x = torch.randn(2, cin, 12, 20, generator=g).to(dev())
                a, b = blk_p(x), blk_f(x)       # (MIOpen may pick another solver without the bias)
                assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
        x = torch.randn(2, 3, 64, 96, generator=g).to(dev())
        for a, b in zip(prod.encoder_query(x)[:3], fused.encoder_query(x)[:3]):
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())
        m = (torch.rand(2, 64, 96, generator=g) > 0.5).float().to(dev())
        for a, b in zip(prod.encoder_memory(x, m, 1 - m)[:3], fused.encoder_memory(x, m, 1 - m)[:3]):
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())
        frames, masks, flows, n_objects = synthetic_clip(3, 3, 96, 160, seed=11, size=1.3)
        a = prod(frames, masks, flows, n_objects, 1)
        b = fused(frames, masks, flows, n_objects, 1)
    assert float((a - b).abs().max()) < 1e-3
    assert (a.argmax(2) == b.argmax(2)).float().mean() > 0.999
    fused.fuse_epilogues(False)
    with torch.no_grad():   # switched back: the plain module graph again (up to MIOpen's own run-to-run choices)
        assert float((fused(frames, masks, flows, n_objects, 1) - a).abs().max()) < 1e-3


@pytest.mark.parametrize('N,C,H,W', [(2, 8, 5, 6), (1, 64, 30, 54), (3, 4, 7, 9), (2, 256, 12, 20), (1, 12, 6, 5)])
def test_channels_last_glue_kernels_are_the_nchw_kernels(N, C, H, W):
    """[r6] The channels-last entries (rmnet_channel_affine_nhwc_f32, rmnet_upsample2x_add_nhwc_f32, rmnet_affine_relu_maxpool_nhwc_f32) compute
    exactly what the NCHW entries compute -- same expressions, same rounding: bit-identical outputs on the same values in the other layout,
    incl. in-place use, a skip in the other layout, NaN propagation in the max-pool and LeakyReLU."""
    from rmnet_amd import ops
    g = torch.Generator().manual_seed(N * 100 + C)
    x = torch.randn(N, C, H, W, generator=g).to(dev())
    res = torch.randn(N, C, H, W, generator=g).to(dev())
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(dev()), torch.randn(C, generator=g).to(dev())
    rs, rh = (torch.rand(C, generator=g) + 0.5).to(dev()), torch.randn(C, generator=g).to(dev())
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for kw in (dict(relu=True), dict(relu=False), dict(relu='leaky'), dict(res=res, relu=True), dict(res=res, res_scale=rs, res_shift=rh, relu=True)):
        want = ops.channel_affine(x, sc, sh, **kw)
        kw_cl = dict(kw, res=cl(kw['res'])) if 'res' in kw else kw
        got = ops.channel_affine(cl(x), sc, sh, **kw_cl)
        assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want), kw.keys()
        if 'res' in kw:                                       # a skip in the OTHER layout is converted, not misread
            assert torch.equal(ops.channel_affine(cl(x), sc, sh, **kw), want)
    t = cl(x.clone())
    assert ops.channel_affine(t, None, sh, relu=True, out=t) is t and torch.equal(t, ops.channel_affine(x, None, sh, relu=True))     # in place, no scale
    skip = torch.randn(N, C, 2 * H, 2 * W, generator=g).to(dev())
    want = ops.upsample2x_add(x, skip)
    got = ops.upsample2x_add(cl(x), cl(skip))
    assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want)
    assert torch.equal(ops.upsample2x_add(cl(x)), ops.upsample2x_add(x))
    xn = x.clone()
    xn[0, 1, 2, 3] = float('nan')
    want = ops.affine_relu_maxpool(xn, sc, sh)
    got = ops.affine_relu_maxpool(cl(xn), sc, sh)
    assert got.is_contiguous(memory_format=torch.channels_last) and got.shape == want.shape
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))


def test_channels_last_loop_meets_the_bar(oracle_mod):
    """[r6] The frame loop with both networks in channels_last memory format (MIOpen's NHWC kernels, no layout transposes) and the fused glue
    kernels ON -- bench.py --channels-last -- against the CPU path on a live-boundary clip: the same bars as the NCHW loop
    (test_live_boundary_clips_meet_the_bar_in_every_arithmetic), and within fp32 convolution rounding of the NCHW loop itself."""
    prod, ref = _nets(oracle_mod, 'auto')
    name = 'live480-a'
    frames, masks, flows, n_objects, every, delta = lf.make_clip(name)
    lf.shift_foreground_bias(prod, delta)
    lf.shift_foreground_bias(ref, delta)
    prod.fuse_epilogues()
    est_cpu, log_cpu = _cpu_path(oracle_mod, ref, name, frames, masks, flows, n_objects, every)
    with torch.no_grad():
        est_n, log_n = prod(frames, masks, flows, n_objects, every, return_logits=True)
        prod = prod.to(memory_format=torch.channels_last)
        assert prod.encoder_query.conv1.weight.is_contiguous(memory_format=torch.channels_last)
        est, logits = prod(frames, masks, flows, n_objects, every, return_logits=True)
    est, logits = est.cpu(), logits.cpu()
    iou, gap = lf.label_iou(est, est_cpu), lf.logit_gap(logits, log_cpu)
    _table('%s channels_last auto: IoU %.5f  max live fg-logit diff %.2e; vs the NCHW loop: max prob diff %.2e' % (
        name, iou, gap, float((est - est_n.cpu()).abs().max())))
    assert iou >= 0.999 and gap <= LIVE_LOGIT_BAR, (iou, gap)
    assert lf.label_iou(est, est_n.cpu()) >= 0.9995


@pytest.mark.parametrize('N,C,h,w', [(2, 3, 5, 6), (1, 8, 30, 54), (1, 2, 7, 9), (1, 1, 1, 1), (4, 16, 60, 108)])
def test_upsample2x_add_is_torch_bilinear(N, C, h, w):
    """rmnet_upsample2x_add_f32 == skip + F.interpolate(x, scale_factor=2, 'bilinear',
    align_corners=False): same source-index rule and the same expression; the weights are exactly
    0, 0.25, 0.75 or 1, so only FMA contraction inside torch's own kernel can differ (<= 1 ulp of the
    largest term)."""
    import torch.nn.functional as F
    from rmnet_amd import ops
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(N, C, h, w, generator=g).to(dev())
    s = torch.randn(N, C, 2 * h, 2 * w, generator=g).to(dev())
    up = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    got = ops.upsample2x_add(x)
    assert float((got - up).abs().max()) <= 4e-7 * max(1.0, float(x.abs().max()))
    got = ops.upsample2x_add(x, s)
    assert float((got - (s + up)).abs().max()) <= 6e-7 * max(1.0, float(x.abs().max()) + float(s.abs().max()))
    t = s.clone()
    assert ops.upsample2x_add(x, t, out=t) is t and torch.equal(t, got)        # in place on the skip
    with pytest.raises(RuntimeError):
        ops.upsample2x_add(x, s[:, :, :-1].contiguous())


@pytest.mark.parametrize('counts,K,Hp,Wp,pad', [([1], 2, 32, 48, (0, 0, 0, 0)), ([2, 1, 3], 5, 48, 64, (5, 5, 3, 2)),
                                                 ([1, 1, 1, 1], 2, 480, 864, (5, 5, 0, 0)), ([0, 2], 4, 16, 16, (1, 0, 0, 1))])
def test_soft_aggregate_matches_the_module_graph(counts, K, Hp, Wp, pad, oracle_mod):
    """rmnet_soft_aggregate_f32 == softmax(dec)[:, 1] -> RMNet.soft_aggregation -> un-pad (-> softmax)
    evaluated by torch (models/rmnet.py:368-380, 289-302, 450), incl. clips with no object and
    absent channels."""
    import torch.nn.functional as F
    from rmnet_amd import ops
    from rmnet_amd.rmnet import RMNet
    g = torch.Generator().manual_seed(sum(counts) * 10 + K)
    n = sum(counts)
    dec = (torch.randn(max(n, 1), 2, Hp, Wp, generator=g) * 4).to(dev())[:n].contiguous()
    begin = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev())
    lw, uw, lh, uh = pad
    net = RMNet(None)
    if n:
        ps = F.softmax(dec, dim=1)[:, 1]
    else:
        ps = dec.new_zeros(0, Hp, Wp)
    want = net.soft_aggregation(ps, K, counts)[:, :, lh:Hp - uh, lw:Wp - uw]
    logit, prob = ops.soft_aggregate(dec if n else dec.new_zeros(1, 2, Hp, Wp), begin, K, pad, want_prob=True)
    assert logit.shape == want.shape
    np.testing.assert_allclose(logit.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(prob.cpu().numpy(), F.softmax(want, dim=1).cpu().numpy(), rtol=2e-5, atol=1e-6)
    only, none = ops.soft_aggregate(dec if n else dec.new_zeros(1, 2, Hp, Wp), begin, K, pad)
    assert none is None and torch.equal(only, logit)


@pytest.mark.parametrize('B,K,H,W,amp', [(1, 2, 480, 854, 3.0), (2, 3, 97, 131, 8.0), (1, 4, 33, 47, 40.0), (1, 2, 1, 1, 0.3)])
def test_region_map_of_the_warped_mask_is_bit_exact(B, K, H, W, amp):
    """rmnet_region_map_warped_f32 == att_map_generator(RMNet.warp(mask, flow)) (models/rmnet.py:252-287):
    the warped mask it evaluates in registers is bit-identical to the one torch computes on the same
    GPU (grid build, two grid_samples, validity threshold), hence so are the integer boxes and cell
    rectangles -- including flows that leave the frame."""
    from rmnet_amd import ops
    from rmnet_amd.rmnet import RMNet
    net = RMNet(None)
    g = torch.Generator().manual_seed(B * 1000 + H)
    m = torch.rand(B, K, H, W, generator=g).to(dev())
    f = (torch.randn(B, 2, H, W, generator=g) * amp).to(dev())
    f[:, :, : max(1, H // 8)] += 2.0 * max(H, W)              # a band sampled far outside: validity 0
    want = net.warp(m, f)[0].contiguous()
    grid = (3, 1, 16, (H + 15) // 16 + 1, (W + 15) // 16 + 1)
    att0, bb0, rc0 = ops.region_map(want, cell_grid=grid)
    att1, bb1, rc1, got = ops.region_map(m, cell_grid=grid, flow=f, want_warped=True)
    assert torch.equal(got[:, 1:], want[:, 1:]) and float(got[:, 0].abs().max()) == 0.0
    assert torch.equal(bb0, bb1) and torch.equal(rc0, rc1) and torch.equal(att0, att1)
    _, bb2, _ = ops.region_map(m, want_map=False, flow=f)                      # without the extra output
    assert torch.equal(bb2, bb0)
    with pytest.raises(RuntimeError):
        ops.region_map(m, flow=f[:, :1].contiguous())


def test_inference_harness_single_rank(oracle_mod):
    """rmnet_amd.inference.segment_video / segment_videos (core/inference.py:49-63 without the file
    output): flows from TinyFlowNet, device-resident frame loop, argmax labels -- equal to running the
    two networks by hand; without a process group the 'gather' is the identity."""
    from rmnet_amd import inference, networks
    from rmnet_amd.synthetic import synthetic_clip
    from rmnet_amd.tiny_flownet import TinyFlowNet
    prod, _ = _nets(oracle_mod)
    tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev()).eval()
    videos = []
    for seed, n in ((3, 3), (4, 4)):
        frames, masks, _, n_objects = synthetic_clip(n, 3, 96, 160, seed=seed, size=1.3)
        videos.append({'frames': frames[0], 'masks': masks[0], 'n_objects': int(n_objects[0, 0])})
    with torch.no_grad():
        got = inference.segment_videos(videos, lambda v: inference.segment_video(prod, tfn, v, memorize_every=2))
        for i, v in enumerate(videos):
            fr = v['frames'].unsqueeze(0).to(dev())
            n_obj = torch.full((1, fr.shape[1]), v['n_objects'], dtype=torch.long)
            est = prod(fr, v['masks'].unsqueeze(0), tfn(fr), n_obj, 2)
            want = est[0].argmax(1).to(torch.uint8)
            assert got[i].dtype == torch.uint8 and got[i].shape == want.shape
            assert (got[i] == want).float().mean() > 0.999
    assert inference.video_costs(videos) == [3 * videos[0]['n_objects'], 4 * videos[1]['n_objects']]


@pytest.mark.parametrize('N,C,H,W', [(2, 3, 7, 9), (1, 64, 240, 432), (1, 2, 1, 1), (3, 5, 16, 10), (2, 4, 9, 16)])
def test_affine_relu_maxpool_is_the_torch_expression(N, C, H, W):
    """rmnet_affine_relu_maxpool_f32 == F.max_pool2d(relu(x*scale + shift), 3, stride 2, padding 1),
    bit for bit (odd sizes, borders, NaN propagation)."""
    import torch.nn.functional as F
    from rmnet_amd import ops
    g = torch.Generator().manual_seed(H * 10 + W)
    x = torch.randn(N, C, H, W, generator=g).to(dev())
    sc, sh = torch.randn(C, generator=g).to(dev()), torch.randn(C, generator=g).to(dev())
    v = lambda t: t.view(1, C, 1, 1)
    want = F.max_pool2d(torch.relu(x * v(sc) + v(sh)), 3, stride=2, padding=1)
    got = ops.affine_relu_maxpool(x, sc, sh)
    assert got.shape == want.shape and torch.equal(got, want)
    assert torch.equal(ops.affine_relu_maxpool(x), F.max_pool2d(torch.relu(x), 3, stride=2, padding=1))
    x[0, 0, 0, 0] = float('nan')
    got = ops.affine_relu_maxpool(x, sc, sh)
    assert torch.isnan(got[0, 0, 0, 0]) and not torch.isnan(got[0, -1]).any()


def test_tiny_flownet_fused_bias_leaky(golden_dir):
    """TinyFlowNet.fuse_epilogues(): bias + LeakyReLU(0.1) in one kernel per block == the module graph
    (channel_affine act 2 is bit-exact on its own: x + bias, then x > 0 ? x : 0.1 x)."""
    import copy
    from rmnet_amd import networks, ops
    from rmnet_amd.tiny_flownet import TinyFlowNet
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 6, 9, 11, generator=g).to(dev())
    b = torch.randn(6, generator=g).to(dev())
    want = torch.nn.functional.leaky_relu(x + b.view(1, 6, 1, 1), 0.1)
    assert torch.equal(ops.channel_affine(x, None, b, relu='leaky'), want)
    tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev()).eval()
    fused = copy.deepcopy(tfn).fuse_epilogues()
    frames = torch.randn(1, 3, 3, 96, 160, generator=g).to(dev())
    with torch.no_grad():
        a, f = tfn(frames), fused(frames)
    assert float((a - f).abs().max()) <= 1e-4 * max(1.0, float(a.abs().max()))


# ----------------------------------------------------------------------------- round 2: pins and ranges
def test_region_map_boxes_match_the_reference_box_finder(golden_dir, oracle_mod):
    """G1 pinned by the REFERENCE: utils/helpers.py:93-102 get_bounding_boxes on 24 seeded soft masks
    (K = 11, empty channels, values at exactly 0.5; tests/golden/cases.py).  With n_pts_threshold = 1 and
    n_bbox_loose_pixels = 0 the CUDA kernel reduces to that function (reg_att_map_generator.cu:63-74)."""
    import sys
    sys.path.insert(0, golden_dir)
    import cases
    from rmnet_amd import ops
    g = np.load(os.path.join(golden_dir, 'region_boxes.npz'))
    assert len(cases.REGION_BOX_SHAPES) >= 20
    for i, (B, K, H, W) in enumerate(cases.REGION_BOX_SHAPES):
        m = cases.region_box_case(i)
        assert float(m.astype(np.float64).sum()) == float(g['case%02d.checksum' % i])
        want = cases.boxes_from_reference_tight(g['case%02d.tight' % i], K, H, W).reshape(B, K, 4)
        att, bb, _ = ops.region_map(cu(m), 0.5, 1, 0)
        assert np.array_equal(bb.cpu().numpy(), want), i
        o_att, o_bb = oracle_mod.region_map(m, 0.5, 1, 0)
        assert np.array_equal(o_bb, want) and np.array_equal(att.cpu().numpy(), o_att), i


def test_bank_reads_the_reference_golden_vectors(golden_dir):
    """The reference-generated MemoryReader vectors straight through the split-fp16 MemoryBank (the
    dominant kernel bk_main), incl. `peaky` (logit scale 3: the running reference is bumped).  `tiny_p`
    (De = 16 / Do = 32) is not a bank shape."""
    from rmnet_amd import ops
    g = np.load(os.path.join(golden_dir, 'memory_reader.npz'))
    for name in ['dense', 'regional', 'allmasked_query', 'peaky']:
        mk, mv = g[name + '.m_key'], g[name + '.m_val']
        mr = g[name + '.mem_rects'] if name + '.mem_rects' in g.files else None
        qr = g[name + '.qry_rects'] if name + '.qry_rects' in g.files else None
        bank = _fill_bank(ops, mk, mv, mr, capacity=mk.shape[2] + 1)
        got = bank.read(mk.shape[2], cu(g[name + '.q_key']), cu(g[name + '.q_val']), None if qr is None else cu(qr))
        np.testing.assert_allclose(got.cpu().numpy(), g[name + '.mem_val'], atol=MR_ATOL, rtol=MR_RTOL, err_msg=name)
        assert bank.overflow_count() == 0


@pytest.mark.parametrize('kq_scale,v_scale', [(1e-3, 1e-3), (1e-2, 1e-2), (1.0, 1.0), (1.0, 1e2), (3.0, 1e2)])
def test_bank_is_fp32_class_across_input_scales(kq_scale, v_scale, oracle_mod):
    """Split-fp16 range claim.  Keys / queries times kq_scale, values times v_scale: 1e-3 is where the lo
    plane would be subnormal without the 2^6 storage scale (absolute error floor 3e-8 -> 4.7e-10), 1e2
    puts values up to ~500 (window: 1023.5), kq 3 gives logits of +-30.  (Larger key scales are not a
    precision test any more: at |S| ~ 1e4 the fp32 reference itself resolves S to ~1e-2 only.)
    Tolerance relative to the value scale, same constants as everywhere else."""
    from rmnet_amd import ops
    rng = np.random.RandomState(31)
    mk, mv, qk, qv, mr, qr = _random_case(rng, 2, 3, 9, 13, regional=True)
    mk, qk = (mk * kq_scale).astype(np.float32), (qk * kq_scale).astype(np.float32)
    mv, qv = (mv * v_scale).astype(np.float32), (qv * v_scale).astype(np.float32)
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    bank = _fill_bank(ops, mk, mv, mr)
    got = bank.read(3, cu(qk), cu(qv), cu(qr)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=MR_ATOL * v_scale, rtol=MR_RTOL)
    assert bank.overflow_count() == 0
    via, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr))
    np.testing.assert_allclose(via.cpu().numpy(), want, atol=MR_ATOL * v_scale, rtol=MR_RTOL)
    if v_scale < 1:      # the small-value regime, measured against the value scale: fp32-class means ~1e-7
        assert float(np.abs(got - want).max()) < 2e-6 * v_scale


def test_out_of_window_values_are_detected_not_silently_clamped(oracle_mod):
    """One value of 7e4 (beyond fp16) and one NaN-free 2e3 key: the bank COUNTS them
    (rmnet_bank_overflow_offset), the drop-in entry falls back to the exact-fp32 kernel on the device and
    still matches the oracle, and the tensor-backed bank (what the frame loop switches to) is exact too."""
    from rmnet_amd import ops
    rng = np.random.RandomState(32)
    mk, mv, qk, qv, mr, qr = _random_case(rng, 2, 3, 9, 13, regional=True)
    mr[:] = (0, 12, 0, 8)
    qr[:] = (1, 11, 0, 8)
    mv[1, 7, 2, 4, 5] = 7e4
    mv[0, 100, 0, 0, 0] = -2e3
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    bank = _fill_bank(ops, mk, mv, mr)
    assert bank.overflow_count() == 2
    via, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr))          # device-side fallback
    np.testing.assert_allclose(via.cpu().numpy(), want, atol=MR_ATOL, rtol=5e-5)
    tb = ops.TensorBank(2, 4, 9, 13, dev())
    for t in range(3):
        tb.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), cu(mr[:, t]))
    np.testing.assert_allclose(tb.read(3, cu(qk), cu(qv), cu(qr)).cpu().numpy(), want, atol=MR_ATOL, rtol=5e-5)
    # and a clean input right after, through the same workspace-free entry: the fast path again
    mv[1, 7, 2, 4, 5] = 0.5
    mv[0, 100, 0, 0, 0] = 0.25
    want2, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    via2, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr))
    np.testing.assert_allclose(via2.cpu().numpy(), want2, atol=MR_ATOL, rtol=MR_RTOL)


def test_frame_loop_redoes_a_clip_exactly_when_the_bank_overflows(oracle_mod):
    """RMNet.forward checks the bank's overflow word once per clip and re-runs the clip on fp32 tensors +
    the exact kernel; forced here by scaling the value head so that V leaves the fp16 window."""
    from rmnet_amd.synthetic import synthetic_clip
    prod, ref = _nets(oracle_mod)
    frames, masks, flows, n_objects = synthetic_clip(3, 2, 96, 160, seed=4, size=1.4)
    with torch.no_grad():
        _, v4, _, _ = prod._encode_memory(frames[:, 0].to(dev()), masks[:, 0].to(dev()).float(), [1])
        gain = 3000.0 / float(v4.abs().max())            # largest memorised value ~3000 > 1023.5
        for net in (prod, ref):
            net.kv_memory.value_conv.weight.mul_(gain)
            net.kv_memory.value_conv.bias.mul_(gain)
    calls = []
    orig = prod.new_bank
    prod.new_bank = lambda ctx, cap, exact=False, precision=None: (calls.append(exact), orig(ctx, cap, exact, precision))[1]
    with torch.no_grad():
        est = prod(frames, masks, flows, n_objects, 1).cpu()
        est_cpu = ref(frames, masks, flows, n_objects, 1)
    assert calls == [False, True]
    assert float((est - est_cpu).abs().max()) < 5e-3          # (the decoder sees inputs `gain` times larger)
    assert (est.argmax(2) == est_cpu.argmax(2)).float().mean() > 0.995


@pytest.mark.parametrize('H,W,K', [(480, 854, 2), (720, 1280, 3)])
def test_whole_loop_at_baseline_resolutions(H, W, K, oracle_mod):
    """BASELINE configs[1] / configs[4] frame sizes: two frames of the device-resident loop (fused and
    un-fused) against the CPU path -- the 30x54 / 45x80 grids, the 5-pixel pad of 854 and the fused warp
    at full resolution."""
    from rmnet_amd.synthetic import synthetic_clip
    prod, ref = _nets(oracle_mod)
    frames, masks, flows, n_objects = synthetic_clip(3, K, H, W, seed=H, size=1.6)
    with torch.no_grad():
        est_cpu = ref(frames, masks, flows, n_objects, 1)
        est = prod(frames, masks, flows, n_objects, 1).cpu()
        prod.fuse_epilogues()
        est_f = prod(frames, masks, flows, n_objects, 1).cpu()
    for e in (est, est_f):
        assert float((e - est_cpu).abs().max()) < 1e-3
        lab, lab_cpu = e.argmax(2).numpy(), est_cpu.argmax(2).numpy()
        for k in range(1, K):
            assert oracle_mod.iou(lab[:, 1:] == k, lab_cpu[:, 1:] == k) >= 0.999


def test_multi_scale_inference_matches_the_reference(golden_dir):
    """H1: rmnet_amd.helpers.multi_scale_inference (utils/helpers.py:44-78) against outputs of the
    reference's own function on the same clip and weights (FRAME_SCALES [1.0] and [0.75, 1.0] + FLIP_LR),
    host tensors in as from the reference's loader; var_or_cuda (utils/helpers.py:16-24)."""
    import sys
    from types import SimpleNamespace
    sys.path.insert(0, golden_dir)
    import cases
    from rmnet_amd import helpers, networks
    from rmnet_amd.rmnet import RMNet
    from rmnet_amd.synthetic import synthetic_clip
    from rmnet_amd.tiny_flownet import TinyFlowNet
    g = np.load(os.path.join(golden_dir, 'multi_scale_inference.npz'))
    net = networks.procedural_init_(RMNet(None)).to(dev()).eval()
    tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev()).eval()
    c = cases.MSI_CLIP
    frames, masks, _, n_objects = synthetic_clip(c['N'], c['K'], c['H'], c['W'], seed=c['seed'], size=c['size'])
    for name, scales, flip in cases.MSI_CASES:
        cfg = SimpleNamespace(TEST=SimpleNamespace(FRAME_SCALES=scales, FLIP_LR=flip, MEMORIZE_EVERY=c['memorize_every']))
        with torch.no_grad():
            flows, probs = helpers.multi_scale_inference(cfg, tfn, net, frames, masks, n_objects)
        assert flows.shape == (1, c['N'], 2, c['H'], c['W']) and probs.shape == (1, c['N'], c['K'], c['H'], c['W'])
        np.testing.assert_allclose(probs.cpu().numpy(), g[name + '.probs'].astype(np.float32), atol=3e-3, err_msg=name)
        np.testing.assert_allclose(flows.cpu().numpy(), g[name + '.flows'].astype(np.float32), atol=5e-3, rtol=2e-3, err_msg=name)
        assert (probs.argmax(2).cpu().numpy() == g[name + '.argmax']).mean() > 0.999
    x = torch.arange(24.).view(2, 3, 4).transpose(1, 2)
    y = helpers.var_or_cuda(x)
    assert y.is_cuda and y.is_contiguous() and torch.equal(y.cpu(), x)
    z = helpers.var_or_cuda(x, torch.device('cpu'))
    assert not z.is_cuda and z.is_contiguous() and bool(g['var_or_cuda.contiguous'].all())
    assert helpers.var_or_cuda(x, torch.device('cuda', 0)).device == torch.device('cuda', 0)


def test_device_side_jaccard_on_the_gpu(oracle_mod):
    """rmnet_amd.metrics on CUDA tensors (what evaluate_videos runs) vs the oracle's scalar IoU."""
    from rmnet_amd import metrics
    rng = np.random.RandomState(6)
    N, H, W, n = 5, 33, 47, 3
    pred, gt = rng.randint(0, n + 1, size=(N, H, W)), rng.randint(0, n + 1, size=(N, H, W))
    pred[3][pred[3] == 1] = 0
    gt[3][gt[3] == 1] = 0
    j = metrics.jaccard_per_object(cu(pred), cu(gt), n)
    assert j.is_cuda
    j = j.cpu().numpy()
    for t in range(N):
        for o in range(1, n + 1):
            assert abs(j[t, o - 1] - oracle_mod.iou(pred[t] == o, gt[t] == o)) < 1e-6
    assert j[3, 0] == 1.0
    assert abs(float(metrics.mean_jaccard(cu(pred), cu(gt), n)) - j[1:-1].mean()) < 1e-6


def test_inference_only_guard_and_fuse_order(oracle_mod):
    from rmnet_amd.synthetic import synthetic_clip
    prod, _ = _nets(oracle_mod)
    frames, masks, flows, n_objects = synthetic_clip(2, 2, 64, 96, seed=1)
    prod.train()
    with pytest.raises(RuntimeError, match='inference-only'):
        prod(frames, masks, flows, n_objects, 1)
    prod.eval()
    est = prod(frames, masks, flows, n_objects, 1)          # grad mode on: runs under no_grad inside
    assert not est.requires_grad
    prod.fuse_epilogues()
    sd = {k: v.clone() for k, v in prod.state_dict().items()}
    for k in sd:
        if k.endswith('running_var'):
            sd[k] = sd[k] * 1.5
    before = prod.encoder_query.res2[0]._s1.clone()
    prod.load_state_dict(sd)                                 # the fused scale/shift snapshots are refreshed
    assert not torch.equal(before, prod.encoder_query.res2[0]._s1)
    prod.fuse_epilogues(False)
    prod.fuse_for_inference()
    with pytest.raises(RuntimeError, match='already folded'):
        prod.fuse_epilogues()


@pytest.mark.parametrize('ranks', [2, 4, 8])
def test_bench_over_rccl(ranks):
    """bench.py --gpus N with the nccl (= RCCL) backend on as many GPUs as the box has (2 / 4 / 8): the weak-scaling line,
    every rank's own rate and the label-map gather over xGMI."""
    import subprocess
    import sys
    if torch.cuda.device_count() < ranks:
        pytest.skip('needs >= %d GPUs (the test box has %d); the N > 1 path is covered by the gloo tests' % (ranks, torch.cuda.device_count()))
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')      # dmabuf IPC (the host driver has no legacy IPC)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks),
                          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'),
                          '--gpus', str(ranks), '--steps', '2', '--warmup', '1', '--clips-per-gpu', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == ranks and line['value'] > 0
    assert len(line['multi_gpu']['per_rank_fps']) == ranks and line['multi_gpu']['backend'] == 'nccl' and line['multi_gpu']['gather_ms'] > 0


_RCCL_ONE_RANK = r"""
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as tdist
from rmnet_amd import dist as rd, inference, networks
from rmnet_amd.rmnet import RMNet
from rmnet_amd.tiny_flownet import TinyFlowNet
from rmnet_amd.synthetic import synthetic_clip
rank, world, local = rd.init_from_env('nccl', force_group=True)
assert tdist.is_initialized() and tdist.get_backend() == 'nccl' and (rank, world) == (0, 1)
dev = torch.device('cuda', 0)
torch.set_grad_enabled(False)
# device-resident header, device payloads, dist.gather with a device gather list
maps = {0: torch.randint(0, 3, (3, 40, 56), dtype=torch.uint8, device=dev), 2: torch.randint(0, 5, (2, 24, 32), dtype=torch.uint8, device=dev)}
got = rd.gather_label_maps(maps, 3)
assert sorted(got) == [0, 2] and all(got[v].is_cuda and torch.equal(got[v], maps[v]) for v in maps)
assert rd.max_over_ranks(1.25) == 1.25 and rd.sum_over_ranks(2.5) == 2.5
rd.barrier()
# the sharded runner through the same group: label maps gathered, J all-reduced
net = networks.procedural_init_(RMNet(None)).to(dev).eval()
tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev).eval()
videos = []
for i, (n, k) in enumerate([(4, 2), (3, 3)]):
    fr, ms, _, _ = synthetic_clip(n, k + 1, 64, 96, seed=20 + i)
    videos.append({'frames': fr[0], 'masks': ms[0], 'n_objects': k, 'labels': ms[0].argmax(dim=1).to(torch.uint8)})
seg = lambda v: inference.segment_video(net, tfn, v, memorize_every=2)
out = inference.segment_videos(videos, seg)
assert sorted(out) == [0, 1] and tuple(out[0].shape) == (4, 64, 96) and out[0].is_cuda
j = inference.evaluate_videos(videos, seg)
assert 0.0 <= j <= 1.0
tdist.destroy_process_group()
print(json.dumps({'ok': True, 'J': j}))
"""


def test_rccl_branch_on_a_one_rank_group():
    """The `nccl` (= RCCL) branch of rmnet_amd.dist -- device-resident header all_gather, dist.gather of device payloads into a device
    gather list, the float all-reduces, the sharded runner -- executed on the hardware that exists: a ONE-rank RCCL process group on
    cuda:0 (round-5 verdict: the branch had never run).  Then bench.py itself through that backend (`--gpus 1 --dist-backend nccl`):
    its line must carry the multi_gpu block with a gather that went through RCCL."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    out = subprocess.run([sys.executable, '-c', _RCCL_ONE_RANK], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])['ok']      # (RCCL prints its library path at exit)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--dist-backend', 'nccl', '--steps', '2', '--warmup', '1',
                          '--clips-per-gpu', '2', '--no-cpu-baseline', '--no-extras', '--no-miopen-find'],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0
    mg = line['multi_gpu']
    assert mg['backend'] == 'nccl' and mg['gather_error'] is None and mg['gather_ms'] > 0 and len(mg['per_rank_fps']) == 1


def test_sharded_runner_on_a_720p_multi_object_clip(oracle_mod):
    """BASELINE configs[3] shape on one rank: a 720x1280, 3-object clip through
    rmnet_amd.inference.segment_videos / evaluate_videos (flows from TinyFlowNet, frame loop, labels, J on
    the device) == running the two networks by hand."""
    from rmnet_amd import inference, metrics, networks
    from rmnet_amd.synthetic import synthetic_clip
    from rmnet_amd.tiny_flownet import TinyFlowNet
    prod, _ = _nets(oracle_mod)
    prod.fuse_epilogues()
    tfn = networks.procedural_init_(TinyFlowNet(None)).to(dev()).eval().fuse_epilogues()
    frames, masks, _, n_objects = synthetic_clip(4, 4, 720, 1280, seed=72, size=1.5)
    video = {'frames': frames[0], 'masks': masks[0], 'n_objects': 3, 'labels': masks[0].argmax(0).to(torch.uint8)}
    video['labels'] = masks[0].argmax(dim=1).to(torch.uint8)
    seg = lambda v: inference.segment_video(prod, tfn, v, memorize_every=2)
    with torch.no_grad():
        got = inference.segment_videos([video], seg)
        fr = frames.to(dev())
        want = prod(fr, masks, tfn(fr), n_objects, 2)[0].argmax(1).to(torch.uint8)
        j = inference.evaluate_videos([video], seg)
    assert sorted(got) == [0] and got[0].shape == (4, 720, 1280) and got[0].is_cuda
    assert (got[0] == want).float().mean() > 0.999
    jj = metrics.jaccard_per_object(want.long(), video['labels'].to(dev()).long(), 3)[1:-1]
    assert abs(j - float(jj.mean())) < 1e-3


@pytest.mark.parametrize('no,T,h,w,reads', [(2, 3, 6, 10, 60), (5, 3, 9, 13, 60), (8, 5, 30, 54, 12)])
def test_bank_read_is_repeatable(no, T, h, w, reads, oracle_mod):
    """The same bank read many times: every read must equal the oracle.  A hazard-timing bug (an inline-asm
    v_fma_mix reading a v_exp_f32 result one state too early) once made one producer wave publish a wrong
    lo plane in a few launches out of a hundred -- a single passing read proves nothing about those
    (tests/stress_race.py is the long version)."""
    from rmnet_amd import ops
    rng = np.random.RandomState(no * 7 + T)
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=True)
    mr[:, T - 1] = (0, w - 1, 0, h - 1)
    qr[:] = (0, w - 1, 0, h - 1)
    mk[:, :, T - 1, h - 1, w - 1] = qk[:, :, 2, 3] * 9.0
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    bank = _fill_bank(ops, mk, mv, mr, capacity=T + 1)
    qk_d, qv_d, qr_d = cu(qk), cu(qv), cu(qr)
    bad = 0
    for _ in range(reads):
        got = bank.read(T, qk_d, qv_d, qr_d).cpu().numpy()
        bad += not np.allclose(got, want, atol=MR_ATOL, rtol=MR_RTOL)
    assert bad == 0, '%d of %d reads differ from the oracle' % (bad, reads)


def test_fused_warp_self_check_and_fallback(oracle_mod):
    """fuse_epilogues() arms a one-off bit-exactness check of the warp-fused box kernel against this build's
    torch warp; when it fails the frame loop uses region_map(warp(...)) (forced here by patching warp)."""
    from rmnet_amd.synthetic import synthetic_clip
    prod, _ = _nets(oracle_mod)
    prod.fuse_epilogues()
    frames, masks, flows, n_objects = synthetic_clip(3, 2, 96, 160, seed=2, size=1.4)
    with torch.no_grad():
        a = prod(frames, masks, flows, n_objects, 1)
    assert prod._fused_warp is True
    prod.fuse_epilogues()                      # re-arm
    orig = prod.warp
    prod.warp = lambda img, flow: tuple(t + (1e-3 if i == 0 else 0) for i, t in enumerate(orig(img, flow)))
    with pytest.warns(UserWarning, match='not bit-identical'):
        assert prod._fused_warp_ok(dev()) is False
    prod.warp = orig
    with torch.no_grad():
        b = prod(frames, masks, flows, n_objects, 1)      # un-fused warp path, same boxes
    assert float((a - b).abs().max()) < 1e-5


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def test_bench_two_ranks_sharing_one_gpu_over_gloo():
    """The N > 1 path of bench.py end to end on a 1-GPU box: two ranks (gloo) share cuda:0, each runs its own
    clips, the step time is the max over ranks and rank 0 prints the one JSON line with the whole-job rate."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(root, 'bench.py'),
                          '--gpus', '2', '--steps', '2', '--warmup', '1', '--clips-per-gpu', '1', '--dist-backend', 'gloo',
                          '--no-miopen-find'],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['scaling'] == 'weak'
    assert abs(line['value'] - 2 * 1 * 2 / (line['ms_per_step'] * 2 / 1e3)) < 0.05 * line['value']   # N * clips * K / time
    assert 'cpu_baseline' not in line and 'extras' not in line
    mg = line['multi_gpu']                                   # every rank's own rate + the gather of the label maps (GPU-resident payloads)
    assert len(mg['per_rank_fps']) == 2 and min(mg['per_rank_fps']) >= 0.5 * line['value'] / 2 * 0.9
    assert mg['gather_ms'] > 0 and mg['gather_bytes_per_rank'] == 480 * 854 and mg['backend'] == 'gloo'


def _sharded_720p_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rmnet_amd import inference, networks
    from rmnet_amd.rmnet import RMNet
    from rmnet_amd.synthetic import synthetic_clip
    from rmnet_amd.tiny_flownet import TinyFlowNet
    d = torch.device('cuda', 0)
    torch.cuda.set_device(d)
    net = networks.procedural_init_(RMNet(None)).to(d).eval().fuse_epilogues()
    tfn = networks.procedural_init_(TinyFlowNet(None)).to(d).eval().fuse_epilogues()
    videos = []
    for seed, (n, k, h, w) in enumerate([(3, 4, 720, 1280), (4, 2, 480, 854), (3, 3, 480, 854)]):
        frames, masks, _, _ = synthetic_clip(n, k, h, w, seed=40 + seed, size=1.5)
        videos.append({'frames': frames[0], 'masks': masks[0], 'n_objects': k - 1,
                       'labels': masks[0].argmax(dim=1).to(torch.uint8)})
    seen = []

    def seg(v):
        seen.append(tuple(v['frames'].shape))
        return inference.segment_video(net, tfn, v, memorize_every=2)
    with torch.no_grad():
        maps = inference.segment_videos(videos, seg)
        j = inference.evaluate_videos(videos, seg)
    q.put((rank, len(seen) // 2, {k: v.cpu().numpy() for k, v in maps.items()}, j))   # (by value: the sender may exit first)
    dist.destroy_process_group()


def test_sharded_runner_two_ranks_on_real_kernels():
    """BASELINE configs[3] in miniature: YouTube-VOS-shaped clips (one 720p / 3 objects, two 480p) sharded over two
    gloo ranks that share the GPU; every clip runs on exactly one rank through the real frame loop, rank 0 receives all
    label maps, and the all-reduced J is the same number on both ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_720p_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, n0, maps0, j0), (_, n1, maps1, j1) = out
    assert n0 + n1 == 3 and n0 >= 1 and n1 >= 1               # (the 720p 3-object clip alone outweighs the other two)
    assert sorted(maps0) == [0, 1, 2] and maps1 == {}
    assert maps0[0].shape == (3, 720, 1280) and maps0[1].shape == (4, 480, 854) and maps0[0].dtype == np.uint8
    assert abs(j0 - j1) < 1e-12 and 0.0 <= j0 <= 1.0


def test_bank_error_is_in_the_exact_fp32_kernels_class(oracle_mod):
    """'fp32-class' made concrete: against the double-accumulating oracle, the split-fp16 bank read's largest
    error stays within 2x the exact-fp32 MFMA kernel's largest error (+1e-7) on the same inputs -- over several
    seeds, regional and dense, including a peaky soft-max (key scale 3)."""
    from rmnet_amd import ops
    worst = []
    for seed, (no, T, h, w, regional, kq) in enumerate([(2, 3, 12, 20, True, 0.6), (1, 5, 30, 54, True, 0.6),
                                                        (1, 2, 16, 24, False, 0.6), (2, 3, 12, 20, True, 3.0),
                                                        (3, 4, 9, 13, True, 1.5)]):
        rng = np.random.RandomState(900 + seed)
        mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, scale=kq, regional=regional)
        if regional:
            want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
            exact, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr), flags=ops.MR_EXACT_FP32)
        else:
            want, _ = oracle_mod.memory_read(mk, mv, qk, qv)
            exact, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), flags=ops.MR_EXACT_FP32)
        bank = _fill_bank(ops, mk, mv, mr)
        got = bank.read(T, cu(qk), cu(qv), None if qr is None else cu(qr)).cpu().numpy()
        e_bank = float(np.abs(got[:, :512] - want[:, :512]).max())
        e_fp32 = float(np.abs(exact.cpu().numpy()[:, :512] - want[:, :512]).max())
        worst.append((e_bank, e_fp32))
        assert e_bank <= 2.0 * e_fp32 + 1e-7, (seed, e_bank, e_fp32)
    assert max(e for e, _ in worst) < 5e-6


def test_dropin_memory_read_is_graph_capturable(oracle_mod):
    """rmnet_memory_read_f32's default path (memset of the overflow word, staging, gated bank read, gated exact
    kernel, combine) captured into a HIP graph and replayed -- on clean inputs and, with the same graph, on inputs
    that trip the device-side fallback."""
    from rmnet_amd import ops
    rng = np.random.RandomState(77)
    mk, mv, qk, qv, mr, qr = _random_case(rng, 2, 3, 9, 13, regional=True)
    mr[0, 1] = (0, 12, 0, 8)                                 # (the cell poisoned below lies inside its box)
    d_mk, d_mv, d_qk, d_qv, d_mr, d_qr = cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr)
    out = torch.empty(2, 1024, 9, 13, device=dev())
    side = torch.cuda.Stream(dev())
    side.wait_stream(torch.cuda.current_stream(dev()))
    with torch.cuda.stream(side):
        ops.memory_read(d_mk, d_mv, d_qk, d_qv, d_mr, d_qr, out=out)
    torch.cuda.current_stream(dev()).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.memory_read(d_mk, d_mv, d_qk, d_qv, d_mr, d_qr, out=out)
    out.zero_()
    g.replay()
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)
    mv2 = mv.copy()
    mv2[0, 3, 1, 2, 2] = 5e4                                 # out of the fp16 window: the replayed graph must fall back
    d_mv.copy_(cu(mv2))
    g.replay()
    want2, _ = oracle_mod.regional_memory_read(mk, mv2, qk, qv, mr, qr)
    np.testing.assert_allclose(out.cpu().numpy(), want2, atol=MR_ATOL, rtol=5e-5)
    d_mv.copy_(cu(mv))                                       # and back to the fast path
    g.replay()
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)


def test_c_abi_from_a_native_client(tmp_path):
    """The boundary is a C ABI, not a Python module: tests/native/capi_client.cpp (plain C++ + HIP runtime, no
    torch, no Python, its own scalar host references) is compiled against include/rmnet_hip.h, linked with
    librmnet_hip.so and run: region map, flow update (device and host-buffer entries), memory read (dense and
    regional, default and exact-fp32 flags), and the error codes for a null pointer / a short workspace."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    exe = str(tmp_path / 'capi_client')
    libdir = os.path.join(root, 'rmnet_amd')
    cc = subprocess.run([hipcc, '-O2', '-std=c++17', '-w', os.path.join(root, 'tests', 'native', 'capi_client.cpp'),
                         '-I' + os.path.join(root, 'include'), '-L' + libdir, '-lrmnet_hip', '-Wl,-rpath,' + libdir, '-o', exe],
                        capture_output=True, text=True, timeout=600)
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and 'all ops OK' in run.stdout, (run.returncode, run.stdout[-1000:], run.stderr[-1000:])


# ----------------------------------------------------------------------------- round 3: remaining parity holes
def test_region_map_loosen_clamp_count_branches_vs_reference_boxes(golden_dir):
    """G1's point-count fallback and four clamp expressions (reg_att_map_generator.cu:55-77) on the GPU against
    expectations that do not involve oracle/rmnet_oracle.c: the reference's own box finder (utils/helpers.py:93-102,
    run by tests/golden/make_golden.py section region_fuzz) + the .cu's expressions restated in
    tests/golden/cases.py:boxes_after_loosen, for n_bbox_loose_pixels in {0, 1, 63, 64, 65} x n_pts_threshold in
    {1, 9, 10, 11} on 40 masks with box edges at / before / after every switching distance."""
    import sys
    sys.path.insert(0, golden_dir)
    import cases
    from rmnet_amd import ops
    g = np.load(os.path.join(golden_dir, 'region_fuzz.npz'))
    assert len(cases.REGION_FUZZ_SHAPES) == 40
    for i, (B, K, H, W) in enumerate(cases.REGION_FUZZ_SHAPES):
        m = cases.region_fuzz_case(i)
        assert float(m.astype(np.float64).sum()) == float(g['case%02d.checksum' % i])
        md = cu(m)
        for L in cases.REGION_FUZZ_LOOSE:
            for npt in cases.REGION_FUZZ_NPTS:
                want = cases.boxes_after_loosen(g['case%02d.tight' % i], g['case%02d.npts' % i], K, H, W, npt, L).reshape(B, K, 4)
                att, bb, _ = ops.region_map(md, 0.5, npt, L)
                assert np.array_equal(bb.cpu().numpy(), want), (i, L, npt)
                a = att.cpu().numpy()
                for b in range(B):
                    assert not a[b, 0].any()
                    for k in range(1, K):
                        x0, x1, y0, y1 = want[b, k]
                        ref = np.zeros((H, W), np.float32)
                        ref[y0:y1 + 1, x0:x1 + 1] = 1
                        assert np.array_equal(a[b, k], ref), (i, L, npt, b, k)


def test_long_memory_stress_size_meets_the_oracle_on_sampled_queries(oracle_mod):
    """BASELINE configs[4] at its exact shape -- 720p (45x80 cells), 3 objects, T = 20 memory frames, regional --
    against the ORACLE (not a second HIP kernel) on 288 sampled query cells per object: inside the query box, on
    its edges, and masked cells outside it; through the bank (bk_main) and through the drop-in entry."""
    from rmnet_amd import ops
    no, T, h, w = 3, 20, 45, 80
    rng = np.random.RandomState(420)
    g = torch.Generator(device='cpu').manual_seed(420)
    mk = (torch.randn(no, 128, T, h, w, generator=g) * 0.6)
    mv = torch.randn(no, 512, T, h, w, generator=g)
    qk = (torch.randn(no, 128, h, w, generator=g) * 0.6)
    qv = torch.randn(no, 512, h, w, generator=g)
    mr = np.zeros((no, T, 4), np.int32)
    qr = np.zeros((no, 4), np.int32)
    for o in range(no):
        for t in range(T):
            x0, y0 = rng.randint(0, w // 2), rng.randint(0, h // 2)
            mr[o, t] = (x0, x0 + rng.randint(w // 3, w // 2 + 1), y0, y0 + rng.randint(h // 3, h // 2 + 1))
        mr[o, 7] = (1, 0, 1, 0)                                      # one frame with an empty box
        x0, y0 = rng.randint(0, w // 3), rng.randint(0, h // 3)
        qr[o] = (x0, x0 + w // 2, y0, y0 + h // 2)
    # sample: the four corners and edge midpoints of every query box, random cells, cells outside the boxes
    qidx = set()
    for o in range(no):
        x0, x1, y0, y1 = qr[o]
        for (yy, xx) in ((y0, x0), (y0, x1), (y1, x0), (y1, x1), ((y0 + y1) // 2, x0), (y0, (x0 + x1) // 2)):
            qidx.add(yy * w + xx)
    qidx.update(int(v) for v in rng.choice(h * w, size=288 - len(qidx), replace=False))
    qidx = np.array(sorted(qidx)[:288], np.int32)
    want = oracle_mod.regional_memory_read_sampled(mk.numpy(), mv.numpy(), qk.numpy(), mr, qr, qidx)   # [no, nq, 512]
    d = dev()
    bank = ops.MemoryBank(no, T, h, w, d)
    for t in range(T):
        bank.append(t, mk[:, :, t].contiguous().to(d), mv[:, :, t].contiguous().to(d), cu(mr[:, t]))
    got = bank.read(T, qk.to(d), qv.to(d), cu(qr))
    pick = lambda out: out[:, :512].reshape(no, 512, h * w)[:, :, torch.from_numpy(qidx.astype(np.int64)).to(d)].permute(0, 2, 1).cpu().numpy()
    np.testing.assert_allclose(pick(got), want, atol=MR_ATOL, rtol=MR_RTOL)
    assert bank.overflow_count() == 0
    via, _ = ops.memory_read(mk.to(d), mv.to(d), qk.to(d), qv.to(d), cu(mr), cu(qr))
    np.testing.assert_allclose(pick(via), want, atol=MR_ATOL, rtol=MR_RTOL)
    # [r5] the arithmetic the frame loop takes at this configuration (3 objects: 'auto' -> qx) and the plain fp16-operand one, same
    # bank, at their own bar (2^-10 of the largest value; masked cells come from the column sums: fp32-class in every mode)
    inbox = np.zeros((no, h * w), bool)
    for o in range(no):
        m = np.zeros((h, w), bool)
        m[qr[o, 2]:qr[o, 3] + 1, qr[o, 0]:qr[o, 1] + 1] = True
        inbox[o] = m.reshape(-1)
    sel = inbox[:, qidx]                                             # [no, nq]
    for mode in ('qx', 'f16'):
        bank.precision = mode
        g16 = pick(bank.read(T, qk.to(d), qv.to(d), cu(qr)))
        err = np.abs(g16 - want)
        assert float(err.max()) <= F16_ATOL_REL * float(mv.abs().max()), (mode, float(err.max()))
        assert float(err.mean()) < 1e-4, (mode, float(err.mean()))
        np.testing.assert_allclose(g16[~sel], want[~sel], atol=MR_ATOL, rtol=MR_RTOL)
    bank.precision = 'split'
    # the q_val half is q_val * box, exactly
    box = torch.zeros(no, 1, h, w)
    for o in range(no):
        box[o, 0, qr[o, 2]:qr[o, 3] + 1, qr[o, 0]:qr[o, 1] + 1] = 1
    assert torch.equal(got[:, 512:].cpu(), qv * box)


@pytest.mark.parametrize('K,n_obj', [(6, 5), (11, 1)])
def test_whole_loop_480p_five_objects_and_loader_channel_count(K, n_obj, oracle_mod):
    """480x854, two segmented frames of the device-resident loop against the CPU path (OracleRMNet): (a) 5 objects
    (BASELINE configs[2]); (b) K = 11 mask channels with ONE object, as the reference's test loader feeds every
    1-object video (config.py:137 N_MAX_OBJECTS = 10, utils/data_loaders.py:207-232): 9 channels stay empty.
    Bar: label IoU >= 0.999 per object (north_star).  Probabilities: within 1e-3 for one object; with five the 5-way
    soft aggregation (a product of five probabilities, then a logit) amplifies the convolutions' GPU-vs-CPU rounding on a
    handful of boundary pixels -- measured 5e-3 on 18 of 2.4 M values, identically with the exact-fp32 read (_exact=True,
    tools/dbg_loop5.py) -- so the bar there is: fewer than 1e-4 of the values differ by more than 1e-3, none by more than
    2e-2, and the split-fp16 read adds nothing to what the exact read shows."""
    from rmnet_amd.synthetic import synthetic_clip
    prod, ref = _nets(oracle_mod, read_precision='split')       # (the probability statements below are about the fp32-class read;
    prod.fuse_epilogues()                                       #  the default fp16-operand read: label IoU, at the end)
    H, W, N = 480, 854, 3
    frames, masks, flows, n_objects = synthetic_clip(N, n_obj + 1, H, W, seed=K, size=1.1 if n_obj > 1 else 2.1)
    if K > n_obj + 1:
        pad = torch.zeros(1, N, K - n_obj - 1, H, W, dtype=masks.dtype)
        masks = torch.cat([masks, pad], dim=2)
    with torch.no_grad():
        est_cpu = ref(frames, masks, flows, n_objects, 1)
        est = prod(frames, masks, flows, n_objects, 1).cpu()
    assert est.shape == (1, N, K, H, W)
    diff = (est - est_cpu).abs()
    if n_obj == 1:
        assert float(diff.max()) < 1e-3
    else:
        with torch.no_grad():
            est_x = prod(frames, masks, flows, n_objects, 1, _exact=True).cpu()
        diff_x = (est_x - est_cpu).abs()
        assert float(diff.max()) < 2e-2 and float((diff > 1e-3).float().mean()) < 1e-4
        assert int((diff > 1e-3).sum()) <= int((diff_x > 1e-3).sum()) + 16      # same class as the exact-fp32 read
    lab, lab_cpu = est.argmax(2).numpy(), est_cpu.argmax(2).numpy()
    for k in range(1, n_obj + 1):
        assert oracle_mod.iou(lab[:, 1:] == k, lab_cpu[:, 1:] == k) >= 0.999
    if K > n_obj + 1:
        assert float(est[:, 1:, n_obj + 1:].max()) < 1e-6          # channels of objects that do not exist
    prod.read_precision = 'f16'                                     # the frame loop's default arithmetic: the north star's IoU bar
    with torch.no_grad():
        lab16 = prod(frames, masks, flows, n_objects, 1).argmax(2).cpu().numpy()
    for k in range(1, n_obj + 1):
        assert oracle_mod.iou(lab16[:, 1:] == k, lab_cpu[:, 1:] == k) >= 0.999


def test_exact_fallback_with_tiny_memory_boxes_is_not_nan(oracle_mod):
    """Round-2 advisor finding: with the transient bank's areas in its plan, the exact-fp32 fallback (taken on the
    device when a value leaves the fp16 window) planned sum_t ceil(area_t / 32) memory tiles although it walks
    ceil(M / 32) compacted ones; with small boxes a split then began beyond the last cell and produced NaN.
    8 frames of 2x5 cells, one value of 2e3: the drop-in entry must match the oracle."""
    from rmnet_amd import ops
    rng = np.random.RandomState(77)
    no, T, h, w = 2, 8, 9, 13
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=True)
    for o in range(no):
        for t in range(T):
            x0, y0 = rng.randint(0, w - 5), rng.randint(0, h - 2)
            mr[o, t] = (x0, x0 + 4, y0, y0 + 1)                      # 2 x 5 cells
        qr[o] = (1, 11, 1, 7)
    mv[0, 3, 2, mr[0, 2, 2], mr[0, 2, 0]] = 2.0e3                    # inside a box: out of the bank's window
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    via, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr))
    got = via.cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=MR_ATOL, rtol=5e-5)


def test_memory_longer_than_512_frames(oracle_mod):
    """models/rmnet.py:416-426 grows the memory without bound; T = 600 memorised frames (5x6 grid) through the bank
    and through the drop-in entry (default and exact fp32), against the oracle."""
    from rmnet_amd import ops
    rng = np.random.RandomState(600)
    no, T, h, w = 1, 600, 5, 6
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=True)
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    bank = _fill_bank(ops, mk, mv, mr)
    np.testing.assert_allclose(bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)
    for flags in (0, ops.MR_EXACT_FP32):
        got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr), flags=flags)
        np.testing.assert_allclose(got.cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)


def test_bank_at_one_launch_limit(oracle_mod):
    """One launch takes 2048 memorised frames (the workgroups' LDS tile prefix is full): read in both arithmetic modes on a
    3x4 grid (keeps the oracle fast)."""
    from rmnet_amd import ops
    assert ops.BANK_MAX_SLOTS == 2048
    rng = np.random.RandomState(2048)
    no, T, h, w = 1, 2048, 3, 4
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=True)
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    bank = ops.MemoryBank(no, T, h, w, dev())
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), cu(mr[:, t]))
    np.testing.assert_allclose(bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)
    bank.precision = 'f16'
    _f16_bars(bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy(), want, float(np.abs(mv).max()))
    assert bank.overflow_count() == 0


@pytest.mark.parametrize('T,regional', [(2049, True), (4100, True), (4100, False)])
def test_memory_beyond_one_launch_is_read_in_chunks(T, regional, oracle_mod):
    """models/rmnet.py:416-426 has no bound on the memory: T = 2049 ... 4100 memorised frames through the bank -- chunks of
    2048 slots, each an ordinary read that also leaves the soft-max state of its queries, merged by bk_chain -- against
    the oracle, both arithmetic modes; masked query cells, a chunk whose boxes are all empty, and the device-counter entry
    (refused for such a bank: the chunks are planned on the host)."""
    from rmnet_amd import _lib, ops
    rng = np.random.RandomState(T)
    no, h, w = 2, 3, 4
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=regional)
    if regional:
        mr[1, 2048:] = (1, 0, 1, 0)                    # object 1: nothing memorised inside a box after the first chunk
        qr[0] = (1, 3, 0, 1)
        want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    else:
        want, _ = oracle_mod.memory_read(mk, mv, qk, qv)
    bank = ops.MemoryBank(no, T + 3, h, w, dev())
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), None if mr is None else cu(mr[:, t]))
    q_rects = None if qr is None else cu(qr)
    got = bank.read(T, cu(qk), cu(qv), q_rects).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=MR_ATOL, rtol=MR_RTOL)
    bank.precision = 'f16'
    _f16_bars(bank.read(T, cu(qk), cu(qv), q_rects).cpu().numpy(), want, float(np.abs(mv).max()))
    assert bank.overflow_count() == 0 and bank.timeout_count() == 0
    bank.precision = 'split'
    np.testing.assert_allclose(bank.read(2048, cu(qk), cu(qv), q_rects).cpu().numpy()[:, 512:], want[:, 512:], atol=0, rtol=0)   # one launch still works
    bank.committed = T - 1
    np.testing.assert_allclose(bank.read_staged(cu(qk), cu(qv), q_rects).cpu().numpy(), want, atol=MR_ATOL, rtol=MR_RTOL)   # host count for long banks
    with pytest.raises(RuntimeError):
        bank.read(1, cu(qk), cu(qv), q_rects, _t_dev=bank.n_dev)


def test_forward_replays_one_hip_graph_for_the_whole_clip(oracle_mod):
    """RMNet.forward captures the frame step once (second segmented frame) and replays it while the memory
    grows -- the bank's slot / frame count are a DEVICE counter, not kernel arguments.  A 10-frame clip with memorize_every = 3
    (the memory grows from 1 to 4 frames under replay) and an object that appears mid-clip (logit edits outside the graph):
    graph == eager == CPU path."""
    prod, ref = _nets(oracle_mod)
    prod.fuse_epilogues()
    frames, masks, flows, n_objects = _clip_with_late_object(10, 96, 160, seed=31)
    captured = []
    orig = prod._capture_frame_step
    prod._capture_frame_step = lambda *a, **k: (captured.append(1), orig(*a, **k))[1]
    with torch.no_grad():
        est_g = prod(frames, masks, flows, n_objects, 3, graph=True).cpu()
        est_e = prod(frames, masks, flows, n_objects, 3, graph=False).cpu()
        est_cpu = ref(frames, masks, flows, n_objects, 3)
    assert captured == [1]                                        # one capture, nine replays
    assert float((est_g - est_e).abs().max()) < 1e-3            # (MIOpen may pick another algorithm under capture)
    assert float((est_g - est_cpu).abs().max()) < 1e-3
    assert (est_g.argmax(2) == est_cpu.argmax(2)).float().mean() > 0.999


def test_static_half_of_the_read_keeps_the_reference_semantics():
    """The part of MemoryReader.forward that does not depend on the soft-max -- written by bk_main's work queue --:
    (a) q_val x box is a MULTIPLICATION (models/rmnet.py:358): -0.0, Inf and NaN outside the box give -0.0 / NaN / NaN,
    exactly like torch; (b) masked query cells read the mean of m_val over all T*h*w cells, for several objects with
    different boxes in one launch, a grid whose rows cannot be moved 16 bytes at a time (5 x 7), and an object whose
    memory boxes are all empty (every cell reads 0)."""
    from rmnet_amd import ops
    rng = np.random.RandomState(91)
    for (no, T, h, w) in ((3, 2, 9, 12), (2, 3, 5, 7), (9, 2, 30, 54)):
        mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=True)
        qr[0] = (2, w - 3, 1, h - 2)
        mr[no - 1, :] = (1, 0, 1, 0)                                    # nothing memorised inside any box
        qv[0, 3, 0, 0] = np.inf
        qv[0, 4, 0, 1] = np.nan
        qv[0, 5, h - 1, w - 1] = -3.0
        qv[0, 6, 1, 2] = np.nan                                         # inside the box of object 0: stays NaN as well
        bank = _fill_bank(ops, mk, mv, mr)
        got = bank.read(T, cu(qk), cu(qv), cu(qr)).cpu()
        box = torch.zeros(no, 1, h, w)
        for o in range(no):
            x0, x1, y0, y1 = qr[o]
            if x0 <= x1 and y0 <= y1:
                box[o, 0, y0:y1 + 1, x0:x1 + 1] = 1
        want_q = torch.from_numpy(qv) * box                             # the reference's expression
        assert torch.equal(torch.isnan(got[:, 512:]), torch.isnan(want_q))
        assert torch.equal(torch.nan_to_num(got[:, 512:], nan=7.0), torch.nan_to_num(want_q, nan=7.0))
        assert torch.equal(torch.signbit(got[:, 512:]), torch.signbit(want_q))
        mean = torch.from_numpy(oracle_masked_mean(mv, mr))             # [no, 512]
        out_mem = got[:, :512]
        outside = (box == 0).expand(no, 512, h, w)
        np.testing.assert_allclose(out_mem[outside].numpy(), mean[:, :, None, None].expand(no, 512, h, w)[outside].numpy(),
                                   atol=MR_ATOL, rtol=MR_RTOL)
        assert float(out_mem[no - 1].abs().max()) == 0.0                # empty memory: soft-max over zeros of zeros


def oracle_masked_mean(mv, mr):
    """Mean over ALL T*h*w cells of the box-masked values (what a query cell with all-zero logits reads): numpy, fp64."""
    no, C, T, h, w = mv.shape
    m = np.zeros((no, 1, T, h, w))
    for o in range(no):
        for t in range(T):
            x0, x1, y0, y1 = mr[o, t]
            if x0 <= x1 and y0 <= y1:
                m[o, 0, t, max(y0, 0):y1 + 1, max(x0, 0):x1 + 1] = 1
    return ((mv.astype(np.float64) * m).sum(axis=(2, 3, 4)) / (T * h * w)).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------
# fp16-operand mode of the bank read (RMNET_BANK_F16 / RMNET_MR_F16): opt-in, about twice as fast, NOT fp32-class.
# What it promises (include/rmnet_hip.h): K, V, q and the soft-max weights rounded to fp16 (11 significant bits), fp32
# accumulate, the weights normalised by the sum of the ROUNDED weights.  The bars below are that arithmetic's, written
# against the oracle (fp32 semantics): |error| <= 2^-10 of the largest |value| the read-out can average (4e-3 for
# N(0,1) values; measured 3e-4 at worst, 5e-6 when hundreds of cells contribute), mean |error| < 1e-4.  What the
# north star asks of it -- mask IoU within 1e-3 of the CPU path on whole clips -- is the second group of tests.
F16_ATOL_REL = 2.0 ** -10


def _f16_bars(got, want, vmax, smax=0.0):
    """``smax`` = largest |logit| of the case: the rounding of K and q is relative, so a logit S is off by about
    |S| * 2^-11 and the weights of two competing cells by that fraction -- the bar widens with max(1, smax / 8)."""
    err = np.abs(got[:, :512] - want[:, :512])
    wide = max(1.0, smax / 8.0)
    assert not np.isnan(got).any()
    assert float(err.max()) <= F16_ATOL_REL * vmax * wide, float(err.max())
    assert float(err.mean()) < 1e-4 * wide, float(err.mean())
    np.testing.assert_array_equal(got[:, 512:], want[:, 512:])      # the q_val half is a copy: exact in every mode


@pytest.mark.parametrize('no,T,h,w,regional', [
    (1, 1, 4, 5, False), (2, 3, 9, 13, True), (1, 5, 30, 54, True), (1, 7, 16, 24, False), (2, 9, 10, 7, True),
    (14, 2, 6, 9, True), (70, 1, 4, 5, True), (5, 3, 30, 54, True), (5, 5, 30, 54, True), (3, 20, 12, 20, True),
    (1, 70, 5, 6, True), (50, 2, 12, 20, False), (24, 3, 12, 20, True),
    (64, 2, 16, 24, False), (60, 3, 16, 24, True), (20, 1, 30, 54, False)])      # [r6] more pairs than workgroups: rounds
def test_bank_read_f16_mode_vs_oracle(no, T, h, w, regional, oracle_mod):
    """Same cases as test_bank_read_vs_oracle (ragged boxes, empty boxes, > 12 and > 64 objects, odd tile counts --
    a step of the fp16 loop is TWO tiles, which may belong to different frames), fp16-operand arithmetic."""
    from rmnet_amd import ops
    rng = np.random.RandomState(no * 1000 + T * 100 + h + 1)
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=regional)
    bank = ops.MemoryBank(no, T + 2, h, w, dev(), precision='f16')
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), None if mr is None else cu(mr[:, t]))
    if regional:
        want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
        got = bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy()
        # cells outside the query box come from the column sums, not from the MFMAs: fp32-class in this mode too
        inbox = np.zeros((no, 1, h, w), bool)
        for o in range(no):
            x0, x1, y0, y1 = qr[o]
            inbox[o, 0, max(y0, 0):y1 + 1, max(x0, 0):x1 + 1] = x0 <= x1 and y0 <= y1
        np.testing.assert_allclose((got[:, :512] * ~inbox), (want[:, :512] * ~inbox), atol=MR_ATOL, rtol=MR_RTOL)
    else:
        want, _ = oracle_mod.memory_read(mk, mv, qk, qv)
        got = bank.read(T, cu(qk), cu(qv)).cpu().numpy()
    _f16_bars(got, want, float(np.abs(mv).max()))
    # the bank itself is mode-independent: the same bank read in the default mode meets the fp32-class bar
    bank.precision = 'split'
    got3 = bank.read(T, cu(qk), cu(qv), cu(qr) if regional else None).cpu().numpy()
    np.testing.assert_allclose(got3, want, atol=MR_ATOL, rtol=MR_RTOL)


def test_f16_mode_peaked_softmax_and_late_spike(oracle_mod):
    """A soft-max dominated by one cell returns that cell's value with V's fp16 rounding only (the denominator is the
    sum of the rounded weights); a late spike forces the deferred running maximum to bump in the middle of a step."""
    from rmnet_amd import ops
    rng = np.random.RandomState(5)
    no, T, h, w = 2, 4, 8, 16
    mk, mv, qk, qv, _, _ = _random_case(rng, no, T, h, w, regional=False)
    mk[:, :, 3, h - 1, w - 2] = qk[:, :, 2, 3] * 9.0           # query cell (2,3) is dominated by memory cell (3, h-1, w-2)
    # (logit ~ 36.  The rounding of K and q is RELATIVE, so a logit S carries an error of about |S| * 2^-11 * 0.3 and a
    #  weight e^S that relative error: with logits in the hundreds two competing cells are mis-weighted by percents -- a
    #  x40 spike here gives 7e-3.  That is the price of 11-bit operands and the reason the mode is opt-in.)
    bank = ops.MemoryBank(no, T, h, w, dev(), precision='f16')
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), None)
    want, _ = oracle_mod.memory_read(mk, mv, qk, qv)
    got = bank.read(T, cu(qk), cu(qv)).cpu().numpy()
    _f16_bars(got, want, float(np.abs(mv).max()))
    peak = np.abs(got[:, :512, 2, 3] - mv[:, :, 3, h - 1, w - 2])
    assert float((peak / np.maximum(np.abs(mv[:, :, 3, h - 1, w - 2]), 2.0 ** -8)).max()) <= 2.0 ** -10   # V's rounding (2^-11) + a weight-sized rest


def test_dropin_memory_read_f16_flag(golden_dir, oracle_mod):
    """rmnet_memory_read_f32(..., RMNET_MR_F16): staging + fp16-operand read, dense and regional, on the reference's own
    golden MemoryReader vectors and on a random case; the out-of-window fallback still goes to the exact kernel."""
    from rmnet_amd import ops
    g = np.load(os.path.join(golden_dir, 'memory_reader.npz'))
    names = sorted({k.split('.')[0] for k in g.files})
    checked = 0
    for name in names:
        if name + '.m_key' not in g.files or name + '.mem_val' not in g.files:
            continue
        mk, mv, qk, qv = (g[name + '.' + k].astype(np.float32) for k in ('m_key', 'm_val', 'q_key', 'q_val'))
        if mk.shape[1] != 128 or mv.shape[1] != 512:
            continue
        got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), flags=ops.MR_F16)
        no_, _, T_, h_, w_ = mk.shape
        smax = float(np.abs(np.einsum('ocj,oci->oji', mk.reshape(no_, 128, -1), qk.reshape(no_, 128, -1))).max()) / np.sqrt(128.0)
        _f16_bars(got.cpu().numpy(), g[name + '.mem_val'].astype(np.float32), float(np.abs(mv).max()), smax)   # ("peaky": logits to 32)
        checked += 1
    assert checked >= 1
    rng = np.random.RandomState(77)
    mk, mv, qk, qv, mr, qr = _random_case(rng, 3, 4, 12, 20, regional=True)
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr), flags=ops.MR_F16)
    _f16_bars(got.cpu().numpy(), want, float(np.abs(mv).max()))
    mv2 = mv.copy()
    mv2[1, 7, 2, 3, 4] = 5000.0                                  # outside the bank's window: the exact kernel answers
    want2, _ = oracle_mod.regional_memory_read(mk, mv2, qk, qv, mr, qr)
    got2, _ = ops.memory_read(cu(mk), cu(mv2), cu(qk), cu(qv), cu(mr), cu(qr), flags=ops.MR_F16)
    np.testing.assert_allclose(got2.cpu().numpy(), want2, atol=MR_ATOL * 50, rtol=MR_RTOL)


def test_f16_mode_whole_clip_meets_the_iou_bar(oracle_mod):
    """The north star's correctness bar -- mask IoU within 1e-3 of the CPU path on identical inputs -- for the frame loop
    with ``read_precision='f16'`` at the headline workload (480x854, 1 object, memory growing to T = 5) against
    OracleRMNet, fused and un-fused; probabilities within 1e-3."""
    from rmnet_amd import networks
    from rmnet_amd.rmnet import RMNet
    from rmnet_amd.synthetic import synthetic_clip
    prod = networks.procedural_init_(RMNet(None, read_precision='f16'))
    ref = oracle_mod.OracleRMNet()
    ref.load_state_dict(prod.state_dict())
    prod, ref = prod.to(dev()).eval(), ref.eval()
    N, K, H, W = 6, 2, 480, 854
    frames, masks, flows, n_objects = synthetic_clip(N, K, H, W, seed=11, size=1.3)
    with torch.no_grad():
        est_cpu = ref(frames, masks, flows, n_objects, 1)
        est = prod(frames, masks, flows, n_objects, 1).cpu()
        prod.fuse_epilogues()
        est_f = prod(frames, masks, flows, n_objects, 1).cpu()
    lab_cpu = est_cpu.argmax(2).numpy()
    for e in (est, est_f):
        assert float((e - est_cpu).abs().max()) < 1e-3
        lab = e.argmax(2).numpy()
        assert oracle_mod.iou(lab[:, 1:] == 1, lab_cpu[:, 1:] == 1) >= 0.999


_CALIB_CASES = {   # objects, H, W, memorize_every, seed, blob size, frames
    '1obj-480p': (1, 480, 854, 5, 1, 2.1, 8),
    '3obj-480p': (3, 480, 854, 5, 3, 1.1, 20),
    '5obj-480p': (5, 480, 854, 2, 4, 1.6, 14),     # (the configuration on which the fp16-operand loop misses the bar: 0.9989 at 20 frames)
}   # (the 720p 3-object clip and the 30-frame runs are in profiles/r04_iou_calibration.md: tools/iou_calib.py; not re-run by the suite)


_CALIB_PARAMS = [pytest.param(c, m, marks=pytest.mark.xfail(strict=False, reason='documented miss: the fp16-operand read loses 1.1e-3 of '
                                                           'IoU on this 5-object clip (profiles/r04_iou_calibration.md); auto does not use it there'))
                 if (c, m) == ('5obj-480p', 'f16') else pytest.param(c, m)
                 for c in sorted(_CALIB_CASES) for m in ('auto', 'exact', 'f16')]      # ('qx' is opt-in since round 6: test_key_temperature_sweep, live fixtures)


@pytest.mark.parametrize('case,mode', _CALIB_PARAMS)
def test_iou_bar_against_the_cpu_path_on_long_clips(case, mode, oracle_mod):
    """The north star's bar -- mask IoU within 1e-3 of the CPU path -- measured against THAT path (OracleRMNet on the host
    cores) on 20-frame clips with 3 / 5 objects (12 frames with one) at 480x854, for the GPU loop in its default configuration
    ('auto'), with the exact-fp32 read, with the exact-query fp16 read ('qx') and with the plain fp16-operand read forced.  ONE bar for all:
    >= 0.999 per object.  The fp16-operand read misses it on the 5-object clip (0.9989: xfail, not strict -- another MIOpen solver or box may nudge it over the bar; a documented miss --
    'auto' does not use that arithmetic for several objects); profiles/r05_iou_calibration.md has the full table
    (tools/iou_calib.py makes it).  NOTE: the one-object clip here is SATURATED (its mask is the whole frame): it checks the
    plumbing of a long clip, not the read -- the one-object bar that can fail is test_live_boundary_clips_meet_the_bar_in_every_arithmetic."""
    from rmnet_amd.synthetic import synthetic_clip
    n_obj, H, W, every, seed, size, N = _CALIB_CASES[case]
    prod, ref = _nets(oracle_mod, 'auto' if mode == 'exact' else mode)
    prod.fuse_epilogues()
    frames, masks, flows, n_objects = synthetic_clip(N, n_obj + 1, H, W, seed=seed, size=size)
    est_cpu, _ = _cpu_path(oracle_mod, ref, 'calib-' + case, frames, masks, flows, n_objects, every)
    lab_cpu = est_cpu.argmax(2).numpy()
    with torch.no_grad():
        est = prod(frames, masks, flows, n_objects, every, _exact=(mode == 'exact')).cpu()
    lab = est.argmax(2).numpy()
    iou = min(oracle_mod.iou(lab[:, 1:] == k, lab_cpu[:, 1:] == k) for k in range(1, n_obj + 1))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'iou_bar_test_table.txt'), 'a') as fh:
            fh.write('%s %s (auto -> %s): %.5f\n' % (case, mode, prod.resolve_read_precision([n_obj]), iou))
    assert iou >= 0.999, (case, mode, iou)


def test_f16_mode_through_strides_partial_reads_and_graph_replay(oracle_mod):
    """The fp16-operand switch on the remaining call shapes: the drop-in entry with a memory tensor of capacity > T (channel
    strides, a late spike that bumps the deferred maximum in the middle of a two-tile step); and the frame loop with
    ``read_precision='f16'`` replayed as one HIP graph == the same loop run eagerly (the flag is baked into the capture)."""
    from rmnet_amd import networks, ops
    from rmnet_amd.rmnet import RMNet
    from rmnet_amd.synthetic import synthetic_clip
    rng = np.random.RandomState(5)
    no, T, Tcap, h, w = 2, 3, 5, 6, 10
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, Tcap, h, w, regional=True)
    mk[:, :, 2, h - 1, w - 1] = qk[:, :, 2, 3] * 9.0
    mr[:, 2] = (0, w - 1, 0, h - 1)
    qr[:] = (0, w - 1, 0, h - 1)
    want, _ = oracle_mod.regional_memory_read(mk[:, :, :T], mv[:, :, :T], qk, qv, mr[:, :T], qr)
    got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr[:, :T]), cu(qr), T=T, flags=ops.MR_F16)
    _f16_bars(got.cpu().numpy(), want, float(np.abs(mv).max()), smax=36.0)
    want_d, _ = oracle_mod.memory_read(mk[:, :, :T], mv[:, :, :T], qk, qv)
    got_d, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), T=T, flags=ops.MR_F16)
    _f16_bars(got_d.cpu().numpy(), want_d, float(np.abs(mv).max()), smax=36.0)

    net = networks.procedural_init_(RMNet(None, read_precision='f16')).to(dev()).eval()
    net.fuse_epilogues()
    frames, masks, flows, n_objects = synthetic_clip(9, 3, 96, 160, seed=17)
    with torch.no_grad():
        est_g = net(frames, masks, flows, n_objects, 3, graph=True).cpu()
        est_e = net(frames, masks, flows, n_objects, 3, graph=False).cpu()
    assert float((est_g - est_e).abs().max()) < 1e-3            # (MIOpen may pick another algorithm under capture)
    assert (est_g.argmax(2) == est_e.argmax(2)).float().mean() > 0.999


# ----------------------------------------------------------------------------- round 4: parity holes of the round-3 verdict
def _blob_masks(B, K, H, W, seed):
    """Soft blob masks with smooth edges (a >= 0.5 decision never sits within an ulp of the threshold by accident)."""
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    m = np.zeros((B, K, H, W), np.float32)
    for b in range(B):
        for k in range(1, K):
            cy, cx = rng.uniform(0.25, 0.75) * H, rng.uniform(0.25, 0.75) * W
            ry, rx = rng.uniform(0.08, 0.22) * H, rng.uniform(0.08, 0.22) * W
            d = np.sqrt(((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2)
            m[b, k] = 1.0 / (1.0 + np.exp((d - 1.0) * 9.3))
    return m


@pytest.mark.parametrize('case', ['golden', '480x854', '97x131'])
def test_fused_warp_boxes_against_the_cpu_oracle(case, golden_dir, oracle_mod):
    """(f2) against the ORACLE, not against the product: rmnet_region_map_warped_f32's boxes / cell rectangles / box map
    == the CPU path's warp (torch CPU grid_sample, models/rmnet.py:252-278) followed by the C restatement of the region
    map (oracle/rmnet_oracle.c) and of the cell rectangles."""
    from rmnet_amd import ops
    if case == 'golden':
        g = np.load(os.path.join(golden_dir, 'rmnet_clip.npz'))
        m, f = g['warp.in'].astype(np.float32), g['warp.flow'].astype(np.float32)
    else:
        H, W = (480, 854) if case == '480x854' else (97, 131)
        rng = np.random.RandomState(H)
        m = _blob_masks(2, 3, H, W, seed=H)
        f = (rng.randn(2, 2, 1, 1) * 6.0 + rng.randn(2, 2, H, W) * 0.7).astype(np.float32)      # drift + jitter
        f[1, :, : H // 10] += 2.0 * max(H, W)                                                   # a band sampled outside the frame
    B, K, H, W = m.shape
    warped = oracle_mod.OracleRMNet.warp(torch.from_numpy(m), torch.from_numpy(f))[0].numpy()
    att_cpu, bb_cpu = oracle_mod.region_map(warped)
    lw, lh = (16 - W % 16) % 16 // 2, (16 - H % 16) % 16 // 2
    ch, cw = (H + 15) // 16, (W + 15) // 16
    rc_cpu = oracle_mod.cell_rects(bb_cpu, lw, lh, ch, cw)
    att, bb, rc = ops.region_map(cu(m), cell_grid=(lw, lh, 16, ch, cw), flow=cu(f))
    assert np.array_equal(bb.cpu().numpy(), bb_cpu)
    assert np.array_equal(att.cpu().numpy(), att_cpu)
    rc, keep = rc.cpu().numpy()[:, 1:], bb_cpu[:, 1:, 1] >= bb_cpu[:, 1:, 0]
    assert np.array_equal(rc[keep], rc_cpu[:, 1:][keep])


def test_query_outside_the_fp16_window_is_counted_and_read_exactly(oracle_mod):
    """|q_key| = 1e4 (x 8.2 after scaling: beyond fp16) and a NaN-free Inf-free case otherwise: the read counts it in the
    bank's overflow word (it used to saturate silently), the drop-in entry falls back to the exact-fp32 kernels on the
    device and matches the oracle; a query element outside the query box does not count."""
    from rmnet_amd import ops
    rng = np.random.RandomState(41)
    mk, mv, qk, qv, mr, qr = _random_case(rng, 2, 3, 9, 13, regional=True)
    mr[:] = (0, 12, 0, 8)
    qr[:] = (1, 11, 1, 7)
    qk[1, 5, 4, 6] = 1.0e4                                # inside the box of object 1
    want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    for prec in ('split', 'f16'):
        bank = _fill_bank(ops, mk, mv, mr)
        bank.precision = prec
        assert bank.overflow_count() == 0
        bank.read(3, cu(qk), cu(qv), cu(qr))
        assert bank.overflow_count() >= 1, prec
        assert bank.timeout_count() == 0
        via, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), cu(mr), cu(qr), flags=ops.MR_F16 if prec == 'f16' else 0)
        np.testing.assert_allclose(via.cpu().numpy(), want, atol=MR_ATOL, rtol=5e-5)
    qk[1, 5, 4, 6] = 0.3
    qk[1, 5, 0, 0] = 1.0e4                                # outside the query box: multiplied by 0 (models/rmnet.py:357), not counted
    bank = _fill_bank(ops, mk, mv, mr)
    want2, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
    got = bank.read(3, cu(qk), cu(qv), cu(qr))
    assert bank.overflow_count() == 0
    np.testing.assert_allclose(got.cpu().numpy(), want2, atol=MR_ATOL, rtol=5e-5)


def test_device_counter_out_of_range_is_flagged_and_timeouts_stay_zero(oracle_mod):
    """rmnet_bank_*_at with a device counter past the capacity: nothing is written / the count is clamped (memory safety)
    AND the bank's overflow word gets its sticky bit, so a desynchronised graph replay cannot go unnoticed.  The separate
    time-out word is zero after ordinary and repeated reads."""
    from rmnet_amd import ops
    rng = np.random.RandomState(43)
    mk, mv, qk, qv, mr, qr = _random_case(rng, 2, 3, 9, 13, regional=True)
    bank = ops.MemoryBank(2, 3, 9, 13, dev())
    for t in range(3):
        bank.stage(cu(mk[:, :, t]), cu(mv[:, :, t]), cu(mr[:, t]))
        bank.commit()
    bank.assert_synced()
    for _ in range(20):
        bank.read(3, cu(qk), cu(qv), cu(qr))
    assert bank.overflow_count() == 0 and bank.timeout_count() == 0
    bank.n_dev.fill_(3)                                   # the bank is full: a further staged append must not write ...
    lib = __import__('rmnet_amd._lib', fromlist=['load']).load()
    rc = lib.rmnet_bank_append_f32_at(ops._ptr(bank.blob), 2, 3, 9, 13, 0, ops._ptr(bank.n_dev), ops._ptr(cu(mk[:, :, 0])),
                                      ops._ptr(cu(mv[:, :, 0])), ops._ptr(cu(mr[:, 0])), ops._stream(dev()))
    assert rc == 0
    assert bank.overflow_count() & (1 << 30)              # ... and says so
    bank2 = _fill_bank(ops, mk, mv, mr)
    bank2.n_dev.fill_(7)
    bank2.read(1, cu(qk), cu(qv), cu(qr), _t_dev=bank2.n_dev)   # 1 + 7 frames of a 3- (or 4-) slot bank: clamped, flagged
    assert bank2.overflow_count() & (1 << 30)
    with pytest.raises(RuntimeError):
        bank.n_dev.fill_(1)
        bank.assert_synced()


@pytest.mark.parametrize('precision', ['f16', 'split'])
def test_whole_loop_grows_the_memory_to_five_frames(precision, oracle_mod):
    """Both arithmetic modes (f16 = the frame loop's default, split = fp32-class), 480x854, N = 6 with memorize_every = 1:
    the memory read by the last frame holds T = 5 frames (BASELINE configs[1]'s memory length), fused loop against the CPU path."""
    from rmnet_amd.synthetic import synthetic_clip
    prod, ref = _nets(oracle_mod, read_precision=precision)
    prod.fuse_epilogues()
    frames, masks, flows, n_objects = synthetic_clip(6, 2, 480, 854, seed=11, size=2.1)
    with torch.no_grad():
        est_cpu = ref(frames, masks, flows, n_objects, 1)
        est = prod(frames, masks, flows, n_objects, 1).cpu()
    assert float((est - est_cpu).abs().max()) < 1e-3
    lab, lab_cpu = est.argmax(2).numpy(), est_cpu.argmax(2).numpy()
    assert oracle_mod.iou(lab[:, 1:] == 1, lab_cpu[:, 1:] == 1) >= 0.999


# ----------------------------------------------------------------------------- round 5: parity that can fail
# (tests/live_fixture.py: one-object clips whose estimated masks have live boundaries; the exact-query arithmetic)
import live_fixture as lf   # noqa: E402  (tests/ is on sys.path under pytest's rootdir conftest)

_CPU_PATH_CACHE = {}


def _cpu_path(oracle_mod, ref, key, frames, masks, flows, n_objects, every):
    """The CPU path's (probabilities, logits) of a clip, once per session (several tests / parameters compare against it)."""
    if key not in _CPU_PATH_CACHE:
        threads = torch.get_num_threads()
        torch.set_num_threads(min(16, threads))       # (fastest count on the GPU box: bench.py cpu_baseline.sweep)
        oracle_mod.set_num_threads(min(16, threads))
        try:
            with torch.no_grad():
                _CPU_PATH_CACHE[key] = ref(frames, masks, flows, n_objects, every, return_logits=True)
        finally:
            torch.set_num_threads(threads)
            oracle_mod.set_num_threads(threads)
    return _CPU_PATH_CACHE[key]


def _table(line):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'live_iou_table.txt'), 'a') as fh:
            fh.write(line + '\n')


# Bar on the foreground LOGIT of a live pixel (|logit| < 10 on both sides), GPU loop vs CPU path.  Where it comes from: a label
# flips when the logit error exceeds the pixel's distance from 0; on these fixtures 1-2 % of the pixels lie within 0.4 logit units
# of the threshold around a foreground of ~20 % of the frame, so an IoU loss of 1e-3 corresponds to a uniform logit error of about
# 2e-2.  Measured (profiles/r05_iou_calibration.md): exact-fp32 GPU loop vs CPU path <= 2e-3 (MIOpen vs CPU convolutions), the
# arithmetic modes <= 6e-3.
LIVE_LOGIT_BAR = 2e-2


@pytest.mark.parametrize('name,mode', [(n, m) for n in ('live480-a', 'live480-b') for m in ('auto', 'auto-cl', 'exact', 'split', 'qx', 'f16')] + [('live480-c', m) for m in ('auto', 'auto-cl', 'exact', 'f16')] + [('live720', 'auto-cl')] +
                         [('live720', 'auto'), ('live720', 'exact'), ('live720', 'qx')])      # (720x1280: the resolution of configs[3] / [4])
def test_live_boundary_clips_meet_the_bar_in_every_arithmetic(name, mode, oracle_mod):
    """The north star's bar -- mask IoU within 1e-3 of the CPU path -- on one-object 480x854 (and 720x1280) clips whose masks HAVE a boundary
    (cover 10-60 %, >= 1 % of the pixels within 0.1 of the threshold on every frame: asserted), for the GPU loop with the
    exact-fp32 read, the three bank arithmetics and the default.  Also compared: the logits (LIVE_LOGIT_BAR) and the
    probabilities.  That this comparison CAN fail is test_mutated_memory_read_fails_the_parity_metric."""
    prod, ref = _nets(oracle_mod, 'auto' if mode in ('exact', 'auto-cl') else mode)
    frames, masks, flows, n_objects, every, delta = lf.make_clip(name)
    lf.shift_foreground_bias(prod, delta)
    lf.shift_foreground_bias(ref, delta)
    prod.fuse_epilogues()
    if mode == 'auto-cl':          # [r6] bench.py's configuration: both networks channels_last (MIOpen NHWC kernels), channels-last glue kernels
        prod = prod.channels_last()
    est_cpu, log_cpu = _cpu_path(oracle_mod, ref, name, frames, masks, flows, n_objects, every)
    lf.assert_live(est_cpu, name)
    with torch.no_grad():
        est, logits = prod(frames, masks, flows, n_objects, every, _exact=(mode == 'exact'), return_logits=True)
    est, logits = est.cpu(), logits.cpu()
    iou, gap, dp = lf.label_iou(est, est_cpu), lf.logit_gap(logits, log_cpu), float((est - est_cpu).abs().max())
    _table('%s %-5s (auto -> %s): IoU %.5f  max live fg-logit diff %.2e  max prob diff %.2e  cover/near %s' % (
        name, mode, prod.resolve_read_precision([1]), iou, gap, dp, ' '.join('%.3f/%.3f' % cn for cn in lf.liveness(est_cpu))))
    assert iou >= 0.999, (name, mode, iou)
    assert gap <= LIVE_LOGIT_BAR, (name, mode, gap)
    assert dp <= LIVE_LOGIT_BAR / 4 * 1.5, (name, mode, dp)      # (d p = p (1 - p) d logit <= d logit / 4; x1.5: clamped pixels)


class _MaxLogitReader:
    """MemoryReader.forward (models/rmnet.py:147-165) in torch that also keeps the largest affinity logit it has seen."""

    def __init__(self):
        self.smax = 0.0

    def __call__(self, m_key, m_val, q_key, q_val):
        no, De, T, h, w = m_key.shape
        S = torch.bmm(m_key.reshape(no, De, -1).transpose(1, 2), q_key.reshape(no, De, -1)) / (De ** 0.5)
        self.smax = max(self.smax, float(S.max()))
        mem = torch.bmm(m_val.reshape(no, -1, T * h * w), torch.softmax(S, dim=1)).reshape(no, -1, h, w)
        return torch.cat([mem, q_val], dim=1), None


_TEMPERATURE_SEEN = {}


@pytest.mark.parametrize('mode,scale', [(m, s_) for m in ('auto', 'split') for s_ in (1.0, 2.0, 4.0, 8.0)] +
                         [(m, s_) for m in ('qx', 'f16') for s_ in (1.0, 4.0)])      # (all 16 points: tools/iou_temperature.py --gpu)
def test_key_temperature_sweep(mode, scale, oracle_mod):
    """Round-5 verdict item 3: the arithmetic modes priced where the affinity soft-max is PEAKED, as a trained network's is -- the
    procedural weights give logits of O(1), a near-uniform soft-max, the friendliest case for fp16 operands.  The key convolutions of
    both KeyValue heads are scaled by s (every logit by s^2: max |S| 9 -> 36 -> 146 -> 583, mean top-1 mass 0.001 -> 0.04 -> 0.53 ->
    0.84 on this clip) and the decoder bias re-chosen per point so that the one-object 480x854 clip stays live (asserted).  Asserted:
      * 'split' (fp32-class) and 'auto' are within 5e-4 of the CPU path's masks at EVERY temperature;
      * 'auto' stays in 'f16' at s = 1 (the largest logit the bank measured is below rmnet.AUTO_LOGIT_BOUND) and re-reads the clip in
        'split' from s = 2 on -- the decision is made from what the read kernel measured on the clip, not from the object count;
      * the bank's logit word agrees with the CPU path's largest logit (a lower bound within 8: the kernel's reference is deferred);
      * 'f16' / 'qx' are recorded (gpurun_out/live_iou_table.txt -> profiles/r06_iou_temperature.md); at s = 1 they meet the 0.999 bar,
        and wherever they fall below 0.9995 'auto' must not have used them."""
    from rmnet_amd import rmnet as rmnet_mod
    prod, ref = _nets(oracle_mod, mode)
    name = 'live480-a'
    frames, masks, flows, n_objects, every, _ = lf.make_clip(name)
    delta = lf.TEMPERATURE_POINTS[name][scale]
    for net in (prod, ref):
        lf.shift_foreground_bias(lf.scale_keys(net, scale), delta)
    prod.fuse_epilogues()
    key = 'temp-%s-%g' % (name, scale)
    if key not in _CPU_PATH_CACHE:
        ref.reader = _MaxLogitReader()
        _cpu_path(oracle_mod, ref, key, frames, masks, flows, n_objects, every)
        _TEMPERATURE_SEEN[key] = ref.reader.smax
    est_cpu, log_cpu = _CPU_PATH_CACHE[key]
    smax_cpu = _TEMPERATURE_SEEN[key]
    lf.assert_live(est_cpu, key)
    with torch.no_grad():
        est, logits = prod(frames, masks, flows, n_objects, every, return_logits=True)
    est, logits = est.cpu(), logits.cpu()
    info = prod.last_clip
    iou, gap = lf.label_iou(est, est_cpu), lf.logit_gap(logits, log_cpu)
    _table('temperature s=%g (largest logit: CPU path %.1f, bank word %.1f) %-5s -> read %s%s: IoU %.5f  max live fg-logit diff %.2e' % (
        scale, smax_cpu, info['logit_max'], mode, info['read_precision'], ' (re-read: %s)' % info['reread'] if info['reread'] else '', iou, gap))
    # the measured logit: a lower bound of the CPU path's largest logit, within the deferral (8) + what MIOpen-vs-CPU keys differ by
    assert smax_cpu - 8.5 <= info['logit_max'] <= smax_cpu * 1.01 + 0.1, (scale, smax_cpu, info)
    if mode in ('auto', 'split'):
        assert iou >= 0.9995, (mode, scale, iou)
    if mode == 'auto':
        if scale == 1.0:
            assert info['read_precision'] == 'f16' and info['reread'] is None and info['logit_max'] <= rmnet_mod.AUTO_LOGIT_BOUND
        else:
            assert info['reread'] is not None and info['reread'].startswith('split') and info['logit_max'] > rmnet_mod.AUTO_LOGIT_BOUND
    if mode in ('f16', 'qx') and scale == 1.0:
        assert iou >= 0.999, (mode, iou)


def test_auto_rule_is_sticky_across_clips_of_one_network():
    """'auto' on a network whose soft-max is peaked (keys x 4): the FIRST one-object clip starts in 'f16', measures logits beyond
    rmnet.AUTO_LOGIT_BOUND and is re-read in 'split'; the SECOND clip of the same network starts in 'split' -- it is not processed twice
    -- and gives the masks of the first clip's re-read; ``reset_auto_precision()`` returns to the optimistic start."""
    from rmnet_amd import networks, rmnet as rmnet_mod
    from rmnet_amd.rmnet import RMNet
    name, scale = 'live240', 4.0
    prod = networks.procedural_init_(RMNet(None)).to(dev()).eval()
    lf.shift_foreground_bias(lf.scale_keys(prod, scale), lf.TEMPERATURE_POINTS[name][scale])
    prod.fuse_epilogues()
    frames, masks, flows, n_objects, every, _ = lf.make_clip(name)
    with torch.no_grad():
        first = prod(frames, masks, flows, n_objects, every).cpu()
        info1 = dict(prod.last_clip)
        second = prod(frames, masks, flows, n_objects, every).cpu()
        info2 = dict(prod.last_clip)
    assert info1['reread'] is not None and info1['reread'].startswith('split') and info1['logit_max'] > rmnet_mod.AUTO_LOGIT_BOUND, info1
    assert info2['read_precision'] == 'split' and info2['reread'] is None, info2
    assert prod.resolve_read_precision([1]) == 'split'
    assert float((first - second).abs().max()) < 1e-4          # two 'split' runs of the same clip
    prod.reset_auto_precision()
    assert prod.resolve_read_precision([1]) == 'f16'


@pytest.mark.parametrize('clip,mutation', [('live480-b', 'zero'), ('live480-b', 'noise-1pct'), ('3obj-480p', 'noise-1pct')])
def test_mutated_memory_read_fails_the_parity_metric(clip, mutation, oracle_mod, monkeypatch):
    """Mutation check of the parity metric: the SAME comparison as the parity tests (GPU loop vs CPU path, label IoU >= 0.999
    per object) with the memory half of every read-out of the GPU loop zeroed / noised by 1 % of its standard deviation must
    FAIL -- and with the read-out left alone it passes.  On a one-object live-boundary clip (the fixture with the softest
    boundary: CPU emulation of the 1 % noise gives 0.9970) and on the 3-object calibration clip (boundaries between objects).
    (On the saturated one-object clips of rounds 1-4 the zeroed read-out left the IoU at 0.99985: those assertions could not
    see the read.)"""
    from rmnet_amd import ops
    from rmnet_amd.synthetic import synthetic_clip
    prod, ref = _nets(oracle_mod, 'split')
    if clip in lf.LIVE_CLIPS:
        frames, masks, flows, n_objects, every, delta = lf.make_clip(clip)
        lf.shift_foreground_bias(prod, delta)
        lf.shift_foreground_bias(ref, delta)
        n_obj, key = 1, clip
    else:
        n_obj, H, W, every, seed, size, N = _CALIB_CASES[clip]
        frames, masks, flows, n_objects = synthetic_clip(N, n_obj + 1, H, W, seed=seed, size=size)
        key = 'calib-' + clip
    prod.fuse_epilogues()
    est_cpu, _ = _cpu_path(oracle_mod, ref, key, frames, masks, flows, n_objects, every)
    worst = lambda est: min(lf.label_iou(est, est_cpu, k) for k in range(1, n_obj + 1))
    with torch.no_grad():
        clean = prod(frames, masks, flows, n_objects, every).cpu()
    assert worst(clean) >= 0.999
    orig = ops.MemoryBank.read_staged
    gen = torch.Generator(device=dev()).manual_seed(5)

    def mutated(self, *a, **k):
        out = orig(self, *a, **k)
        mem = out[:, :512]
        if mutation == 'zero':
            mem.zero_()
        else:
            mem += 0.01 * mem.std() * torch.randn(mem.shape, generator=gen, device=mem.device)
        return out
    monkeypatch.setattr(ops.MemoryBank, 'read_staged', mutated)
    with torch.no_grad():
        bad = prod(frames, masks, flows, n_objects, every).cpu()
    iou = worst(bad)
    _table('%s mutation %s: IoU %.5f' % (clip, mutation, iou))
    assert iou < 0.999, (clip, mutation, iou)
    if mutation == 'zero':
        assert iou < 0.9, iou


@pytest.mark.parametrize('no,T,h,w,regional', [
    (1, 1, 4, 5, False), (2, 3, 9, 13, True), (1, 5, 30, 54, True), (1, 7, 16, 24, False), (2, 9, 10, 7, True),
    (14, 2, 6, 9, True), (70, 1, 4, 5, True), (5, 3, 30, 54, True), (5, 5, 30, 54, True), (3, 20, 12, 20, True),
    (1, 70, 5, 6, True), (50, 2, 12, 20, False), (24, 3, 12, 20, True),
    (64, 2, 16, 24, False), (60, 3, 16, 24, True), (20, 1, 30, 54, False)])      # [r6] more pairs than workgroups: rounds
def test_bank_read_qx_mode_vs_oracle(no, T, h, w, regional, oracle_mod):
    """The fp16-operand arithmetic with an exact query (RMNET_BANK_QX) on the cases of test_bank_read_vs_oracle: the same bars as
    the fp16-operand mode (K, P and V are still rounded to fp16), and never a larger mean error than that mode on the same bank."""
    from rmnet_amd import ops
    rng = np.random.RandomState(no * 1000 + T * 100 + h + 1)
    mk, mv, qk, qv, mr, qr = _random_case(rng, no, T, h, w, regional=regional)
    bank = ops.MemoryBank(no, T + 2, h, w, dev(), precision='qx')
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), None if mr is None else cu(mr[:, t]))
    if regional:
        want, _ = oracle_mod.regional_memory_read(mk, mv, qk, qv, mr, qr)
        got = bank.read(T, cu(qk), cu(qv), cu(qr)).cpu().numpy()
    else:
        want, _ = oracle_mod.memory_read(mk, mv, qk, qv)
        got = bank.read(T, cu(qk), cu(qv)).cpu().numpy()
    _f16_bars(got, want, float(np.abs(mv).max()))
    assert bank.overflow_count() == 0
    bank.precision = 'f16'
    got16 = bank.read(T, cu(qk), cu(qv), cu(qr) if regional else None).cpu().numpy()
    assert float(np.abs(got[:, :512] - want[:, :512]).mean()) <= float(np.abs(got16[:, :512] - want[:, :512]).mean()) * 1.05 + 1e-7


def test_qx_mode_on_large_logits_and_flag_rules(golden_dir, oracle_mod):
    """RMNET_MR_QX through the drop-in entry on logits in the tens -- the reference's golden 'peaky' vectors (logits to 32) and a
    x9 spike (logit ~36): inside the fp16-operand bars and not worse than RMNET_MR_F16 (on ONE read of random data the two are
    close: P's and V's roundings dominate; what the exact query buys is measured on whole clips, where q's error is fed back --
    test_iou_bar_against_the_cpu_path_on_long_clips, profiles/r05_iou_calibration.md).  The two flags are mutually exclusive;
    out-of-window values fall back to the exact kernel."""
    from rmnet_amd import ops
    g = np.load(os.path.join(golden_dir, 'memory_reader.npz'))
    checked = 0
    for name in sorted({k.split('.')[0] for k in g.files}):
        if name + '.m_key' not in g.files or name + '.mem_val' not in g.files:
            continue
        mk, mv, qk, qv = (g[name + '.' + k].astype(np.float32) for k in ('m_key', 'm_val', 'q_key', 'q_val'))
        if mk.shape[1] != 128 or mv.shape[1] != 512:
            continue
        got, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), flags=ops.MR_QX)
        no_, _, T_, h_, w_ = mk.shape
        smax = float(np.abs(np.einsum('ocj,oci->oji', mk.reshape(no_, 128, -1), qk.reshape(no_, 128, -1))).max()) / np.sqrt(128.0)
        _f16_bars(got.cpu().numpy(), g[name + '.mem_val'].astype(np.float32), float(np.abs(mv).max()), smax)
        checked += 1
    assert checked >= 1
    rng = np.random.RandomState(5)
    no, T, h, w = 2, 4, 8, 16
    mk, mv, qk, qv, _, _ = _random_case(rng, no, T, h, w, regional=False)
    mk[:, :, 3, h - 1, w - 2] = qk[:, :, 2, 3] * 9.0
    want, _ = oracle_mod.memory_read(mk, mv, qk, qv)
    got_x, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), flags=ops.MR_QX)
    got_h, _ = ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), flags=ops.MR_F16)
    ex = np.abs(got_x.cpu().numpy()[:, :512] - want[:, :512])
    eh = np.abs(got_h.cpu().numpy()[:, :512] - want[:, :512])
    _f16_bars(got_x.cpu().numpy(), want, float(np.abs(mv).max()), smax=36.0)
    assert float(ex.mean()) <= 1.02 * float(eh.mean()), (float(ex.mean()), float(eh.mean()))   # (never worse; where it pays is the whole clip)
    with pytest.raises(RuntimeError):
        ops.memory_read(cu(mk), cu(mv), cu(qk), cu(qv), flags=ops.MR_QX | ops.MR_F16)
    # the bank entry with otherwise VALID arguments: an unknown bit and F16 | QX are refused, each known flag alone is served
    # (round-5 advisor: tests/test_capi.py can only make these calls with NULL pointers, which are refused whatever the flags)
    bank = ops.MemoryBank(no, T, h, w, dev())
    for t in range(T):
        bank.append(t, cu(mk[:, :, t]), cu(mv[:, :, t]), None)
    for flags, ok in ((0, True), (ops.BANK_F16, True), (ops.BANK_QX, True), (16, False), (ops.BANK_F16 | ops.BANK_QX, False), (ops.BANK_QX | 16, False)):
        saved = dict(ops._PRECISION_FLAGS)
        try:
            ops._PRECISION_FLAGS['split'] = flags
            if ok:
                got_b = bank.read(T, cu(qk), cu(qv))
                assert np.abs(got_b.cpu().numpy()[:, :512] - want[:, :512]).max() < 0.2
            else:
                with pytest.raises(RuntimeError, match='invalid'):
                    bank.read(T, cu(qk), cu(qv))
        finally:
            ops._PRECISION_FLAGS.clear()
            ops._PRECISION_FLAGS.update(saved)
    mv2 = mv.copy()
    mv2[1, 7, 2, 3, 4] = 5000.0
    want2, _ = oracle_mod.memory_read(mk, mv2, qk, qv)
    got2, _ = ops.memory_read(cu(mk), cu(mv2), cu(qk), cu(qv), flags=ops.MR_QX)
    np.testing.assert_allclose(got2.cpu().numpy(), want2, atol=MR_ATOL * 50, rtol=MR_RTOL)


def test_graph_replay_is_refused_for_banks_longer_than_one_launch(oracle_mod, monkeypatch):
    """Round-4 advisor: a bank of more than BANK_MAX_SLOTS frames is read in host-planned chunks, so its frame count would be
    baked into a captured frame step.  forward(graph=True) must run such a clip eagerly (same masks as graph=False), and
    read_staged must refuse to be captured.  (BANK_MAX_SLOTS is lowered to 3 for the test: the kernels' own limit stays 2048.)"""
    from rmnet_amd import networks, ops
    from rmnet_amd.rmnet import RMNet
    from rmnet_amd.synthetic import synthetic_clip
    monkeypatch.setattr(ops, 'BANK_MAX_SLOTS', 3)
    net = networks.procedural_init_(RMNet(None)).to(dev()).eval()
    net.fuse_epilogues()
    frames, masks, flows, n_objects = synthetic_clip(8, 2, 96, 160, seed=3)
    with torch.no_grad():
        est_g = net(frames, masks, flows, n_objects, 1, graph=True).cpu()     # capacity 7 > 3: must not capture
        est_e = net(frames, masks, flows, n_objects, 1, graph=False).cpu()
    assert float((est_g - est_e).abs().max()) < 1e-3            # (two eager runs: MIOpen may pick another algorithm)
    assert (est_g.argmax(2) == est_e.argmax(2)).float().mean() > 0.999
    bank = ops.MemoryBank(1, 8, 6, 10, dev())
    k4, v4 = torch.randn(1, 128, 6, 10, device=dev()), torch.randn(1, 512, 6, 10, device=dev())
    bank.stage(k4, v4, None)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        bank.read_staged(k4, v4)
    torch.cuda.current_stream().wait_stream(side)
    with pytest.raises(RuntimeError, match='chunked'):
        with torch.cuda.graph(g):
            bank.read_staged(k4, v4)
    torch.cuda.synchronize()
    assert ops.TensorBank(1, 2, 6, 10, dev()).timeout_count() == 0


def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` from a plain shell (no torch.distributed.run around it, WORLD_SIZE unset): bench.py becomes the
    launcher of its two ranks -- here over gloo, both on cuda:0 -- and rank 0's single JSON line comes back."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--steps', '2',
                          '--warmup', '1', '--clips-per-gpu', '1', '--no-cpu-baseline', '--no-extras', '--no-miopen-find'],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['multi_gpu']['backend'] == 'gloo' and line['value'] > 0
