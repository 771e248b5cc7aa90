# -*- coding: utf-8 -*-
"""oracle/oracle.py -- CPU checker for the RMNet hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing under ``rmnet_amd/`` does (tests/test_host_logic.py greps for it).

Two layers:

* thin ctypes wrappers over ``liboracle.so`` (built from ``rmnet_oracle.c`` by
  ``oracle/Makefile``): the C restatements of the reference's two native ops, of
  ``MemoryReader.forward`` and of the regional K/V masking;
* ``OracleRMNet`` -- a plain-torch, CPU restatement of the reference's per-frame loop
  (``models/rmnet.py:191-452``: pad_memory / memorize / warp / get_att_map / soft_aggregation /
  segment / forward) in the reference's own data layout (``[B, K, C, T, h, w]`` memory grown
  with ``torch.cat``, K/V multiplied by full 0/1 maps, dense MemoryReader).  The product
  (``rmnet_amd.rmnet``) computes the same function with a pre-allocated bank and the fused
  regional HIP kernel, so agreement between the two is a real cross-check.

Pinning of this oracle against the reference itself: ``tests/golden/make_golden.py`` imports
``/root/reference`` (in the build container only) and stores its outputs; tests/test_oracle.py
compares every function here with those fixtures.
"""

import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'rmnet_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a, ty=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


# ----------------------------------------------------------------------------- native ops
def region_map(mask, prob_threshold=0.5, n_pts_threshold=10, n_bbox_loose_pixels=64):
    """mask [B,K,H,W] f32 -> (att_map [B,K,H,W] f32, bboxes [B,K,4] int32)."""
    mask = _f32(mask)
    B, K, H, W = mask.shape
    att = np.empty_like(mask)
    bb = np.empty((B, K, 4), dtype=np.int32)
    lib().oracle_region_map_f32(_p(mask), B, K, H, W, ctypes.c_float(prob_threshold),
                                int(n_pts_threshold), int(n_bbox_loose_pixels), _p(att),
                                _p(bb, ctypes.c_int32))
    return att, bb


def cell_rects(bboxes, lw, lh, h, w, stride=16):
    """bboxes [B,K,4] int32 (pixel, in the un-padded frame when lw/lh != 0) -> [B,K,4] cell rects."""
    bb = np.ascontiguousarray(bboxes, dtype=np.int32)
    B, K, _ = bb.shape
    out = np.empty_like(bb)
    lib().oracle_cell_rects_i32(_p(bb, ctypes.c_int32), B, K, int(lw), int(lh), int(stride), int(h),
                                int(w), _p(out, ctypes.c_int32))
    return out


def rect_mask(x, rects):
    """x [n,C,T,h,w] f32, rects [n,T,4] -> x * (0/1 rectangle map)."""
    x = _f32(x)
    n, C, T, h, w = x.shape
    r = np.ascontiguousarray(rects, dtype=np.int32).reshape(n, T, 4)
    y = np.empty_like(x)
    lib().oracle_rect_mask_f32(_p(x), n, C, T, h, w, _p(r, ctypes.c_int32), _p(y))
    return y


def memory_read(m_key, m_val, q_key, q_val, want_p=False):
    """Dense MemoryReader.forward on CPU (C, OpenMP).  Returns (mem_val, p or None)."""
    m_key, m_val, q_key, q_val = map(_f32, (m_key, m_val, q_key, q_val))
    no, De, T, h, w = m_key.shape
    Do = m_val.shape[1]
    out = np.empty((no, 2 * Do, h, w), dtype=np.float32)
    p = np.empty((no, T * h * w, h * w), dtype=np.float32) if want_p else None
    lib().oracle_memory_read_f32(_p(m_key), _p(m_val), _p(q_key), _p(q_val), no, De, Do, T, h, w,
                                 _p(out), _p(p) if want_p else None)
    return out, p


def regional_memory_read(m_key, m_val, q_key, q_val, mem_rects, qry_rects, want_p=False):
    """What the fused regional kernel must equal: mask K/V with the rectangles
    (models/rmnet.py:247-248, 357-358) and run the dense read (:361)."""
    no = m_key.shape[0]
    T = m_key.shape[2]
    mr = np.asarray(mem_rects, dtype=np.int32).reshape(no, T, 4)
    qr = np.asarray(qry_rects, dtype=np.int32).reshape(no, 1, 4)
    mk, mv = rect_mask(m_key, mr), rect_mask(m_val, mr)
    qk = rect_mask(np.asarray(q_key)[:, :, None], qr)[:, :, 0]
    qv = rect_mask(np.asarray(q_val)[:, :, None], qr)[:, :, 0]
    return memory_read(mk, mv, qk, qv, want_p)


def memory_read_sampled(m_key, m_val, q_key, qidx):
    """The read-out rows of ``memory_read`` for the query cells ``qidx`` (flat indices into h*w, the
    same list for every object) only: [no, len(qidx), Do].  Same arithmetic, element for element
    (rmnet_oracle.c); lets full-size cases (720p, T = 20) be checked in seconds."""
    m_key, m_val, q_key = map(_f32, (m_key, m_val, q_key))
    no, De, T, h, w = m_key.shape
    Do = m_val.shape[1]
    qi = np.ascontiguousarray(qidx, dtype=np.int32)
    out = np.empty((no, qi.size, Do), dtype=np.float32)
    lib().oracle_memory_read_sampled_f32(_p(m_key), _p(m_val), _p(q_key), no, De, Do, T, h, w,
                                         _p(qi, ctypes.c_int32), int(qi.size), _p(out))
    return out


def regional_memory_read_sampled(m_key, m_val, q_key, mem_rects, qry_rects, qidx):
    """``regional_memory_read`` on a sample of query cells (see ``memory_read_sampled``)."""
    no = m_key.shape[0]
    T = m_key.shape[2]
    mr = np.asarray(mem_rects, dtype=np.int32).reshape(no, T, 4)
    qr = np.asarray(qry_rects, dtype=np.int32).reshape(no, 1, 4)
    mk, mv = rect_mask(m_key, mr), rect_mask(m_val, mr)
    qk = rect_mask(np.asarray(q_key)[:, :, None], qr)[:, :, 0]
    return memory_read_sampled(mk, mv, qk, qidx)


def flow_affine(flow, m1, m2):
    """flow [H,W,2] f32, m1/m2 [2,3] f32 -> [H,W,2] f32."""
    flow, m1, m2 = _f32(flow), _f32(m1), _f32(m2)
    H, W, _ = flow.shape
    out = np.empty_like(flow)
    lib().oracle_flow_affine_f32(_p(flow), _p(m1), _p(m2), H, W, _p(out))
    return out


def iou(seg, ann):
    """Region similarity J (utils/metrics.py:84-102): 1 when both are empty."""
    seg, ann = np.asarray(seg).astype(bool), np.asarray(ann).astype(bool)
    if not ann.any() and not seg.any():
        return 1.0
    return float((seg & ann).sum()) / float((seg | ann).sum())


# ----------------------------------------------------------------------------- model restatement
def _pad16(tensors, size):
    h, w = int(size[0]), int(size[1])
    dh, dw = (16 - h % 16) % 16, (16 - w % 16) % 16
    pad = (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2)
    return [F.pad(t, pad) for t in tensors], pad


def torch_memory_read(m_key, m_val, q_key, q_val):
    """models/rmnet.py:147-165 with torch CPU ops (the 'port' timed as cpu_baseline)."""
    no, De, T, h, w = m_key.shape
    Do = m_val.shape[1]
    aff = torch.bmm(m_key.reshape(no, De, -1).transpose(1, 2), q_key.reshape(no, De, -1))
    aff = F.softmax(aff / math.sqrt(De), dim=1)
    mem = torch.bmm(m_val.reshape(no, Do, -1), aff).reshape(no, Do, h, w)
    return torch.cat([mem, q_val], dim=1), aff


class OracleRMNet(torch.nn.Module):
    """CPU restatement of ``RMNet`` (models/rmnet.py:179-452).  Sub-module names equal the
    reference's so one state dict loads into the reference, this oracle and the product."""

    def __init__(self, cfg=None, reader='torch'):
        super().__init__()
        from rmnet_amd import networks as nets  # pure-torch conv stacks only (no HIP involved)
        self.cfg = cfg
        self.encoder_memory = nets.EncoderMemory()
        self.encoder_query = nets.EncoderQuery()
        self.kv_memory = nets.KeyValue(1024, keydim=128, valdim=512)
        self.kv_query = nets.KeyValue(1024, keydim=128, valdim=512)
        self.decoder = nets.Decoder(256)
        self.reader = reader
        self.last = {}

    # models/rmnet.py:280-287 + extensions/reg_att_map_generator/__init__.py:31-33
    def get_att_map(self, prev_mask, flow=None):
        m = prev_mask if flow is None else self.warp(prev_mask, flow)[0]
        att, bb = region_map(m.detach().cpu().numpy())
        return torch.from_numpy(att), torch.from_numpy(bb)

    # models/rmnet.py:252-278
    @staticmethod
    def warp(img, flow):
        B, C, H, W = img.shape
        xs = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
        ys = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1).expand(B, 1, H, W)
        g = torch.cat((xs, ys), 1) + flow
        gx = 2.0 * g[:, 0] / max(W - 1, 1) - 1.0
        gy = 2.0 * g[:, 1] / max(H - 1, 1) - 1.0
        grid = torch.stack((gx, gy), dim=3)
        out = F.grid_sample(img, grid, align_corners=True)
        valid = F.grid_sample(torch.ones_like(img), grid, align_corners=True)
        valid = (valid >= 0.9999).to(img.dtype)
        return out * valid, valid

    # models/rmnet.py:191-205, 207-250
    def memorize(self, frame, masks, n_objects):
        B, K, H, W = masks.shape
        (frame, masks), _ = _pad16([frame, masks], (H, W))
        fs, ms, os_ = [], [], []
        for b in range(B):
            for o in range(1, n_objects[b] + 1):
                fs.append(frame[b:b + 1])
                ms.append(masks[b, o:o + 1])
                others = masks[b, 1:o].sum(0, keepdim=True) + masks[b, o + 1:n_objects[b] + 1].sum(0, keepdim=True)
                os_.append(others.clamp(0, 1))
        r4 = self.encoder_memory(torch.cat(fs), torch.cat(ms), torch.cat(os_))[0]
        k4, v4 = self.kv_memory(r4)
        h, w = k4.shape[-2:]
        pk = torch.zeros(B, K, k4.shape[1], 1, h, w)
        pv = torch.zeros(B, K, v4.shape[1], 1, h, w)
        at = 0
        for b in range(B):
            pk[b, 1:n_objects[b] + 1, :, 0] = k4[at:at + n_objects[b]]
            pv[b, 1:n_objects[b] + 1, :, 0] = v4[at:at + n_objects[b]]
            at += n_objects[b]
        att, bboxes = self.get_att_map(masks)
        att = F.interpolate(att, scale_factor=1 / 16)[:, :, None, None]
        return pk * att, pv * att, bboxes

    # models/rmnet.py:289-302
    @staticmethod
    def soft_aggregation(ps, K, n_objects):
        B = len(n_objects)
        em = torch.zeros(B, K, *ps.shape[1:])
        at = 0
        for b in range(B):
            sl = ps[at:at + n_objects[b]]
            em[b, 0] = torch.prod(1 - sl, dim=0)
            em[b, 1:n_objects[b] + 1] = sl
            at += n_objects[b]
        em = em.clamp(1e-7, 1 - 1e-7)
        return torch.log(em / (1 - em))

    # models/rmnet.py:304-383
    def segment(self, frame, att_map, keys, values, n_objects):
        B, K = keys.shape[:2]
        (frame, att_map), pad = _pad16([frame, att_map], frame.shape[2:])
        r4, r3, r2, _, _ = self.encoder_query(frame)
        k4, v4 = self.kv_query(r4)
        sel = lambda x, b: x[b:b + 1].expand(n_objects[b], -1, -1, -1)
        k4e = torch.cat([sel(k4, b) for b in range(B)])
        v4e = torch.cat([sel(v4, b) for b in range(B)])
        r3e = torch.cat([sel(r3, b) for b in range(B)])
        r2e = torch.cat([sel(r2, b) for b in range(B)])
        key = torch.cat([keys[b, 1:n_objects[b] + 1] for b in range(B)])
        val = torch.cat([values[b, 1:n_objects[b] + 1] for b in range(B)])
        att = torch.cat([att_map[b, 1:n_objects[b] + 1].unsqueeze(1) for b in range(B)])
        att = F.interpolate(att, scale_factor=1 / 16)
        k4e, v4e = k4e * att, v4e * att
        if callable(self.reader):        # a test's own restatement (tests/live_fixture.py: rounded / mutated readers)
            m4, _ = self.reader(key, val, k4e, v4e)
        elif self.reader == 'torch':
            m4, _ = torch_memory_read(key, val, k4e, v4e)
        else:
            m4 = torch.from_numpy(memory_read(key.numpy(), val.numpy(), k4e.numpy(), v4e.numpy())[0])
        self.last = {'m4': m4, 'key': key, 'val': val, 'k4e': k4e, 'v4e': v4e}
        ps = F.softmax(self.decoder(m4, r3e, r2e), dim=1)[:, 1]
        logit = self.soft_aggregation(ps, K, n_objects)
        lw, uw, lh, uh = pad
        return logit[:, :, lh:logit.shape[2] - uh, lw:logit.shape[3] - uw]

    # models/rmnet.py:385-452
    def forward(self, frames, masks, optical_flows, n_objects, memorize_every, device=None, return_logits=False):
        B, N, _, H, W = frames.shape
        K = masks.shape[2]
        est = torch.zeros(B, N, K, H, W)
        logits = torch.zeros(B, N, K, H, W)
        est[:, 0] = masks[:, 0].float()
        n_max = [int(n.max()) for n in n_objects]
        existing = [torch.unique(torch.argmax(masks[b, 0], dim=0)).tolist() for b in range(B)]
        commit = set(range(0, N, memorize_every))
        fresh = {j for j in range(1, N) if bool((n_objects[:, j] != n_objects[:, j - 1]).any())}
        keys = values = None
        for t in range(1, N):
            pk, pv, _ = self.memorize(frames[:, t - 1], est[:, t - 1], n_max)
            tk = pk if t == 1 else torch.cat([keys, pk], dim=3)
            tv = pv if t == 1 else torch.cat([values, pv], dim=3)
            if (t - 1) in commit or (t - 1) in fresh:
                keys, values = tk, tv
            att, _ = self.get_att_map(est[:, t - 1], optical_flows[:, t])
            logit = self.segment(frames[:, t], att, tk, tv, n_max)
            if t in fresh:
                for b in range(B):
                    for j in torch.unique(torch.argmax(masks[b, t], dim=0)).tolist():
                        if j not in existing[b]:
                            existing[b].append(j)
                            logit[b, j] = masks[b, t, j].float() * 32.0605 - 16.1181
            for b in range(B):
                for j in range(n_max[b] + 1):
                    if j not in existing[b]:
                        logit[b, j] = -16.1181
            est[:, t] = F.softmax(logit, dim=1)
            logits[:, t] = logit
        return (est, logits) if return_logits else est
