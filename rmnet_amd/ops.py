# -*- coding: utf-8 -*-
"""Torch-facing wrappers over the C ABI.  PyTorch is plumbing here (device memory, current stream,
current device); all arithmetic happens in librmnet_hip.so.

Input validation mirrors the reference's pybind layer
(extensions/reg_att_map_generator/reg_att_map_generator_cuda.cpp:14-19: CUDA + contiguous, else
RuntimeError) and additionally checks dtype, which the reference only assumes.
"""

import ctypes

import numpy as np
import torch

from . import _lib


MR_FORCE_GENERIC = 1     # include/rmnet_hip.h RMNET_MR_*
MR_EXACT_FP32 = 2
BANK_MAX_SLOTS = 2048    # csrc/bank.hip, csrc/memory_read.hip kMaxT: frames per LAUNCH (longer banks are read in chunks)


def _check(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must be a CUDA tensor' % name)      # CHECK_CUDA
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)          # CHECK_CONTIGUOUS
    if t.dtype != dtype:
        raise RuntimeError('%s must be %s, got %s' % (name, dtype, t.dtype))


def _is_cl(t):
    """True for a 4-D tensor that is channels-last in memory ([N,H,W,C]) and NOT also plain-contiguous (C == 1 or H == W == 1 are both)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def _check_act(t, name):
    """An activation of the glue kernels: fp32, CUDA, NCHW-contiguous or channels-last."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a CUDA tensor' % name)
    if t.dtype != torch.float32:
        raise RuntimeError('%s must be torch.float32, got %s' % (name, t.dtype))
    if not (t.is_contiguous() or _is_cl(t)):
        raise RuntimeError('%s must be contiguous (NCHW) or channels-last' % name)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _ws(nbytes, dev):
    # the caching allocator makes this a pointer bump; stream-ordered like any torch temporary
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def region_map(mask, prob_threshold=0.5, n_pts_threshold=10, n_bbox_loose_pixels=64,
               want_map=True, cell_grid=None, flow=None, want_warped=False):
    """mask [B,K,H,W] f32 cuda -> (att_map [B,K,H,W] f32 | None, bboxes [B,K,4] i32, rects | None).

    ``cell_grid = (pad_l, pad_t, stride, cells_h, cells_w)`` additionally returns the boxes as cell
    rectangles on the 1/stride feature grid (see include/rmnet_hip.h).
    ``flow`` [B,2,H,W]: the boxes are those of the flow-warped mask (RMNet.get_att_map with a flow,
    models/rmnet.py:252-287), computed without materialising it; with ``want_warped`` the warped mask
    (channels 1..K-1; channel 0 is zero) is returned as a fourth value."""
    _check(mask, 'mask')
    if mask.dim() != 4:
        raise RuntimeError('mask must be [B, K, H, W]')
    if flow is not None:
        return _region_map_warped(mask, flow, prob_threshold, n_pts_threshold, n_bbox_loose_pixels,
                                  want_map, cell_grid, want_warped)
    if want_warped:
        raise RuntimeError('want_warped needs a flow')
    lib = _lib.load()
    B, K, H, W = mask.shape
    dev = mask.device
    with torch.cuda.device(dev):
        att = torch.empty_like(mask) if want_map else None
        bboxes = torch.empty(B, K, 4, dtype=torch.int32, device=dev)
        rects = torch.empty(B, K, 4, dtype=torch.int32, device=dev) if cell_grid is not None else None
        pl, pt, st, ch, cw = cell_grid if cell_grid is not None else (0, 0, 16, 1, 1)
        nb = lib.rmnet_region_map_workspace_bytes(B, K, H, W)
        ws = _ws(nb, dev)
        rc = lib.rmnet_region_map_f32(_ptr(mask), B, K, H, W, float(prob_threshold),
                                      int(n_pts_threshold), int(n_bbox_loose_pixels), _ptr(att),
                                      _ptr(bboxes), _ptr(rects), int(pl), int(pt), int(st), int(ch),
                                      int(cw), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(rc, 'rmnet_region_map_f32')
    return att, bboxes, rects


def _region_map_warped(mask, flow, prob_threshold, n_pts_threshold, n_bbox_loose_pixels, want_map,
                       cell_grid, want_warped):
    _check(flow, 'flow')
    B, K, H, W = mask.shape
    if tuple(flow.shape) != (B, 2, H, W):
        raise RuntimeError('flow must be [B, 2, H, W]')
    lib = _lib.load()
    dev = mask.device
    with torch.cuda.device(dev):
        att = torch.empty_like(mask) if want_map else None
        bboxes = torch.empty(B, K, 4, dtype=torch.int32, device=dev)
        rects = torch.empty(B, K, 4, dtype=torch.int32, device=dev) if cell_grid is not None else None
        warped = torch.zeros_like(mask) if want_warped else None
        pl, pt, st, ch, cw = cell_grid if cell_grid is not None else (0, 0, 16, 1, 1)
        nb = lib.rmnet_region_map_workspace_bytes(B, K, H, W)
        ws = _ws(nb, dev)
        rc = lib.rmnet_region_map_warped_f32(_ptr(mask), _ptr(flow), B, K, H, W, float(prob_threshold),
                                             int(n_pts_threshold), int(n_bbox_loose_pixels), _ptr(att),
                                             _ptr(bboxes), _ptr(rects), int(pl), int(pt), int(st),
                                             int(ch), int(cw), _ptr(warped), _ptr(ws), ws.numel(),
                                             _stream(dev))
    _lib.check(rc, 'rmnet_region_map_warped_f32')
    return (att, bboxes, rects, warped) if want_warped else (att, bboxes, rects)


def boxes_to_cell_rects(bboxes, pad_l, pad_t, stride, cells_h, cells_w, k_per_batch=0):
    """bboxes [..., 4] i32 cuda -> cell rectangles, same shape.  ``k_per_batch`` > 0 marks every
    k-th box (channel 0) as empty."""
    _check(bboxes, 'bboxes', torch.int32)
    lib = _lib.load()
    dev = bboxes.device
    out = torch.empty_like(bboxes)
    with torch.cuda.device(dev):
        rc = lib.rmnet_boxes_to_cell_rects_i32(_ptr(bboxes), bboxes.numel() // 4, int(k_per_batch),
                                               int(pad_l), int(pad_t), int(stride), int(cells_h),
                                               int(cells_w), _ptr(out), _stream(dev))
    _lib.check(rc, 'rmnet_boxes_to_cell_rects_i32')
    return out


def memory_read(m_key, m_val, q_key, q_val, mem_rects=None, qry_rects=None, want_p=False, flags=0,
                T=None, out=None, events=None):
    """Fused (regional) memory read.

    m_key [no,De,Tcap,h,w], m_val [no,Do,Tcap,h,w] (only the first ``T`` frames are read; default
    all), q_key [no,De,h,w], q_val [no,Do,h,w]; optional mem_rects [no,T,4] / qry_rects [no,4] i32.
    ``events`` = (ev_start, ev_mid, ev_end) raw hipEvent_t handles (ints) recorded around the two
    kernels of the fast path (profiling only).
    Returns (mem_val [no,2*Do,h,w], p [no,T*h*w,h*w] | None)."""
    for t, n in ((m_key, 'm_key'), (m_val, 'm_val'), (q_key, 'q_key'), (q_val, 'q_val')):
        _check(t, n)
    if m_key.dim() != 5 or m_val.dim() != 5 or q_key.dim() != 4 or q_val.dim() != 4:
        raise RuntimeError('expected m_key/m_val [no,C,T,h,w] and q_key/q_val [no,C,h,w]')
    no, De, Tcap, h, w = m_key.shape
    Do = m_val.shape[1]
    if (m_val.shape[0], m_val.shape[2], m_val.shape[3], m_val.shape[4]) != (no, Tcap, h, w) or \
            tuple(q_key.shape) != (no, De, h, w) or tuple(q_val.shape) != (no, Do, h, w):
        raise RuntimeError('memory/query shapes do not agree')
    T = Tcap if T is None else int(T)
    if not 1 <= T <= Tcap:
        raise RuntimeError('T must be in [1, %d]' % Tcap)
    if (mem_rects is None) != (qry_rects is None):
        raise RuntimeError('mem_rects and qry_rects must be given together')
    if mem_rects is not None:
        _check(mem_rects, 'mem_rects', torch.int32)
        _check(qry_rects, 'qry_rects', torch.int32)
        if mem_rects.numel() != no * T * 4 or qry_rects.numel() != no * 4:
            raise RuntimeError('mem_rects must be [no,T,4] and qry_rects [no,4]')
    lib = _lib.load()
    dev = m_key.device
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty(no, 2 * Do, h, w, dtype=torch.float32, device=dev)
        else:
            _check(out, 'out')
        p = torch.empty(no, T * h * w, h * w, dtype=torch.float32, device=dev) if want_p else None
        nb = lib.rmnet_memory_read_workspace_bytes(no, De, Do, T, h, w, int(flags))
        ws = _ws(nb, dev)
        cs = Tcap * h * w
        ev = [ctypes.c_void_p(e) if e else None for e in (events or (None, None, None))]
        rc = lib.rmnet_memory_read_f32_ev(_ptr(m_key), _ptr(m_val), _ptr(q_key), _ptr(q_val), no, De, Do,
                                          T, h, w, cs, cs * De, cs, cs * Do, _ptr(out), _ptr(p),
                                          _ptr(mem_rects), _ptr(qry_rects), int(flags), _ptr(ws),
                                          ws.numel(), _stream(dev), ev[0], ev[1], ev[2])
    _lib.check(rc, 'rmnet_memory_read_f32')
    return out, p


BANK_F16 = 4                      # include/rmnet_hip.h: RMNET_BANK_F16 (== RMNET_MR_F16)
MR_F16 = 4
BANK_QX = 8                       # RMNET_BANK_QX (== RMNET_MR_QX)
MR_QX = 8
_PRECISION_FLAGS = {'split': 0, 'f16': BANK_F16, 'qx': BANK_QX}


def _precision(p):
    """'split': K, V, q and P enter the MFMAs as fp16 hi/lo pairs, three terms, fp32-class accuracy (default).
    'f16': hi planes only -- fp16 operands, fp32 accumulate, about 2^-11 relative (include/rmnet_hip.h).
    'qx': 'f16' with the query as a hi/lo pair (its rounding is the logit error that does not average out)."""
    if p not in _PRECISION_FLAGS:
        raise ValueError("precision must be 'split', 'qx' or 'f16'")
    return p


def _loop_precision(p):
    """Arithmetic of the frame loop's bank read: 'auto' (default), 'split', 'qx' or 'f16' -- see RMNet.__init__ and
    profiles/r05_iou_calibration.md for what 'auto' picks and why."""
    if p not in ('auto', 'split', 'qx', 'f16'):
        raise ValueError("read_precision must be 'auto', 'split', 'qx' or 'f16'")
    return p


class MemoryBank:
    """Device-resident regional memory of one clip: ``no`` objects x ``capacity`` frame slots on an
    h x w feature grid (csrc/bank.hip).  ``append`` writes a slot from the un-masked KeyValue outputs
    and the frame's cell rectangles; ``read`` runs the fused regional read over the first T slots."""

    def __init__(self, no, capacity, h, w, device, precision='split'):
        lib = _lib.load()
        self.no, self.capacity, self.h, self.w = int(no), int(capacity), int(h), int(w)
        self.device = torch.device(device)
        self.precision = _precision(precision)        # arithmetic of ``read``: 'split' (fp32-class, default), 'qx' or 'f16'
        nb = lib.rmnet_bank_bytes(self.no, self.capacity, self.h, self.w)
        if nb == 0:
            raise RuntimeError('invalid bank geometry')
        with torch.cuda.device(self.device):
            self.blob = torch.zeros(nb, dtype=torch.uint8, device=self.device)
            # device-resident copy of `committed`: stage() / read_staged() hand it to the kernels, so a captured HIP graph
            # of the frame step stays valid while the memory grows (rmnet_bank_*_at in include/rmnet_hip.h)
            self.n_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.committed = 0

    def append(self, slot, k4, v4, rects=None):
        _check(k4, 'k4')
        _check(v4, 'v4')
        if tuple(k4.shape) != (self.no, 128, self.h, self.w) or tuple(v4.shape) != (self.no, 512, self.h, self.w):
            raise RuntimeError('k4/v4 must be [no,128,h,w] / [no,512,h,w]')
        if rects is not None:
            _check(rects, 'rects', torch.int32)
            if rects.numel() != self.no * 4:
                raise RuntimeError('rects must be [no,4]')
        lib = _lib.load()
        with torch.cuda.device(self.device):
            rc = lib.rmnet_bank_append_f32(_ptr(self.blob), self.no, self.capacity, self.h, self.w, int(slot),
                                           _ptr(k4), _ptr(v4), _ptr(rects), _stream(self.device))
        _lib.check(rc, 'rmnet_bank_append_f32')

    def areas(self):
        """[no, capacity] int32 view of the per-slot cell counts kept in the blob (debug / accounting only)."""
        off = _lib.load().rmnet_bank_area_offset(self.no, self.capacity, self.h, self.w)
        return self.blob[off:off + self.no * self.capacity * 4].view(torch.int32).view(self.no, self.capacity)

    def overflow_count(self):
        """Number of 16-byte groups appended so far that held an element outside the bank's fp16
        window (|x| >= 1023.5, NaN, Inf; include/rmnet_hip.h).  Synchronises the stream -- call it once
        per clip, not per frame.  Non-zero = the read-outs of this bank are not trustworthy."""
        off = _lib.load().rmnet_bank_overflow_offset(self.no, self.capacity, self.h, self.w)
        return int(self.blob[off:off + 4].view(torch.int32).item())

    def timeout_count(self):
        """Merges of a read that gave up waiting for another workgroup's partial (the int32 behind the overflow word).  Always 0
        on a healthy device; a non-zero value also sets a sticky bit in the overflow word.  Synchronises the stream."""
        off = _lib.load().rmnet_bank_overflow_offset(self.no, self.capacity, self.h, self.w)
        return int(self.blob[off + 4:off + 8].view(torch.int32).item())

    def status(self):
        """(overflow word, merge time-outs, largest logit so far) in ONE device-to-host copy -- what the frame loop checks once per clip."""
        off = _lib.load().rmnet_bank_overflow_offset(self.no, self.capacity, self.h, self.w)
        w = self.blob[off:off + 12].cpu()
        return int(w[0:4].view(torch.int32).item()), int(w[4:8].view(torch.int32).item()), float(w[8:12].view(torch.float32).item()) * 0.6931471805599453

    def logit_max(self):
        """Largest affinity logit (natural units, S = k . q / sqrt(128): models/rmnet.py:155-157) any read of this bank has used as a
        soft-max reference so far -- the kernel's deferred reference, i.e. a lower bound within 8 of the true maximum (include/rmnet_hip.h,
        "logit word").  Synchronises the stream: once per clip."""
        off = _lib.load().rmnet_bank_overflow_offset(self.no, self.capacity, self.h, self.w)
        return float(self.blob[off + 8:off + 12].view(torch.float32).item()) * 0.6931471805599453

    def assert_synced(self):
        """Debug aid: the host mirror ``committed`` and the device counter ``n_dev`` agree (they only move together in
        ``commit``; a captured ``frame_step(commit=True)`` or a foreign write to either would desynchronise them silently --
        the kernels flag an out-of-range slot in the overflow word, an in-range wrong slot only this check finds).  Synchronises."""
        n = int(self.n_dev.item())
        if n != self.committed:
            raise RuntimeError('MemoryBank: device frame counter %d != host counter %d' % (n, self.committed))

    def stage(self, k4, v4, rects):
        """Write one frame into the first free slot without committing it (the tentative previous
        frame of models/rmnet.py:416-426).  Returns the number of frames visible to ``read``.  The slot index
        travels as the device counter ``n_dev`` (graph-replayable); ``committed`` mirrors it on the host."""
        if self.committed >= self.capacity:
            raise RuntimeError('memory bank overflow (%d slots)' % self.capacity)
        _check(k4, 'k4')
        _check(v4, 'v4')
        if tuple(k4.shape) != (self.no, 128, self.h, self.w) or tuple(v4.shape) != (self.no, 512, self.h, self.w):
            raise RuntimeError('k4/v4 must be [no,128,h,w] / [no,512,h,w]')
        if rects is not None:
            _check(rects, 'rects', torch.int32)
        lib = _lib.load()
        with torch.cuda.device(self.device):
            rc = lib.rmnet_bank_append_f32_at(_ptr(self.blob), self.no, self.capacity, self.h, self.w, 0, _ptr(self.n_dev),
                                              _ptr(k4), _ptr(v4), _ptr(rects), _stream(self.device))
        _lib.check(rc, 'rmnet_bank_append_f32_at')
        return self.committed + 1

    def commit(self):
        """Keep the staged frame: one-element add on the device counter (after the read that used it as tentative).  Not inside
        a graph capture: the host mirror would advance once, the device counter on every replay."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('MemoryBank.commit() inside a HIP graph capture: commit between replays (RMNet.forward does)')
        self.committed += 1
        self.n_dev += 1

    def read_staged(self, q_key, q_val, qry_rects=None, out=None, events=None, ws=None):
        """``read(committed + 1, ...)`` with the frame count taken from the device counter (committed frames + the
        staged one): the call a captured graph replays.  A bank of more than 2048 slots is read in chunks planned on the
        host (csrc/memory_read.hip: launch_bank_read), so there the host's own count is used."""
        if self.capacity > BANK_MAX_SLOTS:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('MemoryBank.read_staged() of a bank of more than %d slots inside a HIP graph capture: the chunked '
                                   'read is planned on the host, a replay would keep reading the frame count of the capture' % BANK_MAX_SLOTS)
            return self.read(self.committed + 1, q_key, q_val, qry_rects, out=out, events=events, ws=ws)
        return self.read(1, q_key, q_val, qry_rects, out=out, events=events, ws=ws, _t_dev=self.n_dev)

    def read(self, T, q_key, q_val, qry_rects=None, out=None, events=None, ws=None, _t_dev=None):
        _check(q_key, 'q_key')
        _check(q_val, 'q_val')
        if tuple(q_key.shape) != (self.no, 128, self.h, self.w) or tuple(q_val.shape) != (self.no, 512, self.h, self.w):
            raise RuntimeError('q_key/q_val must be [no,128,h,w] / [no,512,h,w]')
        if qry_rects is not None:
            _check(qry_rects, 'qry_rects', torch.int32)
        if not 1 <= int(T) <= self.capacity:
            raise RuntimeError('T out of range')
        lib = _lib.load()
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty(self.no, 1024, self.h, self.w, dtype=torch.float32, device=self.device)
            if ws is None:
                ws = _ws(lib.rmnet_bank_read_workspace_bytes_for(self.no, self.h, self.w, int(T)), self.device)
            ev = [ctypes.c_void_p(e) if e else None for e in (events or (None, None, None))]
            rc = lib.rmnet_bank_read_f32_at(_ptr(self.blob), self.no, self.capacity, self.h, self.w, int(T), _ptr(_t_dev),
                                            _PRECISION_FLAGS[self.precision], _ptr(q_key), _ptr(q_val), _ptr(qry_rects), _ptr(out), _ptr(ws),
                                            ws.numel(), _stream(self.device), ev[0], ev[1], ev[2])
        _lib.check(rc, 'rmnet_bank_read_f32_at')
        return out


class TensorBank:
    """Same interface as ``MemoryBank`` on plain fp32 tensors in the reference's layout
    ([no,C,Tcap,h,w] + cell rectangles), read with the exact-fp32 kernel: no limit on the value range, about
    4x slower.  The frame loop switches to it when a ``MemoryBank`` reported out-of-window values (beyond 2048
    memorised frames it falls to the generic kernels, with a warning: the exact fused kernel takes one launch's frames)."""

    def __init__(self, no, capacity, h, w, device):
        self.no, self.capacity, self.h, self.w = int(no), int(capacity), int(h), int(w)
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            self.m_key = torch.zeros(self.no, 128, self.capacity, self.h, self.w, device=self.device)
            self.m_val = torch.zeros(self.no, 512, self.capacity, self.h, self.w, device=self.device)
            self.rects = torch.zeros(self.no, self.capacity, 4, dtype=torch.int32, device=self.device)
            # the whole-grid rectangle, built ONCE (a torch.tensor([...], device=...) per call is a blocking host-to-device copy
            # in the middle of the frame loop: this is the overflow fallback of every clip)
            self._full = torch.tensor([[0, self.w - 1, 0, self.h - 1]] * self.no, dtype=torch.int32, device=self.device)
        self.committed = 0

    def append(self, slot, k4, v4, rects=None):
        _check(k4, 'k4')
        _check(v4, 'v4')
        self.m_key[:, :, slot] = k4
        self.m_val[:, :, slot] = v4
        self.rects[:, slot] = self._full if rects is None else rects.view(self.no, 4)

    def overflow_count(self):
        return 0

    def logit_max(self):
        """(MemoryBank's interface: the exact-fp32 kernels have no arithmetic whose error grows with the logits.)"""
        return 0.0

    def status(self):
        return 0, 0, 0.0

    def timeout_count(self):
        """(MemoryBank's interface: the exact-fp32 kernels have no cross-workgroup merge that could time out.)"""
        return 0

    def assert_synced(self):
        """(MemoryBank's interface: there is no device-resident frame counter here.)"""

    def stage(self, k4, v4, rects):
        if self.committed >= self.capacity:
            raise RuntimeError('memory bank overflow (%d slots)' % self.capacity)
        self.append(self.committed, k4, v4, rects)
        return self.committed + 1

    def commit(self):
        self.committed += 1

    def read_staged(self, q_key, q_val, qry_rects=None, out=None, events=None, ws=None):
        return self.read(self.committed + 1, q_key, q_val, qry_rects, out=out, events=events, ws=ws)

    def read(self, T, q_key, q_val, qry_rects=None, out=None, events=None, ws=None):
        if qry_rects is None:
            qry_rects = self._full
        out, _ = memory_read(self.m_key, self.m_val, q_key, q_val, self.rects[:, :T].contiguous(),
                             qry_rects.contiguous(), T=T, out=out, events=events,
                             flags=MR_EXACT_FP32 if T <= BANK_MAX_SLOTS else self._generic_flags(T))
        return out

    def _generic_flags(self, T):
        import warnings
        nbytes = self.no * T * (self.h * self.w) ** 2 * 4
        warnings.warn('rmnet_amd: %d memorised frames exceed the fused kernels (%d): falling back to the generic path, which '
                      'materialises the affinity (%.1f GB of workspace)' % (T, BANK_MAX_SLOTS, nbytes / 1e9))
        return MR_FORCE_GENERIC


def rect_mask(x, rects):
    """x [n,C,T,h,w] * 0/1 cell rectangles [n,T,4] (models/rmnet.py:247-248, 357-358)."""
    _check(x, 'x')
    _check(rects, 'rects', torch.int32)
    n, C, T, h, w = x.shape
    if rects.numel() != n * T * 4:
        raise RuntimeError('rects must be [n,T,4]')
    lib = _lib.load()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.rmnet_rect_mask_f32(_ptr(x), n, C, T, h, w, _ptr(rects), _ptr(y), _stream(x.device))
    _lib.check(rc, 'rmnet_rect_mask_f32')
    return y


def channel_affine(x, scale=None, shift=None, res=None, res_scale=None, res_shift=None, relu=False,
                   out=None):
    """out = act(x * scale[c] + shift[c] + (res * res_scale[c] + res_shift[c])) for NCHW fp32 ``x`` in
    one pass (csrc/epilogue.hip); ``relu``: False, True, or 'leaky' (= LeakyReLU(0.1)).  ``out`` may be
    ``x`` or ``res`` (in place); default: a new tensor.
    Replaces BatchNorm2d(eval) / conv bias / skip add / ReLU sequences around the convolutions."""
    _check_act(x, 'x')
    if x.dim() != 4:
        raise RuntimeError('x must be [N,C,H,W]')
    N, C, H, W = x.shape
    cl = _is_cl(x)                  # channels-last activations: the NHWC kernel (same arithmetic; csrc/epilogue.hip)
    for t, n in ((scale, 'scale'), (shift, 'shift'), (res_scale, 'res_scale'), (res_shift, 'res_shift')):
        if t is not None:
            _check(t, n)
            if t.numel() != C:
                raise RuntimeError('%s must have C = %d elements' % (n, C))
    if res is not None:
        _check_act(res, 'res')
        if res.shape != x.shape:
            raise RuntimeError('res must have the shape of x')
        if cl and not res.is_contiguous(memory_format=torch.channels_last):
            res = res.contiguous(memory_format=torch.channels_last)      # (a skip in the other layout: one conversion)
        elif not cl and not res.is_contiguous():
            res = res.contiguous()
    elif res_scale is not None or res_shift is not None:
        raise RuntimeError('res_scale / res_shift without res')
    if out is None:
        out = torch.empty_like(x)       # (preserves the memory format)
    else:
        _check_act(out, 'out')
        if out.shape != x.shape or not (out.is_contiguous(memory_format=torch.channels_last) if cl else out.is_contiguous()):
            raise RuntimeError('out must have the shape and memory format of x')
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if cl and C % 4 == 0:
            rc = lib.rmnet_channel_affine_nhwc_f32(_ptr(x), _ptr(scale), _ptr(shift), _ptr(res), _ptr(res_scale), _ptr(res_shift),
                                                   2 if relu == 'leaky' else (1 if relu else 0), N * H * W, C, _ptr(out), _stream(x.device))
            _lib.check(rc, 'rmnet_channel_affine_nhwc_f32')
            return out
        if cl:
            raise RuntimeError('channels-last channel_affine needs C % 4 == 0')
        rc = lib.rmnet_channel_affine_f32(_ptr(x), _ptr(scale), _ptr(shift), _ptr(res), _ptr(res_scale),
                                          _ptr(res_shift), 2 if relu == 'leaky' else (1 if relu else 0), N, C, H * W, _ptr(out),
                                          _stream(x.device))
    _lib.check(rc, 'rmnet_channel_affine_f32')
    return out


def upsample2x_add(x, skip=None, out=None):
    """skip + F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) in one pass
    (csrc/epilogue.hip); ``skip`` None = plain upsample; ``out`` may be ``skip`` (in place)."""
    _check_act(x, 'x')
    if x.dim() != 4:
        raise RuntimeError('x must be [N,C,h,w]')
    N, C, h, w = x.shape
    shape = (N, C, 2 * h, 2 * w)
    cl = _is_cl(x) or (skip is not None and _is_cl(skip))
    fmt = torch.channels_last if cl else torch.contiguous_format
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)      # (no copy when it already is)
    if skip is not None:
        _check_act(skip, 'skip')
        if tuple(skip.shape) != shape:
            raise RuntimeError('skip must be [N,C,2h,2w]')
        if cl and not _is_cl(skip):
            skip = skip.contiguous(memory_format=torch.channels_last)
    if out is None or (cl and not _is_cl(out)):
        out = torch.empty(shape, dtype=x.dtype, device=x.device, memory_format=fmt)
    else:
        _check_act(out, 'out')
        if tuple(out.shape) != shape:
            raise RuntimeError('out must be [N,C,2h,2w]')
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if cl:
            rc = lib.rmnet_upsample2x_add_nhwc_f32(_ptr(x), _ptr(skip), N, C, h, w, _ptr(out), _stream(x.device))
            _lib.check(rc, 'rmnet_upsample2x_add_nhwc_f32')
            return out
        rc = lib.rmnet_upsample2x_add_f32(_ptr(x), _ptr(skip), N, C, h, w, _ptr(out), _stream(x.device))
    _lib.check(rc, 'rmnet_upsample2x_add_f32')
    return out


def soft_aggregate(dec, obj_begin, K, pad, want_prob=False):
    """Decoder logits [n_tot,2,Hp,Wp] -> (logit [B,K,H,W], prob [B,K,H,W] | None): 2-class soft-max,
    soft aggregation (models/rmnet.py:289-302), un-pad by ``pad = (lw, uw, lh, uh)`` and, when asked,
    the soft-max over the K channels, in one kernel.  ``obj_begin`` int32 [B+1] on the device."""
    _check(dec, 'dec')
    _check(obj_begin, 'obj_begin', torch.int32)
    if dec.dim() != 4 or dec.shape[1] != 2:
        raise RuntimeError('dec must be [n,2,Hp,Wp]')
    B = obj_begin.numel() - 1
    Hp, Wp = dec.shape[2:]
    lw, uw, lh, uh = pad
    H, W = Hp - lh - uh, Wp - lw - uw
    logit = torch.empty(B, K, H, W, dtype=dec.dtype, device=dec.device)
    prob = torch.empty_like(logit) if want_prob else None
    lib = _lib.load()
    with torch.cuda.device(dec.device):
        rc = lib.rmnet_soft_aggregate_f32(_ptr(dec), _ptr(obj_begin), B, K, Hp, Wp, lw, lh, H, W,
                                          _ptr(logit), _ptr(prob), _stream(dec.device))
    _lib.check(rc, 'rmnet_soft_aggregate_f32')
    return logit, prob


def affine_relu_maxpool(x, scale=None, shift=None):
    """max_pool2d(relu(x * scale[c] + shift[c]), 3, stride=2, padding=1) in one pass (csrc/epilogue.hip):
    the ResNet stem's bn1 -> relu -> maxpool without the full-resolution intermediate."""
    _check_act(x, 'x')
    if x.dim() != 4:
        raise RuntimeError('x must be [N,C,H,W]')
    N, C, H, W = x.shape
    cl = _is_cl(x) and C % 4 == 0
    if _is_cl(x) and not cl:
        x = x.contiguous()
    for t, n in ((scale, 'scale'), (shift, 'shift')):
        if t is not None:
            _check(t, n)
            if t.numel() != C:
                raise RuntimeError('%s must have C = %d elements' % (n, C))
    out = torch.empty(N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, dtype=x.dtype, device=x.device,
                      memory_format=torch.channels_last if cl else torch.contiguous_format)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if cl:
            rc = lib.rmnet_affine_relu_maxpool_nhwc_f32(_ptr(x), _ptr(scale), _ptr(shift), N, C, H, W, _ptr(out), _stream(x.device))
            _lib.check(rc, 'rmnet_affine_relu_maxpool_nhwc_f32')
            return out
        rc = lib.rmnet_affine_relu_maxpool_f32(_ptr(x), _ptr(scale), _ptr(shift), N, C, H, W, _ptr(out),
                                               _stream(x.device))
    _lib.check(rc, 'rmnet_affine_relu_maxpool_f32')
    return out


def flow_affine(flow, m1, m2):
    """Device-resident variant: flow [H,W,2] f32 cuda, m1/m2 [2,3] f32 cuda -> [H,W,2]."""
    for t, n in ((flow, 'flow'), (m1, 'm1'), (m2, 'm2')):
        _check(t, n)
    if flow.dim() != 3 or flow.shape[2] != 2 or m1.numel() != 6 or m2.numel() != 6:
        raise RuntimeError('expected flow [H,W,2] and 2x3 matrices')
    lib = _lib.load()
    out = torch.empty_like(flow)
    with torch.cuda.device(flow.device):
        rc = lib.rmnet_flow_affine_f32(_ptr(flow), _ptr(m1), _ptr(m2), flow.shape[0], flow.shape[1],
                                       _ptr(out), _stream(flow.device))
    _lib.check(rc, 'rmnet_flow_affine_f32')
    return out


def update_optical_flow(optical_flow, tr_matrix1, tr_matrix2, device=None):
    """NumPy calling convention of the reference's CPython module
    (flow_affine_transformation.cpp:87-90): ndarray in, new ndarray out -- computed on the GPU.
    Unlike the reference (which reinterprets whatever buffer it is given, .cpp:50-52) the inputs
    are checked/converted to C-contiguous float32."""
    flow = np.ascontiguousarray(optical_flow, dtype=np.float32)
    m1 = np.ascontiguousarray(tr_matrix1, dtype=np.float32)
    m2 = np.ascontiguousarray(tr_matrix2, dtype=np.float32)
    if flow.ndim != 3 or flow.shape[2] != 2 or m1.size != 6 or m2.size != 6:
        raise RuntimeError('expected flow [H,W,2] and 2x3 matrices')
    lib = _lib.load()
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    H, W = flow.shape[:2]
    out = np.empty_like(flow)
    with torch.cuda.device(dev):
        ws = _ws(lib.rmnet_flow_affine_workspace_bytes(H, W), dev)
        rc = lib.rmnet_flow_affine_f32_host(flow.ctypes.data, m1.ctypes.data, m2.ctypes.data, H, W,
                                            out.ctypes.data, _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(rc, 'rmnet_flow_affine_f32_host')
    return out
