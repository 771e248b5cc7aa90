# -*- coding: utf-8 -*-
"""Host-side mirror of the reference's ``models/rmnet.py`` on top of the gfx950 kernels.

Same classes, constructor arguments, method names, argument meaning and state-dict keys as the
reference (``RMNet``, ``MemoryReader``, ``KeyValue``, ``Decoder`` ...; models/rmnet.py:123-452), so
``rmnet.load_state_dict(checkpoint['rmnet'])`` and
``rmnet(frames, masks, optical_flows, n_objects, memorize_every)`` work unchanged.  What differs is
how the per-frame loop is carried out:

=====================================  ==========================================================
reference (models/rmnet.py)            here
=====================================  ==========================================================
full-res 0/1 box maps, x1/16 nearest,  the region kernel emits the box AND its cell rectangle;
4 elementwise multiplies (:244-248,    the full-resolution map is not written and K/V are never
:356-358)                              multiplied -- the read kernel skips masked cells exactly
bmm / div / softmax / bmm / cat with   one fused regional read (csrc/memory_read.hip), p never
p materialised (:147-165)              materialised
K-slot zero padding + torch.cat of     pre-allocated per-object bank [no, C, Tcap, h, w]; memorising
the whole memory every frame           = one strided slab write; the tentative "previous frame"
(:191-205, :416-426)                   slot is simply overwritten until it is committed
est_masks on the host, H2D/D2H and     clip resident on the device, no host sync inside the frame
.item()/.tolist() every frame          loop (object bookkeeping is hoisted to clip start)
(:386-402, :412-413, :436-450)
=====================================  ==========================================================

There is no CPU fallback: every call goes through librmnet_hip.so (rmnet_amd._lib) or raises.
"""

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .helpers import pad_amounts, pad_divide_by
from .networks import Decoder, EncoderMemory, EncoderQuery, KeyValue, Refine, ResBlock  # noqa: F401
from .reg_att_map_generator import RegionalAttentionMapGenerator

_ABSENT_LOGIT = -16.1181          # models/rmnet.py:441, 448
_NEW_OBJECT_SCALE = 32.0605       # models/rmnet.py:441


class MemoryReader(nn.Module):
    """models/rmnet.py:143-165.  ``forward(m_key, m_val, q_key, q_val) -> (mem_val, p)``.

    ``p`` (the [no, THW, HW] affinity the reference returns as ``viz`` and never uses, :361-366) is
    only computed when ``return_affinity=True``; otherwise the second element is ``None``.
    Optional ``mem_rects`` / ``qry_rects`` select the fused regional form."""

    def __init__(self, return_affinity=False, precision='split'):
        super().__init__()
        self.return_affinity = return_affinity
        self.precision = ops._precision(precision)     # 'qx' / 'f16': see ops._precision (ops.MR_QX, ops.MR_F16)

    def forward(self, m_key, m_val, q_key, q_val, mem_rects=None, qry_rects=None, T=None):
        return ops.memory_read(m_key.contiguous(), m_val.contiguous(), q_key.contiguous(),
                               q_val.contiguous(), mem_rects, qry_rects,
                               want_p=self.return_affinity, T=T,
                               flags=ops._PRECISION_FLAGS[self.precision])


# 'auto' read precision: the largest affinity logit (natural units; the bank's deferred soft-max reference, i.e. up to 8 below the true
# maximum) up to which a one-object clip stays in the fp16-operand arithmetic.  Calibrated by the key-temperature sweep
# (profiles/r06_iou_temperature.md): at |S| <= ~10 'f16' is at 0.9999+ of the CPU path's masks, at ~35 it is at 0.9994-0.9999, at ~150 below
# the 0.999 bar.
AUTO_LOGIT_BOUND = 12.0


class RMNet(nn.Module):
    """INFERENCE ONLY.  The memory read, the bank, the box masking and the fused decoder tail are raw
    HIP kernels without autograd: ``forward`` / ``frame_step`` / ``segment`` / ``memorize`` run under
    ``torch.no_grad()`` and refuse to run in training mode (the reference's training path -- losses,
    DataParallel, models/rmnet.py's ``self.training`` branches -- is out of scope)."""

    def __init__(self, cfg=None, read_precision='auto'):
        super().__init__()
        self.cfg = cfg
        # arithmetic of the bank read in the frame loop:
        #   'split' = fp16 hi/lo pairs, three MFMA terms, fp32-class accuracy (1e-7) -- what rounds 1-3 ran everywhere;
        #   'f16'   = fp16 operands (K, q, P, V rounded to 11 bits), fp32 accumulate, 1.7x as fast;
        #   'qx'    = 'f16' with the QUERY as a hi/lo pair (two terms for the logits);
        #   'auto' (default) = decided per clip, from what is MEASURED on the clip [r6]:
        #       * clips with several objects: 'split' (the soft aggregation amplifies the read's error: on 3 / 5-object clips 'qx' is at
        #         0.9993-0.9997 and 'f16' at 0.9986-0.9995 of the CPU path's masks -- inside the task's 1e-3 only just, or not at all);
        #       * clips with one object: 'f16' -- and the bank keeps the largest affinity logit its reads have seen (rmnet_hip.h, logit
        #         word); when that exceeds AUTO_LOGIT_BOUND the clip is read again in 'split', like a clip whose values left the bank's
        #         fp16 window.  Why: rounding K and q to 11 bits costs ~2^-11 |S| per logit.  With near-uniform soft-maxes (the
        #         procedural weights: |S| < 10) that averages out -- 'f16' is at 0.99991-0.99997 on live-boundary clips -- but with the
        #         key convolutions scaled so that the soft-max is peaked (top-1 mass 0.5: |S| ~ 150) 'f16' AND 'qx' fall to 0.9985-0.9990,
        #         below the bar, while 'split' stays on the exact loop (profiles/r06_iou_temperature.md; tests/test_gpu_parity.py
        #         test_key_temperature_sweep).  No trained checkpoint is reachable offline: the bound is what the sweep supports.
        #         The decision is STICKY per network: the logit scale is a property of the weights, so once a one-object clip had to be
        #         re-read, later clips of this network start in 'split' (no clip is processed twice again); ``reset_auto_precision()``
        #         -- also called when reference weights are loaded -- returns to the optimistic start.
        self.read_precision = ops._loop_precision(read_precision)
        self._auto_peaked = False
        self.encoder_memory = EncoderMemory()
        self.encoder_query = EncoderQuery()
        self.kv_memory = KeyValue(1024, keydim=128, valdim=512)
        self.kv_query = KeyValue(1024, keydim=128, valdim=512)
        self.memory = MemoryReader()
        self.decoder = Decoder(256)
        self.att_map_generator = RegionalAttentionMapGenerator()
        self._grid_cache = {}

    # ------------------------------------------------------------------ checkpoint convenience
    def load_reference_state_dict(self, state_dict, strict=True):
        """Accepts the reference's checkpoints with or without DataParallel's ``module.`` prefix
        (core/inference.py:43, utils/eval_server.py:92)."""
        clean = {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        self.reset_auto_precision()          # (new weights, new logit scale)
        return self.load_state_dict(clean, strict=strict)

    def reset_auto_precision(self):
        """'auto' forgets that a clip of this network measured logits beyond AUTO_LOGIT_BOUND (see __init__)."""
        self._auto_peaked = False

    def fuse_for_inference(self):
        """Fold the eval-mode BatchNorms of both ResNet-50 trunks into their convolutions (in place;
        call after loading weights).  Same function up to fp32 rounding, ~6 % fewer GPU cycles."""
        from .networks import fold_batchnorm_
        self.eval()
        fold_batchnorm_(self.encoder_memory)
        fold_batchnorm_(self.encoder_query)
        return self

    def channels_last(self):
        """Both convolution stacks in channels_last memory format ([N,H,W,C] in memory): MIOpen's NHWC kernels then run without the layout
        transposes that wrap them on NCHW tensors, and the fused glue kernels take their channels-last variants (csrc/epilogue.hip) -- +7 %
        frames/s on the 480p workload (profiles/r06_conv_layout.md).  Same parameters, same state dict, same results up to the convolutions'
        own rounding.  ATen hands channels_last tensors to MIOpen as NHWC only when PYTORCH_MIOPEN_SUGGEST_NHWC=1 is in the environment
        BEFORE the first convolution runs: set here if it is not set at all.  The HIP kernels of the read path keep their NCHW interfaces
        (keys / values / read-out are converted at the boundary: small tensors)."""
        import os
        os.environ.setdefault('PYTORCH_MIOPEN_SUGGEST_NHWC', '1')
        return self.to(memory_format=torch.channels_last)

    def fuse_epilogues(self, enable=True):
        """Run BatchNorm(eval) / conv bias / skip add / ReLU as ONE pass per convolution
        (rmnet_channel_affine_f32) in both encoders and the decoder.  Parameters and state dict are
        untouched; results differ from the module graph by fp32 rounding only (tests: 1e-5 relative per
        block, mask probabilities < 1e-3 per clip)."""
        from .networks import fuse_epilogues_
        self.eval()
        fuse_epilogues_(self, enable)
        self._fused_tail = bool(enable)     # decoder tail: rmnet_soft_aggregate_f32 (frame loop only)
        self._fused_warp = None             # warp inside the box reduction: self-checked on first use
        return self

    def _fused_warp_ok(self, device):
        """The warp-fused box kernel reproduces, operation by operation, what THIS torch / ROCm build
        executes for ``warp()`` (reciprocal multiply of the scalar division, grid_sampler's unnormalise,
        FMA order, the 0.9999 validity test).  Another build could differ by one ulp and move a >= 0.5
        decision, i.e. a box -- silently.  So the first use on a device compares one small random warp bit
        for bit; on any difference the loop falls back to region_map(self.warp(...)) and says so."""
        if self._fused_warp is None:
            g = torch.Generator().manual_seed(1234)
            m = torch.rand(1, 3, 37, 53, generator=g).to(device)
            f = (torch.randn(1, 2, 37, 53, generator=g) * 6.0).to(device)
            want = self.warp(m, f)[0].contiguous()
            _, bb, _, got = ops.region_map(m, want_map=False, flow=f, want_warped=True)
            _, bb0, _ = ops.region_map(want, want_map=False)
            self._fused_warp = bool(torch.equal(got[:, 1:], want[:, 1:]) and torch.equal(bb, bb0))
            if not self._fused_warp:
                import warnings
                warnings.warn('rmnet_amd: the warp-fused region kernel is not bit-identical to torch.grid_sample on this '
                              'torch/ROCm build; using region_map(warp(...)) instead')
        return self._fused_warp

    # ------------------------------------------------------------------ small helpers
    @staticmethod
    def _object_index(n_objects, K, device):
        """Flat indices b*K + o of the objects in flight (o = 1..n_objects[b]) and their batch ids."""
        flat = [b * K + o for b, n in enumerate(n_objects) for o in range(1, n + 1)]
        batch = [b for b, n in enumerate(n_objects) for _ in range(n)]
        return (torch.tensor(flat, dtype=torch.long, device=device),
                torch.tensor(batch, dtype=torch.long, device=device))

    def _base_grid(self, B, H, W, device):
        key = (B, H, W, str(device))
        g = self._grid_cache.get(key)
        if g is None:
            xs = torch.arange(W, dtype=torch.float32, device=device).view(1, 1, 1, W).expand(B, 1, H, W)
            ys = torch.arange(H, dtype=torch.float32, device=device).view(1, 1, H, 1).expand(B, 1, H, W)
            g = torch.cat((xs, ys), dim=1).contiguous()
            self._grid_cache = {key: g}
        return g

    # ------------------------------------------------------------------ reference API: pieces
    def pad_memory(self, mems, n_objects, K):
        """models/rmnet.py:191-205 -- kept for API parity (the frame loop below does not use it)."""
        out = []
        B = len(n_objects)
        for mem in mems:
            _, C, H, W = mem.shape
            padded = mem.new_zeros(B, K, C, 1, H, W)
            at = 0
            for b in range(B):
                padded[b, 1:n_objects[b] + 1, :, 0] = mem[at:at + n_objects[b]]
                at += n_objects[b]
            out.append(padded)
        return out

    def warp(self, img0, flow):
        """models/rmnet.py:252-278: bilinear backward warp + validity mask (PyTorch-ROCm ops)."""
        B, C, H, W = img0.shape
        vgrid = self._base_grid(B, H, W, img0.device) + flow
        gx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
        gy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
        grid = torch.stack((gx, gy), dim=3)
        img1 = F.grid_sample(img0, grid, align_corners=True)
        mask = F.grid_sample(torch.ones_like(img0), grid, align_corners=True)
        mask = (mask >= 0.9999).to(img0.dtype)
        return img1 * mask, mask

    def get_att_map(self, prev_mask, flow=None):
        """models/rmnet.py:280-287 -> (att_map [B,K,H,W], bbox [B,K,4])."""
        expt = prev_mask if flow is None else self.warp(prev_mask, flow)[0]
        return self.att_map_generator(expt.contiguous())

    def _inference_only(self):
        if self.training:
            raise RuntimeError('rmnet_amd.RMNet is inference-only (its HIP kernels have no autograd): call .eval()')

    def _encode_memory(self, frame, masks, n_objects):
        """EncoderMemory + KeyValue for every object in flight, and the boxes of the (padded)
        masks as pixel boxes + cell rectangles.  K/V are returned UN-masked."""
        B, K, H, W = masks.shape
        (frame, masks), _ = pad_divide_by([frame, masks.float()], 16, (H, W))
        if all(int(n) == 1 for n in n_objects):       # one object per clip: object i IS clip i, nothing to gather
            f_in, m_in = frame, masks[:, 1]
            o_in = torch.zeros_like(m_in)
        else:
            fs, ms, os_ = [], [], []
            for b in range(B):
                n = n_objects[b]
                for o in range(1, n + 1):
                    fs.append(frame[b:b + 1])
                    ms.append(masks[b, o:o + 1])
                    if n == 1:
                        os_.append(torch.zeros_like(masks[b, o:o + 1]))
                    else:  # same summation order as models/rmnet.py:224-226
                        others = masks[b, 1:o].sum(0, keepdim=True) + masks[b, o + 1:n + 1].sum(0, keepdim=True)
                        os_.append(others.clamp(0, 1))
            f_in, m_in, o_in = torch.cat(fs), torch.cat(ms), torch.cat(os_)
        r4 = self.encoder_memory(f_in, m_in, o_in)[0]
        k4, v4 = self.kv_memory(r4)
        h, w = k4.shape[-2:]
        _, bboxes, rects = ops.region_map(masks.contiguous(), want_map=False, cell_grid=(0, 0, 16, h, w))
        return k4, v4, bboxes, rects

    @torch.no_grad()
    def memorize(self, frame, masks, n_objects):
        """models/rmnet.py:207-250 with the reference's return values:
        (k4 [B,K,128,1,h,w], v4 [B,K,512,1,h,w], bboxes [B,K,4]), K/V box-masked."""
        self._inference_only()
        B, K = masks.shape[:2]
        k4, v4, bboxes, rects = self._encode_memory(frame, masks, n_objects)
        k4, v4 = self.pad_memory([k4, v4], n_objects, K)
        r = rects.view(B * K, 1, 4)
        k4 = ops.rect_mask(k4.view(B * K, -1, 1, *k4.shape[-2:]).contiguous(), r).view_as(k4)
        v4 = ops.rect_mask(v4.view(B * K, -1, 1, *v4.shape[-2:]).contiguous(), r).view_as(v4)
        return k4, v4, bboxes

    def soft_aggregation(self, ps, K, n_objects):
        """models/rmnet.py:289-302."""
        B = len(n_objects)
        em = ps.new_zeros(B, K, *ps.shape[1:])
        at = 0
        for b in range(B):
            sl = ps[at:at + n_objects[b]]
            em[b, 0] = torch.prod(1 - sl, dim=0)
            em[b, 1:n_objects[b] + 1] = sl
            at += n_objects[b]
        em = torch.clamp(em, 1e-7, 1 - 1e-7)
        return torch.log(em / (1 - em))

    def _segment_core(self, frame, qry_rects, m_key, m_val, mem_rects, T, n_objects, K, batch_of_obj,
                      obj_begin=None):
        """Everything of ``segment`` after the box bookkeeping: query encoder, fused regional read,
        decoder, soft aggregation, un-pad.  With ``obj_begin`` (device int32 [B+1]) and
        ``fuse_epilogues()`` the tail is one kernel and the return value is (logit, prob)."""
        (frame,), pad = pad_divide_by([frame], 16, frame.shape[2:])
        r4, r3, r2, _, _ = self.encoder_query(frame)
        k4, v4 = self.kv_query(r4)
        if all(int(n) == 1 for n in n_objects):     # one object per clip: object i IS clip i, no expansion copies
            k4e, v4e, r3e, r2e = k4, v4, r3, r2
        else:
            k4e, v4e = k4.index_select(0, batch_of_obj), v4.index_select(0, batch_of_obj)
            r3e, r2e = r3.index_select(0, batch_of_obj), r2.index_select(0, batch_of_obj)
        ev = getattr(self, '_profile_events', None)
        if isinstance(m_key, (ops.MemoryBank, ops.TensorBank)):   # the frame loop: pre-compacted split-fp16 bank
            if T is None:                               # committed frames + the staged one, counted on the device
                m4 = m_key.read_staged(k4e.contiguous(), v4e.contiguous(), qry_rects, events=ev)
            else:
                m4 = m_key.read(T, k4e.contiguous(), v4e.contiguous(), qry_rects, events=ev)
        else:                                       # reference-layout fp32 tensors (public segment())
            m4, _ = ops.memory_read(m_key, m_val, k4e.contiguous(), v4e.contiguous(), mem_rects, qry_rects,
                                    T=T, events=ev)
        if obj_begin is not None and (getattr(self, '_fused_tail', False) and not self.training):
            return ops.soft_aggregate(self.decoder(m4, r3e, r2e).contiguous(), obj_begin, K, pad, want_prob=True)
        ps = F.softmax(self.decoder(m4, r3e, r2e), dim=1)[:, 1]
        logit = self.soft_aggregation(ps, K, n_objects)
        lw, uw, lh, uh = pad
        return logit[:, :, lh:logit.shape[2] - uh, lw:logit.shape[3] - uw]

    @torch.no_grad()
    def segment(self, frame, att_map, keys, values, prev_bboxes, curr_bbox, n_objects):
        """models/rmnet.py:304-383, same arguments.  ``att_map`` is implied by ``curr_bbox`` (it is
        the box map of those boxes) and the memory masking by ``prev_bboxes`` [B,K,T,4]; keys/values
        [B,K,C,T,h,w] may be masked (as ``memorize`` returns them) or not."""
        self._inference_only()
        B, K, _, T, h, w = keys.shape
        H, W = frame.shape[2:]
        lw, _, lh, _ = pad_amounts(H, W, 16)
        flat, batch_of_obj = self._object_index(n_objects, K, frame.device)
        m_key = keys.reshape(B * K, -1, T, h, w).index_select(0, flat).contiguous()
        m_val = values.reshape(B * K, -1, T, h, w).index_select(0, flat).contiguous()
        # memory boxes are in padded-frame coordinates (memorize pads first), query boxes are not
        mem_rects = ops.boxes_to_cell_rects(
            prev_bboxes.reshape(B * K, T, 4).index_select(0, flat).contiguous().int(), 0, 0, 16, h, w)
        qry_rects = ops.boxes_to_cell_rects(
            curr_bbox.reshape(B * K, 4).index_select(0, flat).contiguous().int(), lw, lh, 16, h, w)
        return self._segment_core(frame, qry_rects, m_key, m_val, mem_rects, T, n_objects, K, batch_of_obj)

    # ------------------------------------------------------------------ reference API: the loop
    class _ClipContext:
        """Per-clip constants of the frame loop (object bookkeeping hoisted out of the loop)."""

        def __init__(self, net, B, K, H, W, n_max, device):
            self.B, self.K, self.H, self.W, self.n_max = B, K, H, W, list(n_max)
            self.flat, self.batch_of_obj = net._object_index(self.n_max, K, device)
            lw, uw, lh, uh = pad_amounts(H, W, 16)
            self.lw, self.lh = lw, lh
            self.h, self.w = (H + lh + uh) // 16, (W + lw + uw) // 16
            self.device = device
            begin = [0]
            for n in self.n_max:
                begin.append(begin[-1] + n)
            self.obj_begin = torch.tensor(begin, dtype=torch.int32, device=device)

    def new_bank(self, ctx, capacity, exact=False, precision=None):
        """Pre-allocated regional memory for one clip (replaces models/rmnet.py:191-205, 416-426): the
        split-fp16 ``MemoryBank`` (any number of frames: beyond 2048 the read runs in chunks that are merged by their
        soft-max state), or -- ``exact`` -- plain fp32 tensors read by the exact-fp32 kernel (``TensorBank``)."""
        if exact:
            return ops.TensorBank(len(ctx.flat), capacity, ctx.h, ctx.w, ctx.device)
        return ops.MemoryBank(len(ctx.flat), capacity, ctx.h, ctx.w, ctx.device,
                              precision=precision or self.resolve_read_precision(ctx.n_max))

    def resolve_read_precision(self, n_objects):
        """The arithmetic a clip with ``n_objects`` objects per batch element STARTS in ('auto': see __init__; a one-object clip whose
        logits turn out large is re-read in 'split' at the end of ``forward``, and one-object clips start in 'split' from then on)."""
        if self.read_precision != 'auto':
            return self.read_precision
        return 'f16' if not getattr(self, '_auto_peaked', False) and all(int(n) <= 1 for n in n_objects) else 'split'

    @torch.no_grad()
    def frame_step(self, ctx, bank, prev_frame, prev_mask, cur_frame, cur_flow, commit):
        """One iteration of models/rmnet.py:410-433: memorise frame t-1 (tentatively, or for good
        when ``commit``), derive the regional query boxes from the flow-warped previous mask, segment
        frame t.  Returns the logits [B,K,H,W] -- or, after ``fuse_epilogues()``, the pair
        (logits, soft-max over K of the logits).  No host synchronisation, and no frame-dependent kernel argument
        (the bank's slot / frame count travel as a device counter): with ``commit=False`` the step can be captured
        into a HIP graph once and replayed for every frame (``forward`` does)."""
        self._inference_only()
        B, K = ctx.B, ctx.K
        k4, v4, boxes, rects = self._encode_memory(prev_frame, prev_mask, ctx.n_max)
        bank.stage(k4.contiguous(), v4.contiguous(), rects.view(B * K, 4).index_select(0, ctx.flat))
        if getattr(self, '_fused_tail', False) and self._fused_warp_ok(prev_mask.device):   # warp fused into the box reduction
            _, _, q_rects = ops.region_map(prev_mask.contiguous(), want_map=False, flow=cur_flow.contiguous(),
                                           cell_grid=(ctx.lw, ctx.lh, 16, ctx.h, ctx.w))
        else:
            expt = self.warp(prev_mask, cur_flow)[0]        # models/rmnet.py:429-431
            _, _, q_rects = ops.region_map(expt.contiguous(), want_map=False,
                                           cell_grid=(ctx.lw, ctx.lh, 16, ctx.h, ctx.w))
        q_rects = q_rects.view(B * K, 4).index_select(0, ctx.flat)
        out = self._segment_core(cur_frame, q_rects, bank, None, None, None, ctx.n_max, K, ctx.batch_of_obj,
                                 obj_begin=ctx.obj_begin)
        if commit:          # AFTER the read: the frame count lives on the device (bank.n_dev), the read used committed + 1
            bank.commit()
        return out

    @torch.no_grad()
    def forward(self, frames, masks, optical_flows, n_objects, memorize_every, device=None, _exact=False, graph=None,
                return_logits=False, _precision=None):
        """models/rmnet.py:385-452.  frames [B,N,3,H,W] f32, masks [B,N,K,H,W] (one-hot, any int or
        float dtype), optical_flows [B,N,2,H,W] f32, n_objects [B,N] int -> est_masks [B,N,K,H,W]
        f32 on the GPU (the reference returns them on the host unless several GPUs are visible).
        ``return_logits=True`` returns ``(est_masks, logits)``: the [B,N,K,H,W] logits the masks are the soft-max of (after the
        edits of models/rmnet.py:436-448; frame 0 is zeros) -- what the parity tests compare besides the probabilities.

        ``graph=True``: replay the frame step as ONE captured HIP graph instead of ~340 launches per frame.
        The first segmented frame always runs eagerly (it warms MIOpen up and runs the fused-warp self-check), the graph
        is captured on the second.  The step has no frame-dependent kernel argument -- the memory length lives in a
        device counter -- so one capture serves the whole clip; only the copies into its static input buffers, the commit
        increment and the rare logit edits of models/rmnet.py:436-448 happen outside it.  Default OFF: measured on MI355X
        (bench.py extras) the single 480p stream is bound by the batch-1 convolutions, not by launches -- replay gives
        170 vs 168 frames/s with the memory pinned and LOSES on a free-running 67-frame clip (157 vs 164: the capture and
        the four input copies per frame cost more than the launch gaps they remove)."""
        self._inference_only()
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        frames = frames.to(dev, non_blocking=True)
        optical_flows = optical_flows.to(dev, non_blocking=True)
        masks_dev = masks.to(dev, non_blocking=True)
        B, N, _, H, W = frames.shape
        K = masks.shape[2]
        est = torch.zeros(B, N, K, H, W, device=dev)
        est[:, 0] = masks_dev[:, 0]
        logits = torch.zeros(B, N, K, H, W, device=dev) if return_logits else None   # (frame 0 has none: models/rmnet.py:397)

        # ---- clip-level bookkeeping, hoisted out of the frame loop (the only host syncs)
        n_obj_host = n_objects.cpu()
        n_max = [int(n_obj_host[b].max()) for b in range(B)]
        existing = [torch.unique(torch.argmax(masks_dev[b, 0], dim=0)).tolist() for b in range(B)]
        fresh = {j for j in range(1, N) if bool((n_obj_host[:, j] != n_obj_host[:, j - 1]).any())}
        commit = set(range(0, N, memorize_every)) | fresh
        ctx = self._ClipContext(self, B, K, H, W, n_max, dev)
        bank = self.new_bank(ctx, sum(1 for j in commit if j <= N - 2) + 1, exact=_exact, precision=_precision)

        # (a bank of more than one launch's frames is read in host-planned chunks: its frame count would be baked into the capture)
        use_graph = bool(graph) and isinstance(bank, ops.MemoryBank) and bank.capacity <= ops.BANK_MAX_SLOTS
        replay = None
        for t in range(1, N):
            if use_graph and t >= 2:
                if replay is None:
                    replay = self._capture_frame_step(ctx, bank, frames[:, t - 1], est[:, t - 1], frames[:, t], optical_flows[:, t])
                    if replay is None:
                        use_graph = False
            if use_graph and t >= 2:
                logit = replay(frames[:, t - 1], est[:, t - 1], frames[:, t], optical_flows[:, t])
                if (t - 1) in commit:
                    bank.commit()
            else:
                logit = self.frame_step(ctx, bank, frames[:, t - 1], est[:, t - 1], frames[:, t],
                                        optical_flows[:, t], (t - 1) in commit)
            prob = None
            if isinstance(logit, tuple):
                logit, prob = logit     # fused tail: soft-max already done (valid if the logits stay as they are)
            if t in fresh:      # models/rmnet.py:436-441
                prob = None
                for b in range(B):
                    for j in torch.unique(torch.argmax(masks_dev[b, t], dim=0)).tolist():
                        if j not in existing[b]:
                            existing[b].append(j)
                            logit[b, j] = masks_dev[b, t, j].float() * _NEW_OBJECT_SCALE + _ABSENT_LOGIT
            for b in range(B):  # models/rmnet.py:444-448
                missing = [j for j in range(n_max[b] + 1) if j not in existing[b]]
                if missing:
                    logit[b, missing] = _ABSENT_LOGIT
                    prob = None
            est[:, t] = F.softmax(logit, dim=1) if prob is None else prob
            if return_logits:
                logits[:, t] = logit
        overflow, timeouts, logit_max = bank.status()      # (one host sync per clip)
        self.last_clip = {'read_precision': getattr(bank, 'precision', 'exact'), 'logit_max': logit_max, 'reread': None}
        if overflow:                # K / V / q_key outside the split-fp16 window: redo the clip exactly
            if timeouts:
                import warnings
                warnings.warn('rmnet_amd: %d merge(s) of the bank read timed out on this device (scheduling problem?); '
                              'the clip is re-read with the exact-fp32 kernels' % timeouts)
            out = self.forward(frames, masks, optical_flows, n_objects, memorize_every, device=dev, _exact=True,
                               return_logits=return_logits)
            self.last_clip['reread'] = 'exact: value outside the bank\'s fp16 window'
            return out
        if self.read_precision == 'auto' and self.last_clip['read_precision'] in ('f16', 'qx') and logit_max > AUTO_LOGIT_BOUND:
            # 'auto' on a clip whose soft-max turned out peaked: the fp16-operand read is not inside the bar there (see __init__)
            out = self.forward(frames, masks, optical_flows, n_objects, memorize_every, device=dev, graph=graph,
                               return_logits=return_logits, _precision='split')
            self.last_clip['reread'] = 'split: largest logit %.1f > %.1f' % (logit_max, AUTO_LOGIT_BOUND)
            self._auto_peaked = True        # (sticky: this network's later one-object clips start in 'split')
            return out
        return (est, logits) if return_logits else est

    def _capture_frame_step(self, ctx, bank, prev_frame, prev_mask, cur_frame, cur_flow):
        """Capture ``frame_step(commit=False)`` on static copies of its four inputs; returns ``replay(prev_frame,
        prev_mask, cur_frame, cur_flow) -> frame_step's return value`` (static output tensors: consume them before the
        next replay), or None when this torch / ROCm build cannot capture the step (the loop then stays eager)."""
        dev = prev_frame.device
        saved_events = getattr(self, '_profile_events', None)
        self._profile_events = None                    # hipEventRecord on a capturing stream is not what a profiler wants
        try:
            bufs = [prev_frame.clone(), prev_mask.clone(), cur_frame.clone(), cur_flow.clone()]
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):              # warm-up on a side stream, as torch's capture rules ask
                self.frame_step(ctx, bank, *bufs, commit=False)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self.frame_step(ctx, bank, *bufs, commit=False)
        except RuntimeError as exc:                    # pragma: no cover - depends on the torch / ROCm build
            # only what torch raises for an unsupported / invalidated capture; anything else (a bug in frame_step) propagates.
            # The eager warm-up above ran first: a failure there is not a capture problem and is re-raised.
            torch.cuda.synchronize(dev)                # leave the stream / allocator in a defined state before going on eagerly
            if 'captur' not in str(exc).lower() and 'graph' not in str(exc).lower():
                raise
            import warnings
            warnings.warn('rmnet_amd: HIP graph capture of the frame step failed (%s); running eagerly' % (exc,))
            return None
        finally:
            self._profile_events = saved_events

        def replay(pf, pm, cf, fl):
            bufs[0].copy_(pf)
            bufs[1].copy_(pm)
            bufs[2].copy_(cf)
            bufs[3].copy_(fl)
            g.replay()
            return out
        replay.graph = g
        return replay
