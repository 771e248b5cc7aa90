# -*- coding: utf-8 -*-
"""Host helpers on the hot path; mirrors the three functions of
the reference's ``utils/helpers.py`` that RMNet's per-frame loop calls."""

import torch
import torch.nn.functional as F


def pad_amounts(h, w, d):
    """(lw, uw, lh, uh) that bring (h, w) to multiples of ``d`` with the reference's split
    (utils/helpers.py:105-121): lower = int(delta / 2), upper = delta - lower."""
    dh = (d - h % d) % d
    dw = (d - w % d) % d
    lh, lw = dh // 2, dw // 2
    return lw, dw - lw, lh, dh - lh


def pad_divide_by(in_list, d, in_size):
    """Zero-pad every tensor of ``in_list`` (last two dims = ``in_size``) to a multiple of
    ``d``.  Returns (padded list, (lw, uw, lh, uh)) exactly like utils/helpers.py:105-124."""
    pad = pad_amounts(int(in_size[0]), int(in_size[1]), d)
    return [F.pad(x, pad) for x in in_list], pad


def var_or_cuda(x, device=None):
    """utils/helpers.py:16-24: contiguous + move to the GPU unless ``device`` is the CPU."""
    x = x.contiguous()
    if torch.cuda.is_available() and (device is None or torch.device(device).type != 'cpu'):
        x = x.cuda(non_blocking=True) if device is None else x.cuda(device=device, non_blocking=True)
    return x


def multi_scale_inference(cfg, tflownet, rmnet, frames, masks, n_objects):
    """utils/helpers.py:44-78.  ``cfg.TEST`` needs FRAME_SCALES, FLIP_LR, MEMORIZE_EVERY."""
    _, n, c, h, w = frames.shape
    # the reference's networks are DataParallel-wrapped, which moves the loader's host tensors to the GPU
    # (core/inference.py:35-37); here the inputs are moved once, up front
    dev = next(rmnet.parameters()).device
    frames, masks = var_or_cuda(frames, dev), var_or_cuda(masks, dev)
    est_flows, est_probs = [], []
    for fs in cfg.TEST.FRAME_SCALES:
        fr = F.interpolate(frames[0], scale_factor=fs, mode='bilinear', align_corners=False).unsqueeze(0)
        mk = F.interpolate(masks[0].float(), scale_factor=fs, mode='nearest').int().unsqueeze(0)
        fl = tflownet(fr)
        pr = rmnet(fr, mk, fl, n_objects, cfg.TEST.MEMORIZE_EVERY)
        est_flows.append(F.interpolate(fl[0], size=(h, w), mode='bilinear',
                                       align_corners=False).unsqueeze(0) / fs)
        est_probs.append(F.interpolate(pr[0], size=(h, w), mode='bilinear',
                                       align_corners=False).unsqueeze(0))
        if cfg.TEST.FLIP_LR:
            fr_f, mk_f = torch.flip(fr, dims=[4]), torch.flip(mk, dims=[4])
            fl_f = torch.flip(fl, dims=[4]).clone()
            fl_f[:, :, 0] = -fl_f[:, :, 0]
            pr_f = torch.flip(rmnet(fr_f, mk_f, fl_f, n_objects, cfg.TEST.MEMORIZE_EVERY), dims=[4])
            est_probs.append(F.interpolate(pr_f[0], size=(h, w), mode='bilinear',
                                           align_corners=False).unsqueeze(0))
    return torch.mean(torch.stack(est_flows), dim=0), torch.mean(torch.stack(est_probs), dim=0)
