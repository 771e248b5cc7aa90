# -*- coding: utf-8 -*-
"""TinyFlowNet on stock PyTorch-ROCm (SURVEY.md section 8 row C2 -- not re-implemented in HIP).

Mirrors ``models/tiny_flownet.py`` of the reference: same constructor signature, same
``forward(frames) -> [B, N, 2, H, W]`` contract and the same state-dict keys
(``conv1.0.weight`` ... ``upsampled_flow3_to_2.weight``, reference lines 20-82) so that the
``'tflownet'`` entry of a public checkpoint loads unchanged.  The only behavioural change is
that the output buffer is allocated on the input's device (the reference allocates it on the
host and copies every frame back, lines 121-132) -- values are identical.
"""

import torch
import torch.nn.functional as F
from torch import nn

from .helpers import pad_divide_by


def _conv(c_in, c_out, k, stride=1):
    return nn.Sequential(nn.Conv2d(c_in, c_out, k, stride=stride, padding=k // 2),
                         nn.LeakyReLU(0.1, inplace=True))


def _deconv(c_in, c_out):
    return nn.Sequential(nn.ConvTranspose2d(c_in, c_out, 4, stride=2, padding=1, bias=True),
                         nn.LeakyReLU(0.1, inplace=True))


def _flow_head(c_in):
    return nn.Conv2d(c_in, 2, 3, padding=1, bias=True)


def _flow_up():
    return nn.ConvTranspose2d(2, 2, 4, stride=2, padding=1, bias=False)


class TinyFlowNet(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.conv1 = _conv(6, 64, 7, 2)
        self.conv2 = _conv(64, 128, 5, 2)
        self.conv3 = _conv(128, 256, 5, 2)
        self.conv3_1 = _conv(256, 256, 3)
        self.conv4 = _conv(256, 512, 3, 2)
        self.conv4_1 = _conv(512, 512, 3)
        self.conv5 = _conv(512, 512, 3, 2)
        self.conv5_1 = _conv(512, 512, 3)
        self.deconv4 = _deconv(512, 256)
        self.deconv3 = _deconv(770, 128)
        self.deconv2 = _deconv(386, 64)
        self.predict_flow5 = _flow_head(512)
        self.predict_flow4 = _flow_head(770)
        self.predict_flow3 = _flow_head(386)
        self.predict_flow2 = _flow_head(194)
        self.upsampled_flow5_to_4 = _flow_up()
        self.upsampled_flow4_to_3 = _flow_up()
        self.upsampled_flow3_to_2 = _flow_up()

    def _forward(self, img0, img1):
        """Flow for one frame pair (reference lines 84-119): pad to /64, halve, encode,
        three refinement levels, x8 bilinear upsample, un-pad."""
        (img0, img1), pad = pad_divide_by([img0, img1], 64, img0.shape[2:])
        pair = torch.cat((F.interpolate(img0, scale_factor=0.5, mode='bilinear'),
                          F.interpolate(img1, scale_factor=0.5, mode='bilinear')), dim=1)
        run = self._fused_block if getattr(self, '_fused', False) and not self.training and pair.is_cuda else (lambda m, x: m(x))
        c2 = run(self.conv2, run(self.conv1, pair))
        c3 = run(self.conv3_1, run(self.conv3, c2))
        c4 = run(self.conv4_1, run(self.conv4, c3))
        c5 = run(self.conv5_1, run(self.conv5, c4))

        cat4 = torch.cat((c4, run(self.deconv4, c5), self.upsampled_flow5_to_4(self.predict_flow5(c5))), 1)
        cat3 = torch.cat((c3, run(self.deconv3, cat4), self.upsampled_flow4_to_3(self.predict_flow4(cat4))), 1)
        cat2 = torch.cat((c2, run(self.deconv2, cat3), self.upsampled_flow3_to_2(self.predict_flow3(cat3))), 1)
        flow = F.interpolate(self.predict_flow2(cat2), scale_factor=8, mode='bilinear')

        lw, uw, lh, uh = pad
        if lh + uh > 0:
            flow = flow[:, :, lh:flow.shape[2] - uh, :]
        if lw + uw > 0:
            flow = flow[:, :, :, lw:flow.shape[3] - uw]
        return flow

    def load_reference_state_dict(self, state_dict, strict=True):
        """Accepts the reference's checkpoints with or without DataParallel's ``module.`` prefix
        (core/inference.py:33-44)."""
        clean = {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        return self.load_state_dict(clean, strict=strict)

    @staticmethod
    def _fused_block(block, x):
        """conv / deconv (+bias) -> LeakyReLU(0.1) with the bias and the activation in ONE pass
        (rmnet_channel_affine_f32, act 2) instead of two elementwise kernels."""
        from . import ops
        conv = block[0]
        if isinstance(conv, nn.ConvTranspose2d):
            t = F.conv_transpose2d(x, conv.weight, None, conv.stride, conv.padding)
        else:
            t = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
        return ops.channel_affine(t, None, conv.bias, relu='leaky', out=t)

    def fuse_epilogues(self, enable=True):
        """Bias + LeakyReLU of every block as one kernel (parameters untouched)."""
        self.eval()
        self._fused = bool(enable)
        return self

    def forward(self, frames, device=None):
        # (the reference's DataParallel wrapper moves host frames to the GPU, core/inference.py:35-37)
        frames = frames.to(self.conv1[0].weight.device, non_blocking=True)
        b, n, _, h, w = frames.shape
        flows = frames.new_zeros(b, n, 2, h, w)
        for t in range(1, n):
            flows[:, t] = self._forward(frames[:, t], frames[:, t - 1])
        return flows
