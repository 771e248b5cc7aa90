# -*- coding: utf-8 -*-
"""Convolutional sub-networks of RMNet, kept on stock PyTorch-ROCm (MIOpen).

SURVEY.md section 8 row C1: these stacks are *not* part of the hand-written hot path; they
are declared here only so that (a) the host-side mirror in ``rmnet_amd.rmnet`` has something
to feed the HIP kernels with and (b) public RMNet checkpoints load unchanged.  Every module
and parameter name below therefore matches the reference's state-dict keys:

    reference                                       here
    models/rmnet.py:24-48   ResBlock                ResBlock        (downsample/conv1/conv2)
    models/rmnet.py:51-80   EncoderMemory           EncoderMemory   (conv1_m/conv1_o/conv1/bn1/res2-4)
    models/rmnet.py:83-104  EncoderQuery            EncoderQuery    (conv1/bn1/res2-4)
    models/rmnet.py:107-120 Refine                  Refine          (convFS/ResFS/ResMM)
    models/rmnet.py:123-140 Decoder                 Decoder         (convFM/ResMM/RF3/RF2/pred2)
    models/rmnet.py:168-176 KeyValue                KeyValue        (key_conv/value_conv)

The reference takes its trunk from ``torchvision.models.resnet50`` (models/rmnet.py:57,86);
torchvision is not available in this image, so the ResNet-50 stem and stages 1-3 are declared
here with torchvision's parameter names (``conv1/bn1/conv2/bn2/conv3/bn3/downsample.{0,1}``),
stride on the 3x3 convolution (the "v1.5" variant torchvision ships).
"""

import torch
import torch.nn.functional as F
from torch import nn


class _Bottleneck(nn.Module):
    """ResNet-50 bottleneck: 1x1 reduce, 3x3 (carries the stride), 1x1 expand (x4)."""

    def __init__(self, c_in, width, stride, project):
        super().__init__()
        c_out = width * 4
        self.conv1 = nn.Conv2d(c_in, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, c_out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(c_out)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(c_in, c_out, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(c_out))

    def forward(self, x):
        if getattr(self, '_fused', False) and not self.training and x.is_cuda:
            return self._forward_fused(x)
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + skip)

    def _forward_fused(self, x):
        """Same block with BatchNorm(eval) / skip add / ReLU fused into one pass per convolution
        (rmnet_channel_affine_f32): 3 elementwise kernels instead of 7-8."""
        from . import ops
        t = self.conv1(x)
        ops.channel_affine(t, self._s1, self._b1, relu=True, out=t)
        t = self.conv2(t)
        ops.channel_affine(t, self._s2, self._b2, relu=True, out=t)
        t = self.conv3(t)
        if self.downsample is None:
            return ops.channel_affine(t, self._s3, self._b3, res=x, relu=True, out=t)
        d = self.downsample[0](x)
        return ops.channel_affine(t, self._s3, self._b3, res=d, res_scale=self._sd, res_shift=self._bd,
                                  relu=True, out=t)


def _stage(c_in, width, n_blocks, stride):
    blocks = [_Bottleneck(c_in, width, stride, project=True)]
    blocks += [_Bottleneck(width * 4, width, 1, project=False) for _ in range(n_blocks - 1)]
    return nn.Sequential(*blocks)


class ResNet50Trunk(nn.Module):
    """Stem + layer1..layer3 of ResNet-50 with torchvision attribute names."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = _stage(64, 64, 3, 1)      # 1/4,  256 ch
        self.layer2 = _stage(256, 128, 4, 2)    # 1/8,  512 ch
        self.layer3 = _stage(512, 256, 6, 2)    # 1/16, 1024 ch


def resnet50(pretrained=False):
    """Stand-in for ``torchvision.models.resnet50``; ``pretrained`` is ignored (no network)."""
    return ResNet50Trunk()


class ResBlock(nn.Module):
    """Pre-activation two-conv residual block (models/rmnet.py:24-48)."""

    def __init__(self, indim, outdim=None, stride=1):
        super().__init__()
        outdim = indim if outdim is None else outdim
        self.downsample = None
        if not (indim == outdim and stride == 1):
            self.downsample = nn.Conv2d(indim, outdim, 3, padding=1, stride=stride)
        self.conv1 = nn.Conv2d(indim, outdim, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(outdim, outdim, 3, padding=1)

    def forward(self, x):
        if getattr(self, '_fused', False) and not self.training and x.is_cuda:
            return self._forward_fused(x)
        r = self.conv2(F.relu(self.conv1(F.relu(x))))
        return (x if self.downsample is None else self.downsample(x)) + r

    def _forward_fused(self, x):
        """Same arithmetic: the convolutions run without bias and one pass adds the bias with the
        ReLU (conv1) or with the skip (conv2) -- 3 elementwise kernels instead of 5.  (Not always
        bit-identical: MIOpen may choose another solver for the bias-free convolution.)"""
        from . import ops
        c1, c2 = self.conv1, self.conv2
        t = F.conv2d(F.relu(x), c1.weight, None, c1.stride, c1.padding)
        ops.channel_affine(t, None, c1.bias, relu=True, out=t)
        r = F.conv2d(t, c2.weight, None, c2.stride, c2.padding)
        if self.downsample is None:
            return ops.channel_affine(r, None, c2.bias, res=x, out=r)
        ds = self.downsample
        d = F.conv2d(x, ds.weight, None, ds.stride, ds.padding)
        return ops.channel_affine(r, None, c2.bias, res=d, res_shift=ds.bias, out=r)


class EncoderMemory(nn.Module):
    """Frame + object mask + other-objects mask -> r4 (models/rmnet.py:51-80)."""

    def __init__(self, trunk_factory=resnet50):
        super().__init__()
        self.conv1_m = nn.Conv2d(1, 64, 7, stride=2, padding=3, bias=False)
        self.conv1_o = nn.Conv2d(1, 64, 7, stride=2, padding=3, bias=False)
        trunk = trunk_factory(pretrained=True)
        self.conv1, self.bn1, self.relu, self.maxpool = trunk.conv1, trunk.bn1, trunk.relu, trunk.maxpool
        self.res2, self.res3, self.res4 = trunk.layer1, trunk.layer2, trunk.layer3

    def forward(self, in_f, in_m, in_o):
        m = in_m.unsqueeze(1).float()
        o = in_o.unsqueeze(1).float()
        if getattr(self, '_fused', False) and not self.training and in_f.is_cuda:
            # conv1(f) + conv1_m(m) + conv1_o(o) is ONE 7x7 convolution over the 5 stacked input
            # channels with the three weights stacked the same way: two convolutions and two
            # full-resolution adds fewer, same sum up to fp32 summation order.
            from . import ops
            c = self.conv1
            t = F.conv2d(torch.cat((in_f, m, o), dim=1), self._w5, None, c.stride, c.padding)
            # bn1 -> relu -> maxpool in one pass; the stem activation c1 itself is not materialised
            # (nothing downstream of the encoders reads it) and is returned as None
            c1, pooled = None, ops.affine_relu_maxpool(t, self._s1, self._b1)
        else:
            c1 = self.relu(self.bn1(self.conv1(in_f) + self.conv1_m(m) + self.conv1_o(o)))
            pooled = self.maxpool(c1)
        r2 = self.res2(pooled)
        r3 = self.res3(r2)
        r4 = self.res4(r3)
        return r4, r3, r2, c1, in_f


class EncoderQuery(nn.Module):
    """Frame -> (r4, r3, r2) (models/rmnet.py:83-104)."""

    def __init__(self, trunk_factory=resnet50):
        super().__init__()
        trunk = trunk_factory(pretrained=True)
        self.conv1, self.bn1, self.relu, self.maxpool = trunk.conv1, trunk.bn1, trunk.relu, trunk.maxpool
        self.res2, self.res3, self.res4 = trunk.layer1, trunk.layer2, trunk.layer3

    def forward(self, in_f):
        if getattr(self, '_fused', False) and not self.training and in_f.is_cuda:
            from . import ops
            t = self.conv1(in_f)
            c1, pooled = None, ops.affine_relu_maxpool(t, self._s1, self._b1)   # (c1 not materialised)
        else:
            c1 = self.relu(self.bn1(self.conv1(in_f)))
            pooled = self.maxpool(c1)
        r2 = self.res2(pooled)
        r3 = self.res3(r2)
        r4 = self.res4(r3)
        return r4, r3, r2, c1, in_f


class Refine(nn.Module):
    """Skip-feature fusion + x2 bilinear upsample of the coarser map (models/rmnet.py:107-120)."""

    def __init__(self, inplanes, planes, scale_factor=2):
        super().__init__()
        self.convFS = nn.Conv2d(inplanes, planes, 3, padding=1)
        self.ResFS = ResBlock(planes, planes)
        self.ResMM = ResBlock(planes, planes)
        self.scale_factor = scale_factor

    def forward(self, f, pm):
        s = self.ResFS(self.convFS(f))
        if getattr(self, '_fused', False) and not self.training and s.is_cuda and self.scale_factor == 2:
            from . import ops
            return self.ResMM(ops.upsample2x_add(pm, s, out=s))     # s + up in one pass
        up = F.interpolate(pm, scale_factor=self.scale_factor, mode='bilinear', align_corners=False)
        return self.ResMM(s + up)


class Decoder(nn.Module):
    """[mem | q_val] (1024 ch) + r3 + r2 -> 2-class logits at full resolution
    (models/rmnet.py:123-140)."""

    def __init__(self, mdim):
        super().__init__()
        self.convFM = nn.Conv2d(1024, mdim, 3, padding=1)
        self.ResMM = ResBlock(mdim, mdim)
        self.RF3 = Refine(512, mdim)
        self.RF2 = Refine(256, mdim)
        self.pred2 = nn.Conv2d(mdim, 2, 3, padding=1)

    def forward(self, r4, r3, r2):
        m4 = self.ResMM(self.convFM(r4))
        m3 = self.RF3(r3, m4)
        m2 = self.RF2(r2, m3)
        p2 = self.pred2(F.relu(m2))
        # (channels-last runs: back to NCHW HERE, on the 2-channel quarter-resolution map -- the decoder tail kernel reads NCHW planes, and
        #  converting after the x4 upsample would move 16x the bytes; a no-op for NCHW tensors)
        return F.interpolate(p2.contiguous(), scale_factor=4, mode='bilinear', align_corners=False)


class KeyValue(nn.Module):
    """Two 3x3 heads on r4 (models/rmnet.py:168-176)."""

    def __init__(self, indim, keydim, valdim):
        super().__init__()
        self.key_conv = nn.Conv2d(indim, keydim, 3, padding=1)
        self.value_conv = nn.Conv2d(indim, valdim, 3, padding=1)

    def forward(self, x):
        return self.key_conv(x), self.value_conv(x)


def _hash_uniform(n, seed):
    """Exact integer hash (murmur3 finaliser on idx*2654435761 + seed) -> float64 in [-1, 1).
    Pure int64 tensor arithmetic: bit-identical on every torch version / platform."""
    m32 = 0xFFFFFFFF
    x = (torch.arange(n, dtype=torch.int64) * 2654435761 + seed * 40503 + 0x9E3779B9) & m32
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & m32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & m32
    x = x ^ (x >> 16)
    return x.to(torch.float64) * (2.0 / 4294967296.0) - 1.0


def procedural_init_(module, gain=0.9):
    """Deterministic, RNG-free weight fill used for fixtures and benchmarks (no checkpoint is
    reachable offline).  Every tensor is filled from an integer hash of (flat index, state-dict
    key); convolution weights get He scaling x ``gain``, the last conv of every residual branch
    is damped so activations stay O(1) through ResNet-50 with identity BatchNorm statistics."""
    import math
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if name.endswith('num_batches_tracked'):
                continue
            seed = sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 1000003
            n = t.numel()
            noise = _hash_uniform(n, seed)
            if name.endswith('running_var'):
                val = torch.ones(n, dtype=torch.float64)
            elif name.endswith('running_mean'):
                val = torch.zeros(n, dtype=torch.float64)
            elif t.dim() == 1 and name.endswith('weight'):      # BN gamma
                val = 1.0 + 0.05 * noise
            elif name.endswith('bias'):
                val = 0.05 * noise
            elif t.dim() == 4:
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                if module_is_transposed(module, name):
                    fan_in = t.shape[0] * t.shape[2] * t.shape[3] // 4
                std = gain * math.sqrt(2.0 / max(fan_in, 1))
                if name.endswith('conv3.weight') or name.endswith('conv2.weight') and '.Res' in name:
                    std *= 0.5          # residual-branch output: keep the running sum from doubling
                val = noise * (std * math.sqrt(3.0))
            else:
                val = 0.05 * noise
            t.copy_(val.reshape(t.shape).to(t.dtype))
    return module


def module_is_transposed(module, key):
    mod = module
    for part in key.split('.')[:-1]:
        mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
    return isinstance(mod, nn.ConvTranspose2d)


# ----------------------------------------------------------------------------------------------
# Inference-time BatchNorm folding (eval mode only).  The reference keeps conv -> BN -> ReLU as three
# kernels (torchvision ResNet-50, models/rmnet.py:66-80, 96-103); in eval mode BN is the affine map
# y = (x - mean) * gamma / sqrt(var + eps) + beta, which folds into the preceding convolution's
# weights and bias.  Mathematically identical, differs only by fp32 rounding (covered by the
# end-to-end parity tests); removes ~2500 BatchNorm launches per 39 frames (6 % of GPU time).
# ----------------------------------------------------------------------------------------------
class _Identity(nn.Module):
    def forward(self, x):
        return x


def _bn_scale_shift(bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale, bn.bias - bn.running_mean * scale


def _fold(conv, bn):
    scale, shift = _bn_scale_shift(bn)
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                      conv.dilation, conv.groups, bias=True).to(conv.weight.device, conv.weight.dtype)
    fused.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
    fused.bias.copy_(shift if conv.bias is None else conv.bias * scale + shift)
    return fused


@torch.no_grad()
def fold_batchnorm_(module):
    """Fold every conv->BN pair of the ResNet-50 trunks of ``module`` (an ``RMNet`` or one of its
    encoders) in place.  Call after loading weights, in eval mode.  The state dict of the folded
    module no longer matches the reference's (keep an un-folded copy if you need to save it)."""
    for m in module.modules():
        if isinstance(m, _Bottleneck):
            m.conv1, m.bn1 = _fold(m.conv1, m.bn1), _Identity()
            m.conv2, m.bn2 = _fold(m.conv2, m.bn2), _Identity()
            m.conv3, m.bn3 = _fold(m.conv3, m.bn3), _Identity()
            if m.downsample is not None:
                m.downsample = nn.Sequential(_fold(m.downsample[0], m.downsample[1]), _Identity())
        elif isinstance(m, EncoderQuery) and isinstance(m.bn1, nn.BatchNorm2d):
            m.conv1, m.bn1 = _fold(m.conv1, m.bn1), _Identity()
        elif isinstance(m, EncoderMemory) and isinstance(m.bn1, nn.BatchNorm2d):
            # bn1(conv1(f) + conv1_m(m) + conv1_o(o)): scale all three, shift once
            scale, shift = _bn_scale_shift(m.bn1)
            fused = _fold(m.conv1, m.bn1)
            m.conv1_m.weight.mul_(scale.view(-1, 1, 1, 1))
            m.conv1_o.weight.mul_(scale.view(-1, 1, 1, 1))
            m.conv1, m.bn1 = fused, _Identity()
    return module


# ----------------------------------------------------------------------------------------------
# Fused elementwise epilogues (eval mode, GPU only).  Unlike fold_batchnorm_ this keeps every
# parameter and the state dict untouched: the BatchNorm statistics are only re-expressed as a
# per-channel (scale, shift) pair held in non-persistent buffers, and the forward passes of
# _Bottleneck / ResBlock / the encoder stems call rmnet_channel_affine_f32 once per convolution
# instead of BatchNorm -> add -> ReLU (or bias -> ReLU / bias -> add) as separate kernels.
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def fuse_epilogues_(module, enable=True):
    """Switch ``module`` (an ``RMNet`` or any sub-module) to the fused elementwise path; call after
    loading weights, in eval mode.  ``enable=False`` switches back."""
    def put(m, name, t):
        if name in m._buffers:
            m._buffers[name] = t
        else:
            m.register_buffer(name, t, persistent=False)

    if enable and any(isinstance(m, _Identity) for m in module.modules()):
        raise RuntimeError('fuse_epilogues() after fuse_for_inference(): the BatchNorms are already folded into '
                           'the convolutions -- use one or the other')
    def snapshot(m):
        """(Re)compute the (scale, shift) pairs / the stacked stem of ONE module from its current parameters."""
        if isinstance(m, _Bottleneck):
            for i, bn in ((1, m.bn1), (2, m.bn2), (3, m.bn3)):
                sc, sh = _bn_scale_shift(bn)
                put(m, '_s%d' % i, sc.contiguous())
                put(m, '_b%d' % i, sh.contiguous())
            if m.downsample is not None:
                sc, sh = _bn_scale_shift(m.downsample[1])
                put(m, '_sd', sc.contiguous())
                put(m, '_bd', sh.contiguous())
        elif isinstance(m, (EncoderMemory, EncoderQuery)):
            sc, sh = _bn_scale_shift(m.bn1)
            put(m, '_s1', sc.contiguous())
            put(m, '_b1', sh.contiguous())
            if isinstance(m, EncoderMemory):
                put(m, '_w5', torch.cat((m.conv1.weight, m.conv1_m.weight, m.conv1_o.weight), dim=1).contiguous())

    def _refresh(mod, incompatible):
        # load_state_dict() runs the post hooks of EVERY module it descends into, so a hook on each snapshot owner
        # covers net.load_state_dict(...) as well as net.encoder_query.load_state_dict(...) or a single block's.
        # (Editing a parameter in place is not observable: call fuse_epilogues() again after doing that.)
        if getattr(mod, '_fused', False):
            with torch.no_grad():
                snapshot(mod)

    for m in module.modules():
        if isinstance(m, (_Bottleneck, EncoderMemory, EncoderQuery)):
            snapshot(m)
            handle = getattr(m, '_fuse_hook', None)
            if enable and handle is None:
                m._fuse_hook = m.register_load_state_dict_post_hook(_refresh)
            elif not enable and handle is not None:
                handle.remove()
                m._fuse_hook = None
            m._fused = bool(enable)
        elif isinstance(m, (ResBlock, Refine)):
            m._fused = bool(enable)
    return module
