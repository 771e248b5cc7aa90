# -*- coding: utf-8 -*-
"""rmnet_amd -- RMNet's per-frame inference hot path, MI355X-native (gfx950).

Public surface (mirrors the reference's module paths):

    rmnet_amd.rmnet                    <-> models/rmnet.py        (RMNet, MemoryReader, KeyValue, Decoder ...)
    rmnet_amd.tiny_flownet             <-> models/tiny_flownet.py (TinyFlowNet)
    rmnet_amd.reg_att_map_generator    <-> extensions/reg_att_map_generator (+ the compiled module)
    rmnet_amd.flow_affine_transformation <-> the compiled CPython module of the same name
    rmnet_amd.helpers                  <-> utils/helpers.py (pad_divide_by, var_or_cuda, multi_scale_inference)
    rmnet_amd.ops                      torch-facing wrappers over the C ABI (include/rmnet_hip.h)
    rmnet_amd.dist                     per-video sharding over the GPUs of a node + RCCL gather

Importing the package does not load the HIP library; the first op call does, and raises
``rmnet_amd._lib.RMNetHipError`` if it is missing -- there is no CPU path.
"""

__version__ = '0.1.0'
