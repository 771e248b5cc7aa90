# -*- coding: utf-8 -*-
"""The C-ABI library builds, loads on a machine without a GPU and exports every symbol that
include/rmnet_hip.h declares (no compute calls here)."""

import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'rmnet_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rmnet_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_every_header_symbol():
    from rmnet_amd import _lib, build
    path = build.build_library()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    syms = _header_symbols()
    assert len(syms) >= 11, syms
    for s in syms:
        assert hasattr(lib, s), 'missing export: ' + s
    assert sorted(_lib.SIGNATURES.keys()) == syms     # the ctypes table covers the whole header


def test_error_strings_and_sizes_need_no_gpu():
    from rmnet_amd import _lib
    lib = _lib.load()
    assert lib.rmnet_abi_version() == 6
    assert lib.rmnet_error_string(0) == b'ok'
    for code in (-1, -2, -3, -4):
        assert len(lib.rmnet_error_string(code)) > 4
    # workspace queries are pure host arithmetic
    assert lib.rmnet_memory_read_workspace_bytes(1, 128, 512, 5, 30, 54, 0) > 0
    assert lib.rmnet_memory_read_workspace_bytes(0, 128, 512, 5, 30, 54, 0) == 0
    assert lib.rmnet_region_map_workspace_bytes(1, 2, 480, 864) > 0
    assert lib.rmnet_flow_affine_workspace_bytes(480, 854) >= 2 * 480 * 854 * 8


def test_invalid_arguments_are_reported_not_launched():
    from rmnet_amd import _lib
    lib = _lib.load()
    assert lib.rmnet_flow_affine_f32(None, None, None, 4, 4, None, None) == -1
    assert lib.rmnet_region_map_f32(None, 1, 2, 8, 8, 0.5, 10, 64, None, None, None, 0, 0, 16, 1, 1,
                                    None, 0, None) == -1
    assert lib.rmnet_memory_read_f32(None, None, None, None, 1, 128, 512, 1, 4, 4, 0, 0, 0, 0, None, None,
                                     None, None, 0, None, 0, None) == -1
    assert lib.rmnet_channel_affine_f32(None, None, None, None, None, None, 1, 1, 4, 16, None, None) == -1
    assert lib.rmnet_upsample2x_add_f32(None, None, 1, 4, 8, 8, None, None) == -1
    assert lib.rmnet_soft_aggregate_f32(None, None, 1, 2, 16, 16, 0, 0, 16, 16, None, None, None) == -1
    assert lib.rmnet_region_map_warped_f32(None, None, 1, 2, 8, 8, 0.5, 10, 64, None, None, None, 0, 0, 16, 1, 1,
                                           None, None, 0, None) == -1


def test_ops_reject_cpu_tensors_like_the_reference():
    """reg_att_map_generator_cuda.cpp:14-19 -> RuntimeError for non-CUDA / non-contiguous input."""
    import torch
    from rmnet_amd import ops
    from rmnet_amd.reg_att_map_generator import RegionalAttentionMapGenerator
    with pytest.raises(RuntimeError, match='CUDA'):
        RegionalAttentionMapGenerator()(torch.zeros(1, 2, 8, 8))
    with pytest.raises(RuntimeError, match='CUDA'):
        ops.memory_read(torch.zeros(1, 128, 1, 2, 2), torch.zeros(1, 512, 1, 2, 2),
                        torch.zeros(1, 128, 2, 2), torch.zeros(1, 512, 2, 2))
    with pytest.raises(RuntimeError, match='CUDA'):
        ops.channel_affine(torch.zeros(1, 4, 2, 2), torch.ones(4), torch.zeros(4))
    with pytest.raises(RuntimeError, match='CUDA'):
        ops.upsample2x_add(torch.zeros(1, 4, 2, 2))
    with pytest.raises(RuntimeError, match='CUDA'):
        ops.soft_aggregate(torch.zeros(1, 2, 16, 16), torch.zeros(2, dtype=torch.int32), 2, (0, 0, 0, 0))
    with pytest.raises(RuntimeError, match='CUDA'):
        ops.region_map(torch.zeros(1, 2, 8, 8), flow=torch.zeros(1, 2, 8, 8))


def test_flag_constants_of_the_python_mirror_match_the_header():
    """ops.MR_* / ops.BANK_* are the header's RMNET_MR_* / RMNET_BANK_* (ABI v3 added the fp16-operand switch), and the
    precision knob of the mirror only takes the two documented values."""
    from rmnet_amd import _lib, ops
    from rmnet_amd.rmnet import MemoryReader, RMNet
    src = open(os.path.join(ROOT, 'include', 'rmnet_hip.h')).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r'^#define\s+(RMNET_\w+)\s+(-?\d+)\b', src, re.M)}
    assert defs['RMNET_ABI_VERSION'] == _lib.ABI_VERSION == 6
    assert defs['RMNET_MR_FORCE_GENERIC'] == ops.MR_FORCE_GENERIC and defs['RMNET_MR_EXACT_FP32'] == ops.MR_EXACT_FP32
    assert defs['RMNET_MR_F16'] == ops.MR_F16 == defs['RMNET_BANK_F16'] == ops.BANK_F16
    assert len({defs['RMNET_MR_FORCE_GENERIC'], defs['RMNET_MR_EXACT_FP32'], defs['RMNET_MR_F16']}) == 3   # distinct bits
    assert ops._precision('split') == 'split' and ops._precision('f16') == 'f16'
    with pytest.raises(ValueError):
        ops._precision('bf16')
    with pytest.raises(ValueError):
        RMNet(None, read_precision='fp8')
    net = RMNet(None)
    assert net.read_precision == 'auto' and MemoryReader().precision == 'split'      # the stand-alone reader: fp32-class unless asked
    assert net.resolve_read_precision([1, 1, 1]) == 'f16' and net.resolve_read_precision([1, 3]) == 'split'   # per-clip choice (profiles/r05_iou_calibration.md, r06_iou_temperature.md)
    assert RMNet(None, read_precision='f16').resolve_read_precision([5]) == 'f16'
    lib = _lib.load()
    assert defs['RMNET_MR_QX'] == ops.MR_QX == defs['RMNET_BANK_QX'] == ops.BANK_QX
    assert len({defs['RMNET_MR_FORCE_GENERIC'], defs['RMNET_MR_EXACT_FP32'], defs['RMNET_MR_F16'], defs['RMNET_MR_QX']}) == 4   # distinct bits
    # RMNET_BANK_F16 and RMNET_BANK_QX are the only flags rmnet_bank_read_f32_at knows, and they exclude each other: an unknown bit (16)
    # or both together are refused before any launch (with NULL pointers every call is refused anyway: tests/test_gpu_parity.py
    # test_qx_mode_on_large_logits_and_flag_rules makes the same two calls with valid arguments on the GPU)
    assert lib.rmnet_bank_read_f32_at(None, 1, 1, 4, 4, 1, None, 16, None, None, None, None, None, 0, None, None, None, None) == -1
    assert lib.rmnet_bank_read_f32_at(None, 1, 1, 4, 4, 1, None, 4 | 8, None, None, None, None, None, 0, None, None, None, None) == -1
