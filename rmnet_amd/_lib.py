# -*- coding: utf-8 -*-
"""ctypes binding of librmnet_hip.so (the C ABI of include/rmnet_hip.h).

There is deliberately no fallback: if the library is missing or a call returns a non-zero code
the caller gets a RuntimeError.  Nothing here (or anywhere under rmnet_amd/) imports oracle/.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RMNET_HIP_LIB') or os.path.join(_HERE, 'librmnet_hip.so')   # override: experiments only

c_f32p = ctypes.c_void_p   # device pointers travel as integers
c_i32p = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/rmnet_hip.h
SIGNATURES = {
    'rmnet_abi_version': (ctypes.c_int, []),
    'rmnet_error_string': (ctypes.c_char_p, [ctypes.c_int]),
    'rmnet_region_map_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'rmnet_region_map_f32': (ctypes.c_int, [
        c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
        ctypes.c_int, c_f32p, c_i32p, c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'rmnet_region_map_warped_f32': (ctypes.c_int, [
        c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
        ctypes.c_int, c_f32p, c_i32p, c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'rmnet_boxes_to_cell_rects_i32': (ctypes.c_int, [
        c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, c_i32p, ctypes.c_void_p]),
    'rmnet_memory_read_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 7),
    'rmnet_memory_read_f32': (ctypes.c_int, [
        c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong,
        ctypes.c_longlong, c_f32p, c_f32p, c_i32p, c_i32p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_size_t, ctypes.c_void_p]),
    'rmnet_memory_read_f32_ev': (ctypes.c_int, [
        c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong,
        ctypes.c_longlong, c_f32p, c_f32p, c_i32p, c_i32p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'rmnet_bank_bytes': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'rmnet_bank_overflow_offset': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'rmnet_bank_area_offset': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'rmnet_bank_append_f32': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p,
        c_f32p, c_i32p, ctypes.c_void_p]),
    'rmnet_bank_append_f32_at': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p, c_f32p,
        c_f32p, c_i32p, ctypes.c_void_p]),
    'rmnet_bank_read_f32_at': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p, ctypes.c_int, c_f32p,
        c_f32p, c_i32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p]),
    'rmnet_bank_read_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'rmnet_bank_read_workspace_bytes_for': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'rmnet_bank_read_f32': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p,
        c_f32p, c_i32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p]),
    'rmnet_rect_mask_f32': (ctypes.c_int, [
        c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p, c_f32p,
        ctypes.c_void_p]),
    'rmnet_channel_affine_f32': (ctypes.c_int, [
        c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int,
        ctypes.c_longlong, c_f32p, ctypes.c_void_p]),
    'rmnet_upsample2x_add_f32': (ctypes.c_int, [
        c_f32p, c_f32p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    'rmnet_soft_aggregate_f32': (ctypes.c_int, [
        c_f32p, c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p]),
    'rmnet_affine_relu_maxpool_f32': (ctypes.c_int, [
        c_f32p, c_f32p, c_f32p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p,
        ctypes.c_void_p]),
    'rmnet_channel_affine_nhwc_f32': (ctypes.c_int, [
        c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    'rmnet_upsample2x_add_nhwc_f32': (ctypes.c_int, [
        c_f32p, c_f32p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    'rmnet_affine_relu_maxpool_nhwc_f32': (ctypes.c_int, [
        c_f32p, c_f32p, c_f32p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    'rmnet_flow_affine_f32': (ctypes.c_int, [
        c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    'rmnet_flow_affine_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'rmnet_flow_affine_f32_host': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
}

ABI_VERSION = 6
_lib = None


class RMNetHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises RMNetHipError when it is not there -- build it with
    ``python -m rmnet_amd.build`` or ``__graft_entry__.build()``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RMNetHipError(
            'librmnet_hip.so is missing (%s). The HIP kernels are the only implementation of this '
            'path; build them with `python -m rmnet_amd.build`.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.rmnet_abi_version() != ABI_VERSION:
        raise RMNetHipError('librmnet_hip.so ABI %d != expected %d' % (lib.rmnet_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().rmnet_error_string(code).decode()
        raise RMNetHipError('%s failed: %s (code %d)' % (what, msg, code))
