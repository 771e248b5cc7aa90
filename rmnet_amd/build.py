# -*- coding: utf-8 -*-
"""Build librmnet_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch build machinery:
the boundary is a plain C ABI, see include/rmnet_hip.h)."""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['capi.hip', 'region_map.hip', 'flow_affine.hip', 'memory_read.hip', 'bank.hip', 'epilogue.hip']
LIB = os.path.join(HERE, 'librmnet_hip.so')
ARCH = 'gfx950'


def hipcc_path():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: librmnet_hip.so cannot be built')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(CSRC, 'common.h'),
        os.path.join(os.path.dirname(HERE), 'include', 'rmnet_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source into one shared library.  Returns the library path."""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-shared', '-fPIC',
           '-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
