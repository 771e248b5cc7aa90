# -*- coding: utf-8 -*-
"""Make the REFERENCE's own Python (models/rmnet.py, utils/data_transforms.py) run on these
kernels without editing it: ``import rmnet_amd.compat`` before importing the reference registers

    reg_att_map_generator          (the pybind module, reg_att_map_generator_cuda.cpp:36-38)
    flow_affine_transformation     (the CPython module, flow_affine_transformation.cpp:92-99)

in ``sys.modules``; the reference's ``extensions/reg_att_map_generator/__init__.py:11`` and
``utils/data_transforms.py:18`` then bind to the gfx950 implementations.  See INTEGRATION.md."""

import sys

from . import flow_affine_transformation as _flow
from . import reg_att_map_generator as _ram


def install(force=False):
    for name, mod in (('reg_att_map_generator', _ram), ('flow_affine_transformation', _flow)):
        if force or name not in sys.modules:
            sys.modules[name] = mod


install()
