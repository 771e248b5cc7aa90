# -*- coding: utf-8 -*-
"""Drop-in for the reference's compiled CPython module ``flow_affine_transformation``
(extensions/flow_affine_transformation/flow_affine_transformation.cpp:87-99): one function,
``update_optical_flow(optical_flow, tr_matrix1, tr_matrix2) -> ndarray`` with NumPy arrays in and a
new NumPy array out (called from utils/data_transforms.py:298-299).  The arithmetic runs in
csrc/flow_affine.hip and is bit-identical to the reference's C++.
``update_optical_flow_cuda`` is the device-resident variant (torch tensors, no host round trip)."""

from .ops import flow_affine as update_optical_flow_cuda  # noqa: F401
from .ops import update_optical_flow  # noqa: F401
