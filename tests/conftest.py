# -*- coding: utf-8 -*-
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# channels_last tensors go to MIOpen as NHWC (bench.py --channels-last sets the same; read by ATen at the first convolution)
os.environ.setdefault('PYTORCH_MIOPEN_SUGGEST_NHWC', '1')
if os.path.join(ROOT, 'tests') not in sys.path:       # (tests/live_fixture.py is imported by name)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def oracle_mod():
    """The CPU checker (test infrastructure).  Builds liboracle.so on first use."""
    from oracle import oracle
    oracle.build()
    return oracle
