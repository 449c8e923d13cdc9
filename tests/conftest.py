import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def switches():
    """``switches(NAME, value)``: set a run-time switch of the library (csrc/switches.h) for the rest of the test; every switch
    touched is restored afterwards.  (The environment is read once, when the library is loaded: a test cannot setenv.)"""
    from fcd_gan_pytorch_amd import _lib
    saved = {}

    def set_(name, value):
        old = _lib.set_switch(name, value)
        saved.setdefault(name, old)
    yield set_
    for k, v in saved.items():
        _lib.set_switch(k, v)


@pytest.fixture(params=['direct', 'winograd'])
def conv_path(request):
    """Run a gradient-level test once on the direct MFMA kernels only and once with the library's default
    layer plan (Winograd F(4x4, 3x3) for the 3x3 layers with >= 128 GEMM rows).  The Winograd transforms
    carry ~1e-5 of fp32 rounding (direct: ~1e-6): forward values stay inside the same bounds, but more
    ReLU / max-pool decisions within rounding distance of the kink fall the other way, so aggregate
    gradient bounds are doubled on that path."""
    from fcd_gan_pytorch_amd import _lib
    prev = _lib.lib.fcd_conv_wino_set(0 if request.param == 'direct' else 4)
    yield request.param
    _lib.lib.fcd_conv_wino_set(prev)
