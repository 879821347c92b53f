import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A GPU test on a box without a GPU is an error in how the suite was invoked, not a skip:
    # the driver selects with -m gpu / -m "not gpu".
    pass


class Golden:
    """Lazy reader of tests/golden/<name>.npz returning torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def keys(self):
        return self._z.files

    def np(self, key):
        return self._z[key]

    def __contains__(self, key):
        return key in self._z.files

    def __getitem__(self, key):
        a = self._z[key]
        if a.dtype.kind in "US":
            return a
        return torch.from_numpy(np.array(a))

    def prefixed(self, prefix):
        return {k[len(prefix):]: self[k] for k in self._z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): the per-tensor relative measure of SURVEY.md 8(d)."""
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


import contextlib


@contextlib.contextmanager
def nca_option(opt_name, value):
    """Set a planner option of the library (nca_set_option) for the duration of a block; None leaves it alone."""
    from nerfca_amd import _capi
    opt = getattr(_capi, "OPT_" + opt_name)
    old = _capi.get_option(opt)
    if value is not None:
        _capi.set_option(opt, value)
    try:
        yield
    finally:
        _capi.set_option(opt, old)
