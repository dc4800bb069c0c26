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
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class Fixture:
    """npz fixture with torch access: fx['name'] -> tensor, fx.sub('prefix.') -> dict."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def __contains__(self, k):
        return k in self._z.files

    def __getitem__(self, k):
        a = self._z[k]
        if a.dtype.kind in "US":
            return a
        if a.dtype == np.uint64:          # exact integer checksums
            return int(a) if a.ndim == 0 else [int(v) for v in a.reshape(-1)]
        return torch.from_numpy(np.array(a))

    def keys(self):
        return list(self._z.files)

    def sub(self, prefix):
        return {k[len(prefix):]: self[k] for k in self._z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Fixture(name)
        return cache[name]

    return load


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())
