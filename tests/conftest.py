import json
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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden(object):
    """Read access to one tests/golden/golden_*.npz fixture."""

    def __init__(self, part):
        self.z = np.load(os.path.join(GOLDEN, "golden_%s.npz" % part), allow_pickle=False)
        self.meta = json.loads(bytes(self.z["meta"]).decode())

    def __getitem__(self, name):
        return self.z[name]

    def has(self, name):
        return (name + ".sample") in self.z

    def sample(self, name):
        return torch.from_numpy(self.z[name + ".sample"]), int(self.z[name + ".stride"])

    def compare(self, name, tensor, atol, rtol=0.0):
        """max |tensor - golden| on the stored (sub-sampled) points, plus the checksum."""
        ref, stride = self.sample(name)
        t = tensor.detach().cpu().float()
        assert tuple(t.shape) == tuple(self.z[name + ".shape"]), (name, t.shape)
        got = t[..., ::stride, ::stride] if stride > 1 else t
        err = (got - ref).abs().max().item()
        bound = atol + rtol * ref.abs().max().item()
        assert err <= bound, "%s: max abs err %.3e > %.3e" % (name, err, bound)
        ref_abs = float(self.z[name + ".abssum"])
        got_abs = t.double().abs().sum().item()
        assert abs(got_abs - ref_abs) <= max(1e-3 * ref_abs, bound * t.numel()), \
            "%s: |.|-checksum %.6e vs golden %.6e" % (name, got_abs, ref_abs)
        return err

    def arrays(self, prefix):
        return {k[len(prefix):]: self.z[k] for k in self.z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden_ops():
    return Golden("ops")


@pytest.fixture(scope="session")
def golden_blocks():
    return Golden("blocks")


@pytest.fixture(scope="session")
def golden_e2e():
    return Golden("e2e")


@pytest.fixture(scope="session")
def golden_train():
    return Golden("train")


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library (built here if the .so is missing; hipcc cross-compiles)."""
    from dsmnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from dsmnet_amd.csrc import build
        build.build(verbose=False)
    return _lib.load()
