import ast
import glob
import os

# host-side tests exercise the differentiable torch composites on CPU tensors (state_dict / autograd semantics without a
# GPU); the product refuses CPU tensors unless this is set (see _native.torch_composite_allowed)
os.environ.setdefault("MAGAT_ALLOW_TORCH_COMPOSITE", "1")
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_paths(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_layer_fixture(path):
    import torch
    z = np.load(path, allow_pickle=False)
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p_")}
    return z, p


def load_model_fixture(path):
    import torch
    z = np.load(path, allow_pickle=False)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    cfg = types.SimpleNamespace(**ast.literal_eval(str(z["cfg"])))
    return z, sd, cfg


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


class _LibOpt:
    """Library options (magat_set_option) changed for one test and restored afterwards - the library reads the
    environment only once, so tests flip its switches through the C ABI."""

    def __init__(self):
        from magat_pathplanning_amd import _native
        self.nat = _native
        self.touched = set()

    def set(self, name, value):
        self.nat.set_option(name, int(value))
        self.touched.add(name)

    def reset(self, name):
        self.nat.reset_option(name)
        self.touched.discard(name)

    def restore(self):
        for n in list(self.touched):
            self.nat.reset_option(n)
        self.touched.clear()


@pytest.fixture
def libopt():
    o = _LibOpt()
    yield o
    o.restore()


class _TagCounts:
    """Launch counts per profiling tag (magat_profile_*; tags in _native.TAGS) over a `with` block: lets a parity test assert
    WHICH kernel produced the numbers it compared (e.g. the one-launch graph layer, tag 19), so that a silent hand-over to
    another form of the same layer cannot keep a test green."""

    def __init__(self):
        from magat_pathplanning_amd import _native
        self.nat = _native
        self.counts = {}

    def __enter__(self):
        lib = self.nat.lib()
        lib.magat_profile_reset()
        lib.magat_profile_enable(1)
        return self

    def __exit__(self, *exc):
        import ctypes
        import torch
        lib = self.nat.lib()
        torch.cuda.synchronize()
        lib.magat_profile_enable(0)
        lib.magat_profile_collect()
        self.counts = {}
        for tag, name in self.nat.TAGS.items():
            c, ms = ctypes.c_longlong(0), ctypes.c_double(0)
            lib.magat_profile_read(tag, ctypes.byref(c), ctypes.byref(ms))
            if c.value:
                self.counts[name] = c.value
        lib.magat_profile_reset()
        return False

    def __getitem__(self, name):
        return self.counts.get(name, 0)


@pytest.fixture
def tag_counts():
    """Factory: `with tag_counts() as tc: ...; tc["gat_layer (one launch)"]`."""
    return _TagCounts
