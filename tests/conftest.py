import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def scenes():
    from pathtracer_amd.scene import SceneData
    return {name: SceneData.load(os.path.join(GOLDEN, f"{name}_scene.npz")) for name in ("cornell", "mixed")}


@pytest.fixture(scope="session")
def golden():
    return {name: dict(np.load(os.path.join(GOLDEN, f"golden_{name}.npz"))) for name in ("cornell", "mixed", "leaf", "bvh", "converged")}


@pytest.fixture(scope="session")
def have_reference():
    from oracle import ref
    return ref.available()


def assert_bits_equal(a, b, what="", allow_signed_zero=False):
    """Bit-exact comparison: float arrays are compared as their bit patterns, so +0 differs from -0 and a NaN
    equals only a NaN with the same sign and payload.  allow_signed_zero=True lets +0 match -0 (and nothing else)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} vs {b.shape}"
    assert a.dtype == b.dtype, f"{what}: dtypes {a.dtype} vs {b.dtype}"
    if a.dtype.kind == "f":
        bits = {2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
        same = a.view(bits) == b.view(bits)
        if allow_signed_zero:
            same |= (a == 0) & (b == 0)
    else:
        same = a == b
    bad = np.argwhere(~same)
    assert len(bad) == 0, f"{what}: {len(bad)} of {a.size} elements differ, first at {bad[0].tolist()}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}"
