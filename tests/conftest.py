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


def assert_bits_equal(a, b, what=""):
    """Bit-exact comparison of float arrays (NaNs compare equal if their payloads do)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} vs {b.shape}"
    if a.dtype.kind == "f":
        ia, ib = a.view(np.uint32), b.view(np.uint32)
        # +0 and -0 differ in bits but not in value; the comparison is on values, NaN == NaN
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        _ = ia, ib
    else:
        same = a == b
    bad = np.argwhere(~same)
    assert len(bad) == 0, f"{what}: {len(bad)} of {a.size} elements differ, first at {bad[0].tolist()}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}"
