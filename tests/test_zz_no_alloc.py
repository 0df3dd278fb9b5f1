"""pt_trace_closest_static, the entry point that allocates no device memory, against pt_trace_closest through the
C-ABI: hit primitive, t, u, v and the per-ray counters bit for bit, on two calls in a row, the second with a
smaller batch (the staging array keeps the first call's contents), then on a second scene that shares the GPU's
array, and the size limit."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from pathtracer_amd import abi, api

pytestmark = pytest.mark.gpu


def rays(scene, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
    o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    return o, d


def same(got, want, what):
    assert np.array_equal(got["prim"], want["prim"]), what
    for k in ("t", "u", "v"):
        assert_bits_equal(got[k], want[k], f"{what}: {k}")
    assert np.array_equal(got["nodes"], want["nodes"]) and np.array_equal(got["prim_tests"], want["prim_tests"]), what


def test_static_staging_equals_allocating_entry_point(scenes):
    cornell, mixed = scenes["cornell"], scenes["mixed"]
    with api.DeviceScene(cornell.to_desc()) as dev, api.DeviceScene(mixed.to_desc()) as other:
        o, d = rays(cornell, abi.PT_TRACE_STATIC_MAX_RAYS, 1)          # the largest batch the array holds
        first = dev.trace_closest_static(o, d)
        same(first, dev.trace_closest(o, d), "full batch")
        assert (first["prim"] >= 0).mean() > 0.3
        o2, d2 = rays(cornell, 1000, 2)                                # smaller: stale results lie behind it
        import torch
        free_before = torch.cuda.mem_get_info(0)[0]
        second = dev.trace_closest_static(o2, d2)
        assert torch.cuda.mem_get_info(0)[0] == free_before, "the static entry point changed the GPU's free memory"
        same(second, dev.trace_closest(o2, d2), "second, smaller batch")
        o3, d3 = rays(mixed, 5000, 3)                                  # another scene on the same GPU, same array
        same(other.trace_closest_static(o3, d3), other.trace_closest(o3, d3), "second scene")
        same(dev.trace_closest_static(o2, d2), dev.trace_closest(o2, d2), "first scene again")
        empty = dev.trace_closest_static(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
        assert len(empty["prim"]) == 0
        big_o, big_d = rays(cornell, abi.PT_TRACE_STATIC_MAX_RAYS + 1, 4)
        with pytest.raises(api.PtError) as e:
            dev.trace_closest_static(big_o, big_d)
        assert e.value.code == abi.PT_ERR_INVALID_ARG and "PT_TRACE_STATIC_MAX_RAYS" in str(e.value)
