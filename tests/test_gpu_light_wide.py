"""The wide nodes in the frame of a directional light (the records ConnectLightWideKernel reads): what a node keeps of a
child must bound the child's own extent - its rectangle on the light's grid and its reach along the rays' main axis - all
the way down, because a shadow ray only descends where these tests pass.  A record keeps rounded figures, so the test
takes of a child's extent what its own record pins down: its first cells exactly, its last cells and its reach from below.  Checked on the records
as the upload's kernels wrote them, read back through pt_read_layout; that films over them equal the reference's is
tests/test_gpu_parallel_shadows.py."""
import numpy as np
import pytest

from pathtracer_amd import abi, api, host

pytestmark = pytest.mark.gpu


def decode(raw):
    """Per node: origin (2), shifts (2), masks, bases, d0, 1 / step, and per slot reach, first u, last u, first v, last v"""
    w = raw.view(np.uint32).reshape(-1, 16)
    n = {"ou": w[:, 0] & 0xFFFF, "ov": w[:, 0] >> 16, "su": w[:, 1] & 0xFF, "sv": (w[:, 1] >> 8) & 0xFF,
         "imask": (w[:, 1] >> 16) & 0xFF, "lmask": w[:, 1] >> 24, "child": w[:, 2], "prim": w[:, 3],
         "d0": w[:, 4].view(np.float32).astype(np.float64), "inv_step": w[:, 5].view(np.float32).astype(np.float64)}
    by = raw.reshape(-1, 64)
    for name, at in (("reach", 24), ("u0", 32), ("u1", 40), ("v0", 48), ("v1", 56)):
        n[name] = by[:, at:at + 8].astype(np.int64)
    return n


def check_nesting(raw, wide_raw, what):
    n = decode(raw)
    count = len(n["ou"])
    wide = wide_raw.view(np.uint32).reshape(-1, 20)
    assert count == len(wide), what
    # same children as the wide layout
    assert (n["imask"] == wide[:, 3] >> 24).all() and (n["lmask"] == (wide[:, 6] & 0xFF)).all(), what
    assert (n["child"] == wide[:, 4]).all() and (n["prim"] == wide[:, 5]).all(), what
    step = 1.0 / n["inv_step"]
    # a node's own extent: the union of its slots (cells, inclusive), and its largest reach
    used = ((n["imask"] | n["lmask"])[:, None] >> np.arange(8)) & 1
    assert (used.sum(axis=1) > 0).all(), what
    empty = used == 0
    assert (n["u0"][empty] == 255).all() and (n["u1"][empty] == 0).all() and (n["v0"][empty] == 255).all() and (n["v1"][empty] == 0).all(), what
    big = np.int64(1) << 40
    su, sv = n["su"][:, None].astype(np.int64), n["sv"][:, None].astype(np.int64)
    ou, ov = n["ou"][:, None].astype(np.int64), n["ov"][:, None].astype(np.int64)
    # the cells a slot lets through: [lo, hi]; its last cell proper lies in [last_least, hi] (the shift rounds down)
    lo_u, hi_u, last_least_u = ou + (n["u0"] << su), ou + ((n["u1"] + 1) << su) - 1, ou + (n["u1"] << su)
    lo_v, hi_v, last_least_v = ov + (n["v0"] << sv), ov + ((n["v1"] + 1) << sv) - 1, ov + (n["v1"] << sv)
    far = n["d0"][:, None] + n["reach"] * step[:, None]                                   # how far a slot is taken to reach
    far_least = n["d0"][:, None] + np.maximum(n["reach"] - 1, 0) * step[:, None]          # its reach proper is beyond this
    # what is known exactly of a node's own extent: its first cells (the record's origin) - and its last cells and its
    # reach are at least the largest of the slots' lower figures
    assert (np.where(used, lo_u, big).min(axis=1) == n["ou"]).all() and (np.where(used, lo_v, big).min(axis=1) == n["ov"]).all(), what
    node_last_u = np.where(used, last_least_u, -big).max(axis=1)
    node_last_v = np.where(used, last_least_v, -big).max(axis=1)
    node_far = np.where(used, far_least, -np.inf).max(axis=1)
    # every inner child, as its parent keeps it, against what the child keeps of its own children
    inner = ((n["imask"][:, None] >> np.arange(8)) & 1).astype(bool)
    rank = np.cumsum(inner, axis=1) - 1
    parents, slots = np.nonzero(inner)
    children = n["child"][parents].astype(np.int64) + rank[parents, slots]
    assert children.max() < count and len(np.unique(children)) == len(children) == count - 1, what
    assert (lo_u[parents, slots] <= n["ou"][children]).all() and (node_last_u[children] <= hi_u[parents, slots]).all(), what
    assert (lo_v[parents, slots] <= n["ov"][children]).all() and (node_last_v[children] <= hi_v[parents, slots]).all(), what
    assert (far[parents, slots] >= node_far[children]).all() and (far[parents, slots] >= n["d0"][children]).all(), what
    assert (n["su"] <= 8).all() and (n["sv"] <= 8).all(), what
    return count


def records(desc):
    with api.DeviceScene(desc) as dev:
        return dev.read_layout(abi.PT_LAYOUT_LIGHT_WIDE_NODES), dev.read_layout(abi.PT_LAYOUT_WIDE_NODES)


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_golden_scenes_under_every_light(name, scenes):
    from test_gpu_parallel_shadows import DIRECTIONS, one_directional_light
    for label, direction in DIRECTIONS.items():
        raw, wide = records(one_directional_light(scenes[name], direction).to_desc())
        assert len(raw) > 0
        check_nesting(raw, wide, f"{name}, light {label}")


def test_atrium():
    with host.HostScene.atrium(seed=1, triangles=262267, texture_size=64) as hs:
        raw, wide = records(hs.desc())
        assert check_nesting(raw, wide, "atrium") > 50000


def test_scenes_without_the_layout(scenes):
    """Two lights, a point light, or the switch that leaves the light's frame out: no records, and the shadow rays of a
    render over the wide layout take the wide nodes themselves"""
    from test_gpu_parallel_shadows import DIRECTIONS, one_directional_light
    s = one_directional_light(scenes["cornell"], DIRECTIONS["oblique"])
    raw, _ = records(s.to_desc(flags=abi.PT_SCENE_NO_PARALLEL_SHADOWS))
    assert len(raw) == 0
    raw, _ = records(s.to_desc(flags=abi.PT_SCENE_HOST_LAYOUTS))
    assert len(raw) == 0
    raw, _ = records(scenes["cornell"].to_desc())
    assert len(raw) == 0
