"""The checker's front-to-back traversal mode (oracle/pt_oracle.c, bvh_intersect_ordered) against its restatement
of the reference's order, on the CPU: same primitive, t, u, v, attributes and occlusion for every ray, same
picture.  The front-to-back mode is no restatement of the reference; it is what the GPU library's
PT_TRAVERSAL_ORDERED counters are compared with, so it has to return the reference's hits itself.
"""
import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle import checker
from pathtracer_amd import abi

ORDERED = abi.PT_TRAVERSAL_ORDERED


def both_orders(desc, o, d, what):
    ref, ordered = checker.OracleScene(desc), checker.OracleScene(desc).set_traversal(ORDERED)
    a, b = ref.intersect(o, d), ordered.intersect(o, d)
    assert np.array_equal(a["prim"], b["prim"]), f"{what}: {(a['prim'] != b['prim']).sum()} primitives differ"
    for k in ("t", "u", "v", "position", "normal", "tangent", "bitangent", "uv"):
        assert_bits_equal(a[k], b[k], f"{what}: {k}")
    hit = a["prim"] >= 0
    tmax = np.where(hit, np.minimum(a["t"], np.float32(1e30)) * np.float32(1.25), np.float32(10)).astype(np.float32)
    tmax[::2] = a["t"][::2]
    oa, ob = ref.occluded(o, d, tmax), ordered.occluded(o, d, tmax)
    assert np.array_equal(oa["occluded"], ob["occluded"]), what
    print(f"{what}: nodes per ray {a['nodes'].mean():.1f} -> {b['nodes'].mean():.1f} (closest), "
          f"{oa['nodes'].mean():.1f} -> {ob['nodes'].mean():.1f} (any)")
    return a, b


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_golden_rays_and_a_random_batch(name, scenes, golden):
    scene, g = scenes[name], golden[name]
    desc = scene.to_desc()
    _, b = both_orders(desc, g["ray_o"], g["ray_d"], f"{name} golden rays")
    assert np.array_equal(b["prim"], g["hit_prim"])            # and with them the reference's own vectors
    assert_bits_equal(b["t"], g["hit_t"], "t vs the reference's vectors")
    rng = np.random.default_rng(2)
    n = 200_000
    lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
    o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 10, rng.integers(0, 3)] = 0.0
    both_orders(desc, o, d, f"{name} random rays")   # includes rays from outside that see coplanar faces from behind


def test_tie_breaking_and_rounding_cases():
    from geometry_cases import cases
    for what, (scene, o, d) in cases().items():
        both_orders(scene.to_desc(), o, d, what)


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_pictures_are_bit_identical(name, scenes):
    desc = scenes[name].with_image(40, 40).to_desc()
    a, sa = checker.OracleScene(desc).li_image(6, seed=3)
    b, sb = checker.OracleScene(desc).set_traversal(ORDERED).li_image(6, seed=3)
    assert_bits_equal(a, b, f"{name}: picture")
    for c in ("rays_closest", "rays_shadow", "shaded_hits", "draws", "samples"):
        assert sa[c] == sb[c], c
    assert sa["rays_redone"] == 0
    print(f"{name}: {sb['rays_redone']} of {sb['rays_closest']} rays traced twice, nodes {sa['nodes_closest']} -> {sb['nodes_closest']}")
