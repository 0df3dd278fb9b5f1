"""GPU parity of the traversal over the eight-wide node layout (PT_RENDER_WIDE / PT_TRAVERSAL_WIDE).

What has to hold: the hit primitive, t, u, v, the occlusion bits and the film are those of the reference's order,
bit for bit - against the reference's own vectors (tests/golden), the checker's restatement of the reference's order
and the library's reference-order mode.  The node counters of this mode count wide nodes and have no counterpart in
the reference or the checker; the ray and shaded-hit counters must equal the reference-order mode's.  Every test
prints the number of mismatches it found (all must be zero) and the visits per ray of both structures.
"""
import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle import checker
from pathtracer_amd import abi, api, host

pytestmark = pytest.mark.gpu

WIDE = abi.PT_TRAVERSAL_WIDE


def mismatches(got, want):
    bad = got["prim"] != want["prim"]
    for k in ("t", "u", "v"):
        bad |= got[k].view(np.uint32) != want[k].view(np.uint32)
    return int(bad.sum())


def check_batch(dev, desc, o, d, what, orc=None):
    """Closest and any hit of a ray batch over the wide layout against the reference order (checker)."""
    orc = orc or checker.OracleScene(desc)
    want = orc.intersect(o, d)
    got = dev.trace_closest(o, d, WIDE)
    bad = mismatches(got, want)
    print(f"{what}: {len(got['prim'])} rays, {bad} mismatches against the reference's order; visits per ray "
          f"{want['nodes'].mean():.1f} binary -> {got['nodes'].mean():.1f} wide (+ second passes), tests per ray "
          f"{want['prim_tests'].mean():.2f} -> {got['prim_tests'].mean():.2f}")
    assert bad == 0, what
    hit = want["prim"] >= 0
    tmax = np.where(hit, np.minimum(want["t"], np.float32(1e30)) * np.float32(1.25), np.float32(10)).astype(np.float32)
    tmax[::2] = want["t"][::2]
    got, ref_ = dev.trace_any(o, d, tmax, WIDE), orc.occluded(o, d, tmax)
    wrong = int((got["occluded"] != ref_["occluded"]).sum())
    print(f"{what}: {wrong} occlusion answers differ; visits per shadow ray {ref_['nodes'].mean():.1f} binary -> {got['nodes'].mean():.1f} wide")
    assert wrong == 0, what
    return want


def check_render(dev, desc, spp, seed, what):
    """The film of the wide mode equals the film of the reference-order mode bit for bit."""
    dev.render(spp, seed=seed)
    ref_film, _, ref_stats = dev.readback(abi.PT_TONEMAP_NONE, tonemapped=False)
    dev.render(spp, seed=seed, wide=True)
    film, _, stats = dev.readback(abi.PT_TONEMAP_NONE, tonemapped=False)
    differing = int((film.view(np.uint32) != ref_film.view(np.uint32)).any(axis=-1).sum())
    print(f"{what}: {differing} pixels differ between the reference order and the wide layout; visits per closest-hit ray "
          f"{ref_stats.nodes_closest / max(1, ref_stats.rays_closest):.1f} binary -> {stats.nodes_closest / max(1, stats.rays_closest):.1f}, "
          f"primitive tests {ref_stats.tri_tests_closest / max(1, ref_stats.rays_closest):.2f} -> {stats.tri_tests_closest / max(1, stats.rays_closest):.2f}, "
          f"{stats.rays_redone} of {stats.rays_closest} rays traced twice")
    assert differing == 0, what
    for c in ("samples", "rays_closest", "rays_shadow", "shaded_hits"):
        assert getattr(stats, c) == getattr(ref_stats, c), c
    assert ref_stats.rays_redone == 0
    dev.render(spp, seed=seed, wide=True, wide_shadows=True)
    film2, _, stats2 = dev.readback(abi.PT_TONEMAP_NONE, tonemapped=False)
    differing = int((film2.view(np.uint32) != ref_film.view(np.uint32)).any(axis=-1).sum())
    print(f"{what}: {differing} pixels differ with the shadow rays over the wide layout too; visits per shadow ray "
          f"{ref_stats.nodes_shadow / max(1, ref_stats.rays_shadow):.1f} binary -> {stats2.nodes_shadow / max(1, stats2.rays_shadow):.1f}, "
          f"primitive tests {ref_stats.tri_tests_shadow / max(1, ref_stats.rays_shadow):.2f} -> {stats2.tri_tests_shadow / max(1, stats2.rays_shadow):.2f}")
    assert differing == 0, what
    for c in ("samples", "rays_closest", "rays_shadow", "shaded_hits"):
        assert getattr(stats2, c) == getattr(ref_stats, c), c
    return film, stats


@pytest.fixture(scope="module")
def device_scenes(scenes):
    out = {}
    for name, s in scenes.items():
        small = s.with_image(48, 48)
        desc = small.to_desc()
        out[name] = (small, api.DeviceScene(desc), desc)
    yield out
    for _, dev, _ in out.values():
        dev.close()


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_reference_vectors(name, device_scenes, golden):
    _, dev, desc = device_scenes[name]
    g = golden[name]
    got = dev.trace_closest(g["ray_o"], g["ray_d"], WIDE)
    assert np.array_equal(got["prim"], g["hit_prim"])
    hit = g["hit_prim"] >= 0
    assert_bits_equal(got["t"], g["hit_t"], "t")
    assert_bits_equal(got["u"][hit], g["hit_u"][hit], "u")
    assert_bits_equal(got["v"][hit], g["hit_v"][hit], "v")
    got = dev.trace_any(g["ray_o"], g["ray_d"], g["shadow_tmax"], WIDE)
    assert np.array_equal(got["occluded"], g["shadow_occluded"])
    check_batch(dev, desc, g["ray_o"], g["ray_d"], f"{name}: the reference's rays")
    for spp in (1, 8):
        dev.render(spp, seed=7, wide=True)
        lin, _, stats = dev.readback()
        dev.render(spp, seed=7)
        assert_bits_equal(lin, dev.readback()[0], f"{name} {spp} spp: wide vs reference order")
        assert stats.rays_closest == int(g[f"rays_spp{spp}"][0])


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_large_batches(name, device_scenes):
    scene, dev, desc = device_scenes[name]
    rng = np.random.default_rng(5)
    n = 300_000
    lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
    o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)   # origins outside see coplanar faces from behind
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 10, rng.integers(0, 3)] = 0.0
    check_batch(dev, desc, o, d, f"{name}: random rays")
    # origins exactly on node planes, directions with zero components: NaN slab values
    nodes = scene.nodes[rng.integers(0, len(scene.nodes), n)]
    corner = np.where(rng.integers(0, 2, (n, 3)) == 1, nodes["aabb_max"], nodes["aabb_min"]).astype(np.float32)
    o = np.where(rng.integers(0, 2, (n, 3)) == 1, corner, o).astype(np.float32)
    zero = rng.integers(0, 3, (n, 3)) == 0
    d = np.where(zero, np.where(rng.integers(0, 2, (n, 3)) == 1, np.float32(0.0), np.float32(-0.0)), d).astype(np.float32)
    d[(d == 0).all(axis=1)] = np.float32([0, 0, 1])
    check_batch(dev, desc, o, d, f"{name}: rays on box planes with zero direction components")


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_extreme_directions_and_origins(name, device_scenes):
    """Direction components from 1e-38 to 1e30 (reciprocals beyond the clamp of the wide box test, denormal
    components whose reciprocal is infinite), origins far outside the scene."""
    scene, dev, desc = device_scenes[name]
    rng = np.random.default_rng(9)
    n = 120_000
    lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
    o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    scale = np.float32(10.0) ** rng.integers(-38, 31, (n, 3)).astype(np.float32)
    d = (d * np.where(rng.integers(0, 3, (n, 3)) == 0, scale, np.float32(1.0))).astype(np.float32)
    d[(d == 0).all(axis=1)] = np.float32([0, 0, 1])
    check_batch(dev, desc, o, d, f"{name}: tiny and huge direction components")
    o = (o + rng.normal(size=(n, 3)).astype(np.float32) * np.float32(1e4)).astype(np.float32)
    centre = ((lo + hi) / 2).astype(np.float32)
    d = (centre - o + rng.normal(size=(n, 3)).astype(np.float32) * (hi - lo).astype(np.float32)).astype(np.float32)
    check_batch(dev, desc, o, d, f"{name}: origins 10^4 units away")


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_films(name, device_scenes):
    _, dev, desc = device_scenes[name]
    check_render(dev, desc, 16, 123, f"{name} 48x48 16 spp")
    dev.render(8, seed=3, wide=True)
    whole = dev.readback()[0]
    dev.render(8, 0, 3, seed=3, wide=True)
    dev.render(8, 3, 6, seed=3, accumulate=True)            # the modes may be mixed freely: same film
    dev.render(8, 6, 8, seed=3, accumulate=True, ordered=True)
    assert_bits_equal(whole, dev.readback()[0], "split ranges in three modes")
    dev.render(8, seed=3, samples_per_pass=1, wide=True)
    assert_bits_equal(whole, dev.readback()[0], "one sample per pass")


def test_tie_breaking_and_rounding_cases():
    from geometry_cases import cases
    for what, (scene, o, d) in cases().items():
        desc = scene.to_desc()
        with api.DeviceScene(desc) as dev:
            check_batch(dev, desc, o, d, what)
            check_render(dev, desc, 4, 3, what)


@pytest.mark.parametrize("name,max_shapes", [("cornell", 5), ("mixed", 16)])
def test_leaves_with_several_shapes(name, max_shapes, scenes):
    from test_gpu_parity import collapse_small_subtrees
    scene = collapse_small_subtrees(scenes[name].with_image(40, 40), max_shapes)
    desc = scene.to_desc()
    rng = np.random.default_rng(3)
    n = 100_000
    lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
    o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    with api.DeviceScene(desc) as dev:
        check_batch(dev, desc, o, d, f"{name}, leaves of up to {max_shapes}")
        check_render(dev, desc, 4, 6, f"{name}, leaves of up to {max_shapes}")


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_random_feature_mixes(name, seed, scenes):
    from test_gpu_parity import mutated_scene
    scene = mutated_scene(scenes[name], 1000 * seed + len(name))
    desc = scene.to_desc()
    with api.DeviceScene(desc) as dev:
        check_render(dev, desc, 6, seed, f"{name} variation {seed}")


def test_scene_without_the_wide_layout(scenes):
    desc = scenes["cornell"].with_image(16, 16).to_desc(flags=abi.PT_SCENE_NO_WIDE_LAYOUT)
    with api.DeviceScene(desc) as dev:
        dev.render(2, seed=1, ordered=True)
        with pytest.raises(api.PtError) as e:
            dev.render(2, seed=1, wide=True)
        assert e.value.code == abi.PT_ERR_STATE
        o, d = np.zeros((4, 3), np.float32), np.ones((4, 3), np.float32)
        with pytest.raises(api.PtError) as e:
            dev.trace_closest(o, d, WIDE)
        assert e.value.code == abi.PT_ERR_STATE
        with pytest.raises(api.PtError) as e:
            dev.trace_any(o, d, np.ones(4, np.float32), WIDE)
        assert e.value.code == abi.PT_ERR_STATE


def test_single_primitive_scene():
    """A tree that is one leaf: the wide root holds the one primitive."""
    from geometry_cases import triangle_scene
    scene = triangle_scene(np.float32([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]))
    assert len(scene.nodes) == 1
    rng = np.random.default_rng(1)
    n = 20_000
    o = (rng.uniform(-1, 2, (n, 3)) * np.float32([1, 1, 0]) + np.float32([0, 0, 2])).astype(np.float32)
    d = (rng.normal(size=(n, 3)) * np.float32([0.3, 0.3, 0.1]) + np.float32([0, 0, -1])).astype(np.float32)
    desc = scene.to_desc()
    with api.DeviceScene(desc) as dev:
        check_batch(dev, desc, o, d, "one primitive")
        check_render(dev, desc, 4, 3, "one primitive")


# ------------------------------------------------------------------------------------------------ configurations

@pytest.fixture(scope="module")
def atrium():
    hs = host.HostScene.atrium(seed=1, triangles=262267, texture_size=1024)
    hs.set_image(1920, 1080, 1920 / 1080)
    desc = hs.desc()
    dev = api.DeviceScene(desc)
    yield hs, dev, desc
    dev.close()
    hs.close()


def test_atrium_million_ray_batch(atrium):
    _, dev, desc = atrium
    rng = np.random.default_rng(17)
    n = 1_000_000
    o = np.tile(np.array([[-9.0, 2.0, 0.0]], np.float32), (n, 1)) + rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    check_batch(dev, desc, o, d, "atrium: a million rays")


def test_atrium_1080p_film(atrium):
    _, dev, desc = atrium
    check_render(dev, desc, 2, 11, "atrium 1080p 2 spp")
    check_render(dev, desc, 24, 5, "atrium 1080p 24 spp")


def test_cornell_1024(scenes):
    desc = scenes["cornell"].with_image(1024, 1024).to_desc()
    with api.DeviceScene(desc) as dev:
        check_render(dev, desc, 4, 0, "cornell 1024^2 4 spp")
        check_render(dev, desc, 64, 1, "cornell 1024^2 64 spp")


def test_knot_4k_primary_and_shadow_only():
    with host.HostScene.knot(seed=2, triangles=871414) as hs:
        hs.set_image(3840, 2160, 3840 / 2160)
        hs.set_max_depth(1)
        desc = hs.desc()
        with api.DeviceScene(desc) as dev:
            check_render(dev, desc, 1, 4, "knot 4K depth 1")


def test_soup_2m_with_environment_map():
    with host.HostScene.soup(seed=3, triangles=2_000_000, edge=0.01, skybox_size=256) as hs:
        hs.set_image(512, 512, 1.0)
        desc = hs.desc()
        with api.DeviceScene(desc) as dev:
            check_render(dev, desc, 4, 8, "soup 2M 512^2 4 spp")


@pytest.mark.timeout(900)
def test_soup_20m_triangles_4k():
    with host.HostScene.soup(seed=3, triangles=20_000_000, edge=0.01, skybox_size=1024) as hs:
        desc = hs.desc()
        with api.DeviceScene(desc) as dev:
            check_render(dev, desc, 1, 21, "soup 20M 4K 1 spp")
