"""The CPU checker (oracle/pt_oracle.c) against the golden vectors the reference itself produced
(tests/golden/make_golden.py).  Everything is compared bit for bit; runs anywhere, no GPU, no reference."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle import checker


def test_slab_test_vectors(golden):
    g = golden["leaf"]
    got = checker.ray_aabb(g["aabb_o"], g["aabb_inv"], g["aabb_min"], g["aabb_max"], g["aabb_maxt"])
    assert np.array_equal(got, g["aabb_hit"])
    assert 0.05 < g["aabb_hit"].mean() < 0.95  # both outcomes are exercised


def test_triangle_test_vectors(golden):
    g = golden["leaf"]
    hit, t, u, v = checker.ray_triangle(g["tri_o"], g["tri_d"], g["tri_v0"], g["tri_v1"], g["tri_v2"], g["tri_maxt"])
    assert np.array_equal(hit, g["tri_hit"])
    # t / u / v are outputs even of rejected tests up to the point of rejection; compare where the reference hit
    m = g["tri_hit"] == 1
    assert m.sum() > 500
    assert_bits_equal(t[m], g["tri_t"][m], "t")
    assert_bits_equal(u[m], g["tri_u"][m], "u")
    assert_bits_equal(v[m], g["tri_v"][m], "v")


def test_sphere_test_vectors(golden):
    g = golden["leaf"]
    hit, t = checker.ray_sphere(g["sph_o"], g["sph_d"], g["sph_c"], g["sph_r"], g["sph_maxt"])
    assert np.array_equal(hit, g["sph_hit"])
    m = g["sph_hit"] == 1
    assert m.sum() > 200
    assert_bits_equal(t[m], g["sph_t"][m], "t")


def test_sampling_warp_vectors(golden):
    g = golden["leaf"]
    got = checker.sampling(g["smp_u1"], g["smp_u2"])
    for key in ("cosine_hemisphere", "concentric_disk", "uniform_sphere", "uniform_triangle"):
        assert_bits_equal(got[key], g["smp_" + key], key)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_tonemap_vectors(golden, i):
    g = golden["leaf"]
    exposure, gamma = [float(x) for x in g[f"tm_params{i}"]]
    assert_bits_equal(checker.tonemap(g["tm_rgb"], exposure, gamma, 0), g[f"tm_uncharted2_{i}"], "uncharted2")
    assert_bits_equal(checker.tonemap(g["tm_rgb"], exposure, gamma, 1), g[f"tm_reinhard_{i}"], "reinhard")
    assert_bits_equal(checker.tonemap(g["tm_rgb"], exposure, gamma, 2), g["tm_rgb"], "none")


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_closest_hit_vectors(name, scenes, golden):
    g = golden[name]
    orc = checker.OracleScene(scenes[name].to_desc())
    got = orc.intersect(g["ray_o"], g["ray_d"])
    assert np.array_equal(got["prim"], g["hit_prim"])
    hit = g["hit_prim"] >= 0
    assert 0.2 < hit.mean() < 0.95
    assert_bits_equal(got["t"], g["hit_t"], "t")
    for key in ("u", "v", "position", "normal", "tangent", "bitangent", "uv"):
        assert_bits_equal(got[key][hit], g["hit_" + key][hit], key)
    assert np.array_equal(got["material"], g["hit_material"])
    assert np.array_equal(got["nodes"], g["hit_nodes"])
    assert np.array_equal(got["prim_tests"], g["hit_prim_tests"])


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_any_hit_vectors(name, scenes, golden):
    g = golden[name]
    orc = checker.OracleScene(scenes[name].to_desc())
    assert np.array_equal(orc.occluded(g["ray_o"], g["ray_d"], g["shadow_tmax"])["occluded"], g["shadow_occluded"])


def test_texture_fetch_vectors(scenes, golden):
    g = golden["mixed"]
    orc = checker.OracleScene(scenes["mixed"].to_desc())
    assert_bits_equal(orc.texture_fetch(0, g["tex_u"], g["tex_v"]), g["tex_rgba"], "texel")


@pytest.mark.parametrize("name", ["cornell", "mixed"])
@pytest.mark.parametrize("spp", [1, 8])
def test_radiance_vectors(name, spp, scenes, golden):
    g = golden[name]
    orc = checker.OracleScene(scenes[name].with_image(48, 48).to_desc())
    got, stats = orc.li_image(spp, seed=7)
    assert_bits_equal(got, g[f"li_sum_spp{spp}"], "radiance sum")
    assert stats["rays_closest"] == int(g[f"rays_spp{spp}"][0])
    assert stats["rays_shadow"] == int(g[f"rays_spp{spp}"][1])
    assert stats["samples"] == 48 * 48 * spp


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_radiance_sample_table_vectors(name, scenes, golden):
    g = golden[name]
    small = scenes[name].with_image(48, 48)
    orc = checker.OracleScene(small.to_desc())
    got, _ = orc.li_image(2, table=g["table"], draws_per_sample=small.draws_per_sample())
    assert_bits_equal(got, g["li_sum_table_spp2"], "radiance sum, table mode")


def test_draws_per_sample_layout(scenes):
    """Two draws for the jitter, then a constant number per bounce: two per area-light sample and two for the
    BRDF sample (src/anti_aliasing.cpp:91, src/shapes.cpp:45,107, src/resource/material.cpp:22)."""
    cornell = scenes["cornell"].with_image(8, 8)
    assert cornell.draws_per_bounce() == 6 and cornell.draws_per_sample() == 32
    _, stats = checker.OracleScene(cornell.to_desc()).li_image(4, seed=1)
    assert stats["draws"] <= 64 * 4 * 32
    assert (stats["draws"] - 2 * stats["samples"]) % 6 == 0
    mixed = scenes["mixed"]
    assert mixed.draws_per_bounce() == 2


def test_sample_ranges_compose(scenes):
    """Sums over [0, a) and [a, n) add up (in order) to the sum over [0, n): the sample-split used across GPUs."""
    orc = checker.OracleScene(scenes["cornell"].with_image(16, 16).to_desc())
    whole, _ = orc.li_image(8, 0, 8, seed=5)
    a, _ = orc.li_image(8, 0, 3, seed=5)
    b, _ = orc.li_image(8, 3, 8, seed=5)
    assert np.allclose(a + b, whole, rtol=2e-6, atol=1e-7)
    one, _ = orc.li_image(8, 0, 8, seed=5, threads=1)
    assert_bits_equal(one, whole, "thread count must not matter")


def test_skybox_lookup_faces():
    """Cubemap lookup (restated from src/resource/skybox.cpp:84-118; no reference build covers it): the six
    axis directions select faces right, left, top, bottom, back, front, and ties go to z, then y."""
    from pathtracer_amd.scene import SceneData
    from pathtracer_amd import abi
    s = SceneData()
    s.nodes = np.zeros(1, abi.NODE_DTYPE)
    s.nodes["aabb_min"][0], s.nodes["aabb_max"][0] = [10, 10, 10], [11, 11, 11]
    s.nodes["num_shapes"][0] = 1
    s.prims = np.zeros(1, abi.PRIM_DTYPE)
    s.triangles = np.zeros(1, abi.TRIANGLE_DTYPE)
    s.triangles[0] = (0, 1, 2, 0)
    s.meshes = np.zeros(1, abi.MESH_DTYPE)
    s.meshes[0] = (0, 3, 0, 0)
    s.vertices = np.array([[10, 10, 10], [11, 10, 10], [10, 11, 10]], np.float32)
    s.normals = np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    s.tangents = np.tile(np.array([1, 0, 0], np.float32), (3, 1))
    s.uvs = np.zeros((3, 2), np.float32)
    s.materials = np.zeros(1, abi.MATERIAL_DTYPE)
    s.materials["albedo_texture"] = -1
    w = h = 4
    sky = np.zeros((6, h, w, 4), np.float32)
    for f in range(6):
        sky[f, :, :, 0] = f
        sky[f, :, :, 1] = np.arange(h)[:, None]
        sky[f, :, :, 2] = np.arange(w)[None, :]
    s.skybox, s.skybox_width, s.skybox_height = sky.reshape(-1), w, h
    orc = checker.OracleScene(s.to_desc())
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
                     [1, 1, 1], [1, 1, 0.5], [2, 0.5, -0.25]], np.float32)
    px = orc.skybox_fetch(dirs)
    assert px[:6, 0].tolist() == [0, 1, 2, 3, 4, 5]
    assert px[6, 0] == 4 and px[7, 0] == 2          # tie order: z first, then y
    # (2, 0.5, -0.25): +x face, u = -z * 0.25 + 0.5 = 0.5625 -> column 2, v = y * 0.25 + 0.5 = 0.625 -> row 2
    assert px[8].tolist()[:3] == [0, 2, 2]


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_converges_to_the_reference_renderer(name, scenes, golden):
    """1024 samples per pixel of the checker (its own seed) against the mean of 8192 samples of the reference's
    own code: every channel's image mean within 1 %."""
    want = golden["converged"][f"{name}_mean"]
    size = want.shape[0]
    img, _ = checker.OracleScene(scenes[name].with_image(size, size).to_desc()).li_image(1024, seed=77)
    got = img / np.float32(1024)
    for c in range(3):
        a, b = float(got[..., c].mean(dtype=np.float64)), float(want[..., c].mean(dtype=np.float64))
        assert abs(a - b) <= 0.01 * b, (c, a, b)
