"""GPU parity: the HIP path, called through its C-ABI, against the CPU checker (oracle/) and the golden vectors
the reference produced (tests/golden).

Bars: hit primitive, t, u, v, occlusion bits and traversal counters are bit-exact.  Radiance in the
deterministic modes (Philox or a replayed sample table): bit-exact too (see radiance_close; the stated tolerance of
1e-4 * max(1, |cpu|) per pixel is kept as the outer bound).
"""
import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle import checker
from pathtracer_amd import abi, api, host

pytestmark = pytest.mark.gpu

RADIANCE_TOL = 1e-4


def radiance_close(gpu, cpu, what):
    """Every operation of the path is restated to the CPU's bit (sin / cos / acos / pow of the C library included); the
    one exception are the double-precision atan2 / sin / cos of a sphere hit's uv and tangent, whose device and host
    versions can round a float differently about once in a billion calls.  So: all pixels equal, at most one may
    differ, and that one within RADIANCE_TOL."""
    err = np.abs(gpu - cpu) / np.maximum(1.0, np.abs(cpu))
    same = (gpu.view(np.uint32) == cpu.view(np.uint32)).all(axis=-1)
    exact = same.mean()
    print(f"{what}: max err {err.max():.3e}, exactly equal pixels {100 * exact:.2f} %")
    assert np.isfinite(gpu).all(), f"{what}: non-finite radiance"
    assert err.max() <= RADIANCE_TOL, f"{what}: max relative error {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    assert (~same).sum() <= 1, f"{what}: {(~same).sum()} pixels differ from the CPU's"
    return exact


@pytest.fixture(scope="module")
def device_scenes(scenes):
    out = {}
    for name, s in scenes.items():
        small = s.with_image(48, 48)
        desc = small.to_desc()
        out[name] = (small, api.DeviceScene(desc), checker.OracleScene(desc))
    yield out
    for _, dev, _ in out.values():
        dev.close()


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_closest_hit_matches_reference_vectors(name, device_scenes, golden):
    _, dev, _ = device_scenes[name]
    g = golden[name]
    got = dev.trace_closest(g["ray_o"], g["ray_d"])
    assert np.array_equal(got["prim"], g["hit_prim"])
    hit = g["hit_prim"] >= 0
    assert_bits_equal(got["t"], g["hit_t"], "t")
    assert_bits_equal(got["u"][hit], g["hit_u"][hit], "u")
    assert_bits_equal(got["v"][hit], g["hit_v"][hit], "v")
    assert np.array_equal(got["nodes"], g["hit_nodes"])
    assert np.array_equal(got["prim_tests"], g["hit_prim_tests"])


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_any_hit_matches_reference_vectors(name, device_scenes, golden):
    _, dev, orc = device_scenes[name]
    g = golden[name]
    got = dev.trace_any(g["ray_o"], g["ray_d"], g["shadow_tmax"])
    assert np.array_equal(got["occluded"], g["shadow_occluded"])
    want = orc.occluded(g["ray_o"], g["ray_d"], g["shadow_tmax"])
    assert np.array_equal(got["nodes"], want["nodes"])
    assert np.array_equal(got["prim_tests"], want["prim_tests"])


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_closest_hit_matches_oracle_on_large_batch(name, device_scenes):
    scene, dev, orc = device_scenes[name]
    rng = np.random.default_rng(5)
    n = 300_000
    lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
    o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 10, rng.integers(0, 3)] = 0.0
    got, want = dev.trace_closest(o, d), orc.intersect(o, d)
    assert np.array_equal(got["prim"], want["prim"])
    assert_bits_equal(got["t"], want["t"], "t")
    assert_bits_equal(got["u"], want["u"], "u")
    assert_bits_equal(got["v"], want["v"], "v")
    assert np.array_equal(got["nodes"], want["nodes"])
    assert np.array_equal(got["prim_tests"], want["prim_tests"])
    tmax = rng.uniform(0, 3, n).astype(np.float32)
    got, want = dev.trace_any(o, d, tmax), orc.occluded(o, d, tmax)
    assert np.array_equal(got["occluded"], want["occluded"])


@pytest.mark.parametrize("name", ["cornell", "mixed"])
@pytest.mark.parametrize("spp", [1, 8])
def test_radiance_matches_reference_vectors(name, spp, device_scenes, golden):
    _, dev, _ = device_scenes[name]
    g = golden[name]
    dev.render(spp, seed=7)
    lin, _, stats = dev.readback()
    want = g[f"li_sum_spp{spp}"] / np.float32(spp)
    radiance_close(lin, want, f"{name} {spp} spp vs reference")
    assert stats.samples == 48 * 48 * spp
    assert stats.rays_closest == int(g[f"rays_spp{spp}"][0])
    # the reference also traces shadow rays whose contribution is zero; the GPU path skips those
    assert stats.rays_shadow <= int(g[f"rays_spp{spp}"][1])


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_radiance_sample_table_mode(name, device_scenes, golden):
    scene, dev, _ = device_scenes[name]
    g = golden[name]
    dev.render(2, table=g["table"], table_draws_per_sample=scene.draws_per_sample())
    lin, _, _ = dev.readback()
    radiance_close(lin, g["li_sum_table_spp2"] / np.float32(2), f"{name} table mode vs reference")


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_radiance_and_counters_match_oracle(name, device_scenes):
    _, dev, orc = device_scenes[name]
    dev.render(16, seed=123)
    lin, _, stats = dev.readback()
    want, ostats = orc.li_image(16, seed=123)
    radiance_close(lin, want / np.float32(16), f"{name} 16 spp vs oracle")
    assert stats.rays_closest == ostats["rays_closest"]
    assert stats.nodes_closest == ostats["nodes_closest"]
    assert stats.tri_tests_closest == ostats["tri_tests_closest"]
    assert stats.sphere_tests_closest == ostats["sphere_tests_closest"]
    assert stats.shaded_hits == ostats["shaded_hits"]


def test_split_ranges_and_passes_are_bit_identical(device_scenes):
    """Sample order is preserved across pt_render calls and across passes, so any split gives the same film."""
    _, dev, _ = device_scenes["cornell"]
    dev.render(8, seed=3)
    whole, _, _ = dev.readback()
    dev.render(8, 0, 3, seed=3)
    dev.render(8, 3, 8, seed=3, accumulate=True)
    split, _, _ = dev.readback()
    assert_bits_equal(whole, split, "split ranges")
    dev.render(8, seed=3, samples_per_pass=1)
    passes, _, _ = dev.readback()
    assert_bits_equal(whole, passes, "one sample per pass")
    dev.render(8, seed=3)
    again, _, _ = dev.readback()
    assert_bits_equal(whole, again, "repeat")
    for kw in ({"samples_per_pass": 3}, {"samples_per_pass": 1, "ordered": True}, {"samples_per_pass": 5, "ordered": True}):
        dev.render(8, seed=3, **kw)
        assert_bits_equal(whole, dev.readback()[0], str(kw))
        assert dev.stats().samples == 48 * 48 * 8


def test_tonemap_matches_reference_vectors(device_scenes, golden):
    """Film -> Uncharted2 / Reinhard -> gamma -> clamp against the checker's tonemap of the same linear image (the
    host C library's powf): bit for bit."""
    scene, dev, _ = device_scenes["cornell"]
    dev.render(4, seed=1)
    lin, tm, _ = dev.readback(abi.PT_TONEMAP_UNCHARTED2)
    want = checker.tonemap(lin, float(scene.camera["exposure"]), float(scene.camera["gamma"]), abi.PT_TONEMAP_UNCHARTED2)
    assert tm.tobytes() == want.tobytes()
    _, tm, _ = dev.readback(abi.PT_TONEMAP_REINHARD)
    want = checker.tonemap(lin, float(scene.camera["exposure"]), float(scene.camera["gamma"]), abi.PT_TONEMAP_REINHARD)
    assert tm.tobytes() == want.tobytes()
    _, tm, _ = dev.readback(abi.PT_TONEMAP_NONE)
    assert_bits_equal(tm, lin, "no tonemap")


@pytest.mark.parametrize("gamma,exposure", [(2.2, 1.0), (1.8, 0.7), (2.4, 3.0), (1.0, 1.0), (0.45, 2.0)])
def test_tonemap_is_bit_exact_on_a_film_of_arbitrary_radiance(gamma, exposure, scenes):
    """614,400 radiance values written into the film (pt_upload_film, one sample per pixel, so the mean is the
    value itself): dense in [0, 1.25], up to 1e4, zeros and subnormals.  Both operators must give the floats of
    the C library's powf path (the checker, and the host library's pth_tonemap), including the arguments where a
    correctly rounded pow differs from powf."""
    scene = scenes["cornell"].with_image(512, 400)
    scene.camera = dict(scene.camera, gamma=np.float32(gamma), exposure=np.float32(exposure))
    rng = np.random.default_rng(int(gamma * 100))
    n = 512 * 400 * 3
    values = rng.uniform(0, 1.25, n).astype(np.float32)
    values[: n // 4] = (rng.uniform(0, 1, n // 4) ** 4 * 1e4).astype(np.float32)
    values[n // 4: n // 4 + 1000] = np.float32(0)
    values[n // 4 + 1000: n // 4 + 2000] = (rng.integers(1, 1 << 23, 1000).astype(np.uint32)).view(np.float32)   # subnormals
    values[n // 4 + 2000: n // 4 + 3000] = rng.uniform(0, 1e-30, 1000).astype(np.float32)
    rng.shuffle(values)
    film = np.concatenate([values.reshape(400, 512, 3), np.ones((400, 512, 1), np.float32)], axis=2)
    with api.DeviceScene(scene.to_desc()) as dev:
        dev.upload_film(film)
        for op in (abi.PT_TONEMAP_UNCHARTED2, abi.PT_TONEMAP_REINHARD):
            lin, tm, _ = dev.readback(op)
            assert lin.tobytes() == values.tobytes()
            want = checker.tonemap(lin, float(exposure), float(gamma), op)
            differing = int((tm.view(np.uint32) != want.view(np.uint32)).sum())
            assert differing == 0, f"operator {op}: {differing} of {n} values differ from the C library's"
            assert tm.tobytes() == host.tonemap(lin, float(exposure), float(gamma), op).tobytes()


def test_depth_limits_and_empty_inputs(scenes):
    base = scenes["cornell"].with_image(16, 16)
    for depth in (0, 1, 2):
        s = base.copy()
        s.max_depth = depth
        desc = s.to_desc()
        with api.DeviceScene(desc) as dev:
            orc = checker.OracleScene(desc)
            dev.render(4, seed=9)
            lin, _, _ = dev.readback()
            want, _ = orc.li_image(4, seed=9)
            radiance_close(lin, want / np.float32(4), f"depth {depth}")
            empty = dev.trace_closest(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
            assert len(empty["prim"]) == 0
            dev.render(4, 2, 2, seed=9)  # empty sample range: film cleared, nothing rendered
            assert dev.stats().samples == 0


def test_invalid_scenes_are_rejected(scenes):
    s = scenes["cornell"].copy()
    s.nodes = s.nodes.copy()
    s.nodes["offset"][0] = 10_000  # second child far outside the array
    with pytest.raises(api.PtError) as e:
        api.DeviceScene(s.to_desc())
    assert e.value.code == abi.PT_ERR_INVALID_SCENE
    s = scenes["cornell"].copy()
    s.triangles = s.triangles.copy()
    s.triangles["i0"][0] = 1_000_000
    with pytest.raises(api.PtError):
        api.DeviceScene(s.to_desc())
    d = scenes["cornell"].to_desc()
    d.version = 99
    with pytest.raises(api.PtError) as e:
        api.DeviceScene(d)
    assert e.value.code == abi.PT_ERR_INVALID_ARG


def test_render_after_a_failed_growth_of_the_path_buffers(scenes):
    """A pass that cannot be allocated (31 samples of 8192 x 8192 pixels in flight: 374 GB of path state) fails with
    PT_ERR_DEVICE and leaves the scene usable: the next, smaller render allocates afresh and gives the checker's
    counters.  With a pass size of the library's own choosing the same request succeeds in smaller passes."""
    scene = scenes["cornell"].with_image(8192, 8192)
    with api.DeviceScene(scene.to_desc()) as dev:
        dev.render(1, seed=1, samples_per_pass=1)
        first = dev.stats()
        with pytest.raises(api.PtError) as e:
            dev.render(31, seed=1, samples_per_pass=31)
        assert e.value.code == abi.PT_ERR_DEVICE
        dev.render(1, seed=1, samples_per_pass=1)
        again = dev.stats()
        assert again.samples == 8192 * 8192 and again.rays_closest == first.rays_closest and again.nodes_closest == first.nodes_closest
        dev.render(3, seed=1)
        assert dev.stats().samples == 3 * 8192 * 8192


def test_film_tensor_and_single_rank_rccl_reduce(device_scenes):
    """The float4 film (sum r, g, b, sample count) is shared with torch without a copy and goes through one
    RCCL reduce; with one rank the reduce must leave it unchanged."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from pathtracer_amd import path_tracer

    _, dev, _ = device_scenes["cornell"]
    dev.render(4, seed=2)
    lin, _, _ = dev.readback(abi.PT_TONEMAP_NONE, tonemapped=False)
    film = path_tracer.film_tensor(dev, 0)
    assert film.is_cuda and film.numel() == 48 * 48 * 4
    host_film = film.cpu().numpy().reshape(48, 48, 4)
    assert (host_film[..., 3] == 4).all()
    assert_bits_equal(host_film[..., :3] / np.float32(4), lin, "film / count")

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        path_tracer.reduce_film(dev, 0, dist)
        after = path_tracer.film_tensor(dev, 0).cpu().numpy().reshape(48, 48, 4)
        assert_bits_equal(after, host_film, "film after a one-rank reduce")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("width,height", [(53, 37), (7, 5), (16, 40), (33, 16)])
def test_image_sizes_that_do_not_fill_whole_tiles(width, height, scenes):
    """Path ids follow 16 x 16 pixel tiles, then the columns right of the tiled area, then the rows below it:
    sizes that exercise each part (and an image smaller than one tile) must give the checker's image."""
    scene = scenes["mixed"].with_image(width, height)
    desc = scene.to_desc()
    with api.DeviceScene(desc) as dev:
        dev.render(3, seed=4)
        lin, _, stats = dev.readback()
        want, ostats = checker.OracleScene(desc).li_image(3, seed=4)
        radiance_close(lin, want / np.float32(3), f"{width}x{height}")
        assert stats.samples == width * height * 3 and stats.rays_closest == ostats["rays_closest"]


def test_film_checkpoint_and_resume_is_bit_identical(scenes):
    """pt_download_film / pt_upload_film: a render interrupted after 3 of 8 samples, carried through host memory
    into a freshly uploaded scene and continued, equals the uninterrupted one."""
    desc = scenes["cornell"].with_image(40, 24).to_desc()
    with api.DeviceScene(desc) as dev:
        dev.render(8, seed=6)
        whole, _, _ = dev.readback()
        dev.render(8, 0, 3, seed=6)
        checkpoint = dev.download_film()
        assert (checkpoint[..., 3] == 3).all()
    with api.DeviceScene(desc) as dev:
        dev.upload_film(checkpoint)
        dev.render(8, 3, 8, seed=6, accumulate=True)
        resumed, _, _ = dev.readback()
        assert (dev.download_film()[..., 3] == 8).all()
    assert_bits_equal(whole, resumed, "resumed render")


def collapse_small_subtrees(scene, max_shapes):
    """The same tree with every subtree of at most max_shapes primitives turned into one leaf (the node layout
    of src/bvh.hpp:10-21 allows numShapes > 1 and BVH::Intersect loops over them, src/bvh.cpp:247-250, although
    the reference's builders never emit such leaves)."""
    old = scene.nodes
    n = len(old)
    end = np.zeros(n, np.int64)      # first node after the subtree
    first = np.zeros(n, np.int64)    # first primitive of the subtree
    count = np.zeros(n, np.int64)    # primitives in the subtree
    for i in range(n - 1, -1, -1):
        if old["num_shapes"][i] > 0:
            end[i], first[i], count[i] = i + 1, old["offset"][i], old["num_shapes"][i]
        else:
            a, b = i + 1, int(old["offset"][i])
            end[i], first[i], count[i] = end[b], first[a], count[a] + count[b]
    keep, i = [], 0
    while i < n:
        keep.append(i)
        i = end[i] if count[i] <= max_shapes else i + 1
    new_index = {k: j for j, k in enumerate(keep)}
    nodes = old[keep].copy()
    for j, k in enumerate(keep):
        if count[k] <= max_shapes:
            nodes["offset"][j], nodes["num_shapes"][j] = first[k], count[k]
        else:
            nodes["offset"][j] = new_index[int(old["offset"][k])]
    out = scene.copy()
    out.nodes = nodes
    return out


@pytest.mark.parametrize("name,max_shapes", [("cornell", 2), ("cornell", 5), ("mixed", 3), ("mixed", 16)])
def test_leaves_with_several_shapes(name, max_shapes, scenes):
    scene = collapse_small_subtrees(scenes[name].with_image(40, 40), max_shapes)
    assert scene.nodes["num_shapes"].max() > 1 and scene.nodes["num_shapes"].max() <= max_shapes
    desc = scene.to_desc()
    with api.DeviceScene(desc) as dev:
        orc = checker.OracleScene(desc)
        rng = np.random.default_rng(3)
        n = 100_000
        lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
        o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        got, want = dev.trace_closest(o, d), orc.intersect(o, d)
        assert np.array_equal(got["prim"], want["prim"])
        for k in ("t", "u", "v"):
            assert_bits_equal(got[k], want[k], k)
        assert np.array_equal(got["nodes"], want["nodes"]) and np.array_equal(got["prim_tests"], want["prim_tests"])
        # the closest hit does not depend on how the primitives are grouped into leaves
        plain = checker.OracleScene(scenes[name].to_desc()).intersect(o, d)
        assert np.array_equal(plain["prim"], want["prim"])
        tmax = rng.uniform(0, 3, n).astype(np.float32)
        got, want = dev.trace_any(o, d, tmax), orc.occluded(o, d, tmax)
        assert np.array_equal(got["occluded"], want["occluded"]) and np.array_equal(got["nodes"], want["nodes"])
        dev.render(4, seed=6)
        lin, _, stats = dev.readback()
        ref, ostats = orc.li_image(4, seed=6)
        radiance_close(lin, ref / np.float32(4), f"{name}, leaves of up to {max_shapes}")
        assert stats.nodes_closest == ostats["nodes_closest"] and stats.tri_tests_closest == ostats["tri_tests_closest"]
        assert stats.sphere_tests_closest == ostats["sphere_tests_closest"]


@pytest.mark.parametrize("name,keep", [("mixed", 0), ("mixed", 1), ("cornell", 0)])
def test_single_light_direct_sum_equals_resolve_pass(name, keep, scenes):
    """With one light and one sample per hit the shadow-ray kernel adds the direct light itself; with
    PT_SCENE_NO_DIRECT_SUM set at upload the separate summation pass runs instead.  Same film, bit for bit, and
    both match the checker."""
    scene = scenes[name].with_image(40, 40)
    scene.lights = scene.lights[keep:keep + 1].copy()
    scene.lights["n_samples"] = 1
    want, _ = checker.OracleScene(scene.to_desc()).li_image(8, seed=4)
    films = []
    for separate_pass in (False, True):
        with api.DeviceScene(scene.to_desc(flags=abi.PT_SCENE_NO_DIRECT_SUM if separate_pass else 0)) as dev:
            dev.render(8, seed=4)
            lin, _, stats = dev.readback()
            assert stats.rays_shadow > 0
            assert (stats.launches[abi.PT_K_RESOLVE] > 0) == separate_pass
            films.append(lin)
    assert_bits_equal(films[0], films[1], "direct sum in the shadow-ray kernel vs separate pass")
    radiance_close(films[0], want / np.float32(8), f"{name}, light {keep} only")


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_rays_on_box_planes_with_zero_direction_components(name, device_scenes):
    """Origins exactly on node planes and directions with zero components: (plane - origin) * (1 / 0) is 0 * inf =
    NaN in the slab test, which the reference's comparisons let pass (src/intersection_tests.cpp:91-110).  Hit,
    t, u, v and the per-ray node and test counts must still equal the checker's."""
    scene, dev, orc = device_scenes[name]
    rng = np.random.default_rng(11)
    n = 60_000
    nodes = scene.nodes[rng.integers(0, len(scene.nodes), n)]
    corner = np.where(rng.integers(0, 2, (n, 3)) == 1, nodes["aabb_max"], nodes["aabb_min"]).astype(np.float32)
    lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
    o = rng.uniform(lo - 0.25, hi + 0.25, (n, 3)).astype(np.float32)
    on_plane = rng.integers(0, 2, (n, 3)) == 1          # some coordinates sit exactly on a plane of some node
    o = np.where(on_plane, corner, o).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    zero = rng.integers(0, 3, (n, 3)) == 0               # a third of the components are +0 or -0
    d = np.where(zero, np.where(rng.integers(0, 2, (n, 3)) == 1, np.float32(0.0), np.float32(-0.0)), d).astype(np.float32)
    d[(d == 0).all(axis=1)] = np.float32([0, 0, 1])
    got, want = dev.trace_closest(o, d), orc.intersect(o, d)
    assert np.array_equal(got["prim"], want["prim"])
    for k in ("t", "u", "v"):
        assert_bits_equal(got[k], want[k], k)
    assert np.array_equal(got["nodes"], want["nodes"]) and np.array_equal(got["prim_tests"], want["prim_tests"])
    tmax = rng.uniform(0, 3, n).astype(np.float32)
    got, want = dev.trace_any(o, d, tmax), orc.occluded(o, d, tmax)
    assert np.array_equal(got["occluded"], want["occluded"]) and np.array_equal(got["nodes"], want["nodes"])


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_converges_to_the_reference_renderer(name, scenes, golden):
    """32768 samples per pixel on the GPU (its own seed) against the mean of 8192 samples per pixel of the
    reference's own code (tests/golden/golden_converged.npz): every channel's image mean within 1 %, and the
    picture itself within a few percent pixel by pixel (what is left is the noise of the 8192-sample estimate)."""
    want = golden["converged"][f"{name}_mean"].astype(np.float64)
    size = want.shape[0]
    with api.DeviceScene(scenes[name].with_image(size, size).to_desc()) as dev:
        dev.render(32768, seed=4242)
        lin, _, stats = dev.readback()
    assert stats.samples == size * size * 32768
    got = lin.astype(np.float64)
    for c in range(3):
        a, b = got[..., c].mean(), want[..., c].mean()
        print(f"{name} channel {c}: gpu {a:.6f}, reference {b:.6f}, {100 * (a / b - 1):+.3f} %")
        assert abs(a - b) <= 0.01 * b
    rel = np.abs(got - want) / (want + 0.02)
    print(f"{name}: mean per-pixel deviation {100 * rel.mean():.3f} %, 99th percentile {100 * np.percentile(rel, 99):.3f} %")
    assert rel.mean() < 0.02 and np.percentile(rel, 99) < 0.10


def mutated_scene(base, seed):
    """A variation of a fixture scene with a random mix of features: several point / directional lights, area
    lights on triangles and on spheres with 1-3 samples each, emissive materials, a cube map or a constant
    background, depth 1-6.  Geometry and BVH stay the fixture's."""
    rng = np.random.default_rng(seed)
    s = base.with_image(36, 28, 36 / 28)
    s.max_depth = int(rng.integers(1, 7))
    s.materials = s.materials.copy()
    for m in range(len(s.materials)):
        if rng.uniform() < 0.4:
            s.materials["Ke"][m] = rng.uniform(0, 3, 3).astype(np.float32)
    lights = []

    def light(kind, **kw):
        rec = np.zeros(1, abi.LIGHT_DTYPE)
        rec["type"], rec["n_samples"], rec["prim"] = kind, 1, -1
        rec["Lemit"] = rng.uniform(0.5, 20, 3).astype(np.float32)
        for k, v in kw.items():
            rec[k] = v
        lights.append(rec)

    lo, hi = s.nodes["aabb_min"][0], s.nodes["aabb_max"][0]
    for _ in range(int(rng.integers(0, 3))):
        light(abi.PT_LIGHT_POINT, position=rng.uniform(lo, hi).astype(np.float32))
    for _ in range(int(rng.integers(0, 3))):
        d = rng.normal(size=3).astype(np.float32)
        light(abi.PT_LIGHT_DIRECTIONAL, direction=d / np.float32(np.linalg.norm(d)))
    for prim in rng.choice(len(s.prims), size=int(rng.integers(0, 4)), replace=False):
        light(abi.PT_LIGHT_AREA, prim=int(prim), n_samples=int(rng.integers(1, 4)))
    spheres = np.flatnonzero(s.prims["type"] == abi.PT_PRIM_SPHERE)
    if len(spheres) and rng.uniform() < 0.7:
        light(abi.PT_LIGHT_AREA, prim=int(rng.choice(spheres)), n_samples=int(rng.integers(1, 3)))
    rng.shuffle(lights)
    s.lights = np.concatenate(lights) if lights else np.zeros(0, abi.LIGHT_DTYPE)
    if rng.uniform() < 0.5:
        n = int(rng.integers(1, 9))
        s.skybox_width = s.skybox_height = n
        sky = rng.uniform(0, 1, (6, n, n, 4)).astype(np.float32)
        if n % 2:   # the bytes of an 8-bit picture
            s.skybox = (np.floor(sky * 255) / np.float32(255)).astype(np.float32).reshape(-1)
        else:       # the floats of a Radiance picture, far above 1
            s.skybox = (sky * np.exp(rng.uniform(-3, 5, (6, n, n, 1))).astype(np.float32)).reshape(-1)
    else:
        s.background_radiance = rng.uniform(0, 1, 3).astype(np.float32)
    return s


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_random_feature_mixes_match_the_checker(name, seed, scenes):
    scene = mutated_scene(scenes[name], 1000 * seed + len(name))
    desc = scene.to_desc()
    with api.DeviceScene(desc) as dev:
        dev.render(6, seed=seed)
        lin, _, stats = dev.readback()
    want, ostats = checker.OracleScene(desc).li_image(6, seed=seed)
    what = f"{name} variation {seed}: depth {scene.max_depth}, {len(scene.lights)} lights, {'cube map' if scene.skybox_width else 'background'}"
    radiance_close(lin, want / np.float32(6), what)
    for c in ("rays_closest", "nodes_closest", "tri_tests_closest", "sphere_tests_closest", "shaded_hits", "samples"):
        assert getattr(stats, c) == ostats[c], c
    assert stats.rays_shadow <= ostats["rays_shadow"]


def test_tie_breaking_and_rounding_cases():
    """Coincident triangles (equal t: the first in primitive order wins), shared edges and vertices, zero-area
    triangles, coordinates around 1e6 and 1e-4 (tests/geometry_cases.py; the checker is pinned against the
    reference's own code on the same cases): hit primitive, t, u, v, occlusion with tMax exactly at the hit, and
    the per-ray node and test counts."""
    from geometry_cases import cases
    for what, (scene, o, d) in cases().items():
        desc = scene.to_desc()
        with api.DeviceScene(desc) as dev:
            orc = checker.OracleScene(desc)
            got, want = dev.trace_closest(o, d), orc.intersect(o, d)
            hit = want["prim"] >= 0
            assert np.array_equal(got["prim"], want["prim"]), what
            for k in ("t", "u", "v"):
                assert_bits_equal(got[k], want[k], f"{what}: {k}")
            assert np.array_equal(got["nodes"], want["nodes"]) and np.array_equal(got["prim_tests"], want["prim_tests"]), what
            tmax = np.where(hit, np.minimum(want["t"], np.float32(1e30)) * np.float32(1.5), np.float32(10)).astype(np.float32)
            tmax[::2] = want["t"][::2]
            got, ref_ = dev.trace_any(o, d, tmax), orc.occluded(o, d, tmax)
            assert np.array_equal(got["occluded"], ref_["occluded"]) and np.array_equal(got["nodes"], ref_["nodes"]), what
            # and through the wave-scheduled kernels: a render of the same geometry under a constant sky
            dev.render(4, seed=3)
            lin, _, stats = dev.readback()
            img, ostats = orc.li_image(4, seed=3)
            radiance_close(lin, img / np.float32(4), what)
            assert stats.nodes_closest == ostats["nodes_closest"] and stats.tri_tests_closest == ostats["tri_tests_closest"]
            assert stats.shaded_hits == ostats["shaded_hits"] and stats.shaded_hits > 100, what


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_reference_host_objects_over_the_gpu_library(name, scenes, tmp_path):
    """The drop-in as a maintainer of the reference would build it: the reference's own PT::Scene, Camera,
    PathTracer and Image::Save (compiled from its sources in the checking library), with the sample loop of
    PathTracer::Render replaced by oracle/pt_render_binding.hpp (flatten -> pt_upload_scene -> pt_render ->
    pt_readback into renderedImage) and its tonemap pass kept.  The picture and the PNG file SaveImage writes
    from it equal, bit for bit and byte for byte, those of the reference's CPU Li over the same Philox samples
    followed by its tonemap and Image::Save.  With the tonemap done by the library instead, the picture is the
    Python binding's to the bit and within 1e-6 of the CPU one."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/libptref.so not built")
    rotation = np.array([0.1, -0.3, 0.0], np.float32)
    scene = scenes[name].with_image(40, 30, 40 / 30)
    view, up, right = host.camera_vectors(rotation)
    scene.camera = dict(scene.camera, view_dir=view, up_dir=up, right_dir=right)
    desc = scene.to_desc()
    rs = ref.RefScene(desc)
    saved = str(tmp_path / "reference_host.png")
    picture = rs.render_gpu(rotation, spp=5, seed=9, save_as=saved)

    sums, _ = rs.li_image(5, seed=9)                     # the reference's CPU Li, same Philox samples
    cpu_picture, _ = ref.tonemap((sums / np.float32(5)).reshape(-1, 3), float(scene.camera["exposure"]), float(scene.camera["gamma"]))
    cpu_picture = cpu_picture.reshape(30, 40, 3)
    assert picture.tobytes() == cpu_picture.tobytes()
    ref.save_image(str(tmp_path / "cpu.png"), cpu_picture)
    assert open(saved, "rb").read() == (tmp_path / "cpu.png").read_bytes()
    host.write_image(str(tmp_path / "host.png"), picture)
    assert open(saved, "rb").read() == (tmp_path / "host.png").read_bytes()
    assert picture.max() > 0.05

    on_gpu = rs.render_gpu(rotation, spp=5, seed=9, tonemap_on_gpu=True)
    with api.DeviceScene(desc) as dev:
        dev.render(5, seed=9)
        _, tonemapped, _ = dev.readback()
    assert on_gpu.tobytes() == tonemapped.tobytes()
    assert np.abs(on_gpu - cpu_picture).max() <= 1e-6
