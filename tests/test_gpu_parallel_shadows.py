"""Shadow rays of a scene lit by one directional light, traced in the light's frame (PT_RENDER_ORDERED with the third
node layout, PT_RENDER_WIDE_SHADOWS with the wide nodes in that frame): the film must equal the film of the reference's own traversal bit for bit - every occlusion answer is
the reference's - for lights along an axis, oblique and almost along an axis, for leaves of several shapes, spheres,
geometry far from the origin, and at the size of the benchmark's scene."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from pathtracer_amd import abi, api

pytestmark = pytest.mark.gpu

DIRECTIONS = {
    "down": (0, -1, 0), "up": (0, 1, 0), "along +x": (1, 0, 0), "along -z": (0, 0, -1), "unnormalised axis": (0, -6, 0),
    "oblique": (0.3, -0.8, 0.5), "oblique 2": (-0.7, -0.2, -0.68), "almost down": (1e-4, -1, 0), "diagonal": (1, 1, 1),
    "denormal component": (1e-40, -1, 0),
}


def one_directional_light(base, direction, size=40, depth=4):
    s = base.with_image(size, size, 1.0)
    s.max_depth = depth
    rec = np.zeros(1, abi.LIGHT_DTYPE)
    rec["type"], rec["n_samples"], rec["prim"] = abi.PT_LIGHT_DIRECTIONAL, 1, -1
    rec["Lemit"] = (9, 8, 7)
    rec["direction"] = np.asarray(direction, np.float32)
    s.lights = rec
    return s


def films(desc, spp, seed):
    """(film in the reference's order, film front to back with the light-space shadow rays, stats of both)"""
    with api.DeviceScene(desc) as dev:
        dev.render(spp, seed=seed)
        reference = dev.download_film()
        ref_stats = dev.stats()   # the counters are those of the last pt_render
        dev.render(spp, seed=seed, ordered=True)
        fast, st = dev.download_film(), dev.stats()
        # and over the wide layout, whose shadow rays then run over the wide nodes in the light's frame
        dev.render(spp, seed=seed, wide=True, wide_shadows=True)
        assert_bits_equal(dev.download_film(), reference, "wide layout in the light's frame")
        assert dev.stats().rays_shadow == ref_stats.rays_shadow
        return reference, fast, ref_stats, st


def check(scene, what, spp=8, seed=4, flags=0):
    desc = scene.to_desc(flags=flags)
    reference, fast, ref_stats, st = films(desc, spp, seed)
    assert_bits_equal(fast, reference, what)
    assert st.rays_shadow == ref_stats.rays_shadow and st.samples == ref_stats.samples
    assert reference[..., :3].max() > 0
    return ref_stats, st


@pytest.mark.parametrize("name", ["cornell", "mixed"])
@pytest.mark.parametrize("direction", list(DIRECTIONS))
def test_film_equals_the_reference_order(name, direction, scenes):
    # `mixed` has spheres (their any-hit test runs in world space) and textures
    check(one_directional_light(scenes[name], DIRECTIONS[direction]), f"{name}, light {direction}")


def test_the_light_space_layout_is_what_ran(scenes):
    """Node counts of the shadow rays differ from the reference's descent when the third layout is in use, and equal
    it when the scene was uploaded without it."""
    scene = one_directional_light(scenes["cornell"], DIRECTIONS["oblique"])
    ref_stats, st = check(scene, "with the layout")
    with_layout = st.nodes_shadow
    ref_stats2, st2 = check(scene, "without the layout", flags=abi.PT_SCENE_NO_PARALLEL_SHADOWS)
    assert st2.nodes_shadow == ref_stats2.nodes_shadow
    assert with_layout != st2.nodes_shadow
    # no direct-light shortcut: the shadow records go through ResolveKernel
    check(scene, "separate summation pass", flags=abi.PT_SCENE_NO_DIRECT_SUM)


def test_two_lights_fall_back_to_the_tree(scenes):
    s = one_directional_light(scenes["cornell"], DIRECTIONS["down"])
    s.lights = np.concatenate([s.lights, s.lights])
    s.lights["direction"][1] = (1, 0, 0)
    desc = s.to_desc()
    reference, fast, ref_stats, st = films(desc, 4, 1)
    assert_bits_equal(fast, reference, "two directional lights")
    assert st.nodes_shadow == ref_stats.nodes_shadow


def test_geometry_far_from_the_origin(scenes):
    """The same box moved by (900, -700, 1300): the projections of the ray origins lose 10 bits, the layout widens
    its rectangles accordingly."""
    base = scenes["cornell"]
    for offset in ((900.0, -700.0, 1300.0), (-65536.0, 0.0, 30000.0)):
        s = one_directional_light(base, DIRECTIONS["oblique"])
        t = np.asarray(offset, np.float32)
        s.vertices = s.vertices + t
        s.nodes = s.nodes.copy()
        s.nodes["aabb_min"] += t
        s.nodes["aabb_max"] += t
        s.camera = dict(s.camera)
        s.camera["position"] = s.camera["position"] + t
        check(s, f"moved by {offset}")
        s.lights["direction"][0] = (0, -1, 0)
        check(s, f"moved by {offset}, light down")


@pytest.mark.parametrize("name,max_shapes", [("cornell", 3), ("mixed", 8), ("mixed", 16)])
def test_leaves_of_several_shapes(name, max_shapes, scenes):
    """Subtrees collapsed into leaves of several shapes: a hit is checked against its leaf's box, which is then the
    box of all the leaf's shapes."""
    from test_gpu_parity import collapse_small_subtrees
    scene = collapse_small_subtrees(one_directional_light(scenes[name], DIRECTIONS["oblique 2"]), max_shapes)
    assert scene.nodes["num_shapes"].max() > 1
    ref_stats, st = check(scene, f"{name}, leaves of up to {max_shapes}")
    # a leaf of more than eight shapes does not fit the layout's record: such a scene keeps the tree
    if scene.nodes["num_shapes"].max() > 8:
        assert st.nodes_shadow == ref_stats.nodes_shadow
    check(one_directional_light(scene, DIRECTIONS["down"]), f"{name}, leaves of up to {max_shapes}, light down")


def test_benchmark_scene_at_its_size():
    from pathtracer_amd import host
    with host.HostScene.atrium(seed=1, triangles=262267, texture_size=64) as hs:
        hs.set_image(960, 540, 960 / 540)
        reference, fast, ref_stats, st = films(hs.desc(), 4, 7)
        assert_bits_equal(fast, reference, "atrium 960x540")
        print(f"shadow rays {st.rays_shadow}: nodes per ray {ref_stats.nodes_shadow / ref_stats.rays_shadow:.1f} in the tree, "
              f"{st.nodes_shadow / st.rays_shadow:.1f} in the light's frame")


@pytest.mark.parametrize("direction", [(0, -1, 0), (0.3, -0.8, 0.5), (-0.6, 0.2, 0.77)])
def test_large_trees(direction):
    """A 2 M triangle soup (4 M nodes, 13 levels of wide nodes) and the knot under one directional light: deep stacks,
    rectangles of a few cells, nodes whose rectangle takes the largest shift"""
    from pathtracer_amd import host
    from test_gpu_layouts import lit_from
    for make, what in ((lambda: host.HostScene.soup(seed=5, triangles=2_000_000, edge=0.01, skybox_size=64), "soup 2M"),
                       (lambda: host.HostScene.knot(seed=2, triangles=871414), "knot")):
        with make() as hs:
            hs.set_image(512, 512, 1.0)
            hs.set_max_depth(3)
            reference, fast, ref_stats, st = films(lit_from(hs, 0, direction), 2, 11)
            assert_bits_equal(fast, reference, f"{what}, light {direction}")
            assert ref_stats.rays_shadow > 100000


def test_the_wide_nodes_in_the_lights_frame_are_what_ran(scenes):
    """Shadow rays of a render over the wide layout: over the wide nodes in the light's frame when the scene has them (other
    node counts than over the wide nodes themselves), the same film either way"""
    scene = one_directional_light(scenes["cornell"], DIRECTIONS["oblique"])
    visits, pictures = {}, {}
    for what, flags in (("light's frame", 0), ("wide nodes", abi.PT_SCENE_NO_PARALLEL_SHADOWS)):
        with api.DeviceScene(scene.to_desc(flags=flags)) as dev:
            assert (len(dev.read_layout(abi.PT_LAYOUT_LIGHT_WIDE_NODES)) > 0) == (flags == 0)
            dev.render(8, seed=4, wide=True, wide_shadows=True)
            pictures[what], visits[what] = dev.download_film(), dev.stats().nodes_shadow
    assert_bits_equal(pictures["light's frame"], pictures["wide nodes"], "shadow rays in the light's frame")
    assert visits["light's frame"] != visits["wide nodes"] and min(visits.values()) > 0
