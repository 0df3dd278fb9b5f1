"""The record arrays that pt_upload_scene's kernels write from the scene description are, byte for byte, the ones its
host loops write (PT_SCENE_HOST_LAYOUTS): reference-order nodes, front-to-back nodes with their octant links, the
intersection and attribute records of the primitives, the wide nodes and their primitive copy, and the light-space
records of a scene with one directional light."""
import numpy as np
import pytest

from pathtracer_amd import abi, api, host

pytestmark = pytest.mark.gpu

LAYOUTS = {"reference nodes": abi.PT_LAYOUT_REFERENCE_NODES, "front-to-back nodes": abi.PT_LAYOUT_ORDERED_NODES,
           "intersection records": abi.PT_LAYOUT_PRIMS, "attribute records": abi.PT_LAYOUT_ATTRIBUTES,
           "wide nodes": abi.PT_LAYOUT_WIDE_NODES, "wide primitives": abi.PT_LAYOUT_WIDE_PRIMS, "light-space nodes": abi.PT_LAYOUT_SHADOW_NODES,
           "light-space top words": abi.PT_LAYOUT_SHADOW_TOP}


def compare(make_desc, what, expect=LAYOUTS):
    with api.DeviceScene(make_desc(0)) as dev, api.DeviceScene(make_desc(abi.PT_SCENE_HOST_LAYOUTS)) as ref:
        sizes = {}
        for name, which in expect.items():
            got, want = dev.read_layout(which), ref.read_layout(which)
            sizes[name] = len(want)
            assert len(got) == len(want), f"{what}: {name}: {len(got)} bytes, the host writes {len(want)}"
            if len(want) == 0:
                continue
            differing = np.flatnonzero(got != want)
            assert differing.size == 0, f"{what}: {name}: {differing.size} bytes differ, the first at {differing[0]}"
        print(f"{what}: equal; bytes " + ", ".join(f"{k} {v}" for k, v in sizes.items()))
        return sizes


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_golden_scenes(name, scenes):
    s = scenes[name].with_image(32, 32)
    sizes = compare(lambda flags: s.to_desc(flags=flags), name)
    assert sizes["reference nodes"] and sizes["front-to-back nodes"] and sizes["intersection records"] and sizes["wide nodes"]


@pytest.mark.parametrize("name,max_shapes", [("cornell", 5), ("mixed", 16)])
def test_leaves_with_several_shapes(name, max_shapes, scenes):
    from test_gpu_parity import collapse_small_subtrees
    s = collapse_small_subtrees(scenes[name].with_image(32, 32), max_shapes)
    compare(lambda flags: s.to_desc(flags=flags), f"{name}, leaves of up to {max_shapes}")


@pytest.mark.parametrize("seed", range(4))
def test_feature_mixes(seed, scenes):
    from test_gpu_parity import mutated_scene
    s = mutated_scene(scenes["mixed"], 77 * seed + 5)
    compare(lambda flags: s.to_desc(flags=flags), f"mixed variation {seed}")


def test_geometry_cases():
    from geometry_cases import cases
    for what, (scene, _, _) in cases().items():
        compare(lambda flags: scene.to_desc(flags=flags), what)


@pytest.mark.parametrize("top_nodes", [1, 2, 3, 64, 1000])
def test_small_tops(top_nodes, scenes):
    """The top of the tree that is stored first: from the root alone to most of the tree"""
    s = scenes["mixed"].with_image(32, 32)

    def make(flags):
        desc = s.to_desc(flags=flags)
        desc.top_nodes = top_nodes
        return desc
    compare(make, f"mixed, top of {top_nodes}")


@pytest.mark.parametrize("name,max_shapes", [("cornell", 1), ("mixed", 1), ("mixed", 8)])
def test_one_directional_light(name, max_shapes, scenes):
    """The light-space records, for lights along an axis (exact projections) and oblique ones; `mixed` has spheres, whose
    extent in the light's frame takes a double-precision square root"""
    from test_gpu_parallel_shadows import DIRECTIONS, one_directional_light
    from test_gpu_parity import collapse_small_subtrees
    for label, direction in DIRECTIONS.items():
        s = one_directional_light(scenes[name], direction)
        if max_shapes > 1:
            s = collapse_small_subtrees(s, max_shapes)
        for top_nodes in (0, 3, 200):
            def make(flags):
                desc = s.to_desc(flags=flags)
                if top_nodes:
                    desc.top_nodes = top_nodes
                return desc
            sizes = compare(make, f"{name}, leaves of up to {max_shapes}, light {label}, top of {top_nodes or 'default'}")
            assert sizes["light-space nodes"] > 0 and sizes["light-space top words"] > 0


def with_flags(hs, flags):
    desc = hs.desc()
    desc.flags |= flags
    return desc


def test_atrium():
    """One directional light: the light-space records are there too"""
    with host.HostScene.atrium(seed=1, triangles=262267, texture_size=64) as hs:
        sizes = compare(lambda flags: with_flags(hs, flags), "atrium")
        assert sizes["light-space nodes"] > 0


def test_knot():
    with host.HostScene.knot(seed=2, triangles=871414) as hs:
        compare(lambda flags: with_flags(hs, flags), "knot")


def test_soup_2m():
    with host.HostScene.soup(seed=3, triangles=2_000_000, edge=0.01, skybox_size=64) as hs:
        compare(lambda flags: with_flags(hs, flags), "soup 2M")


def lit_from(hs, flags, direction):
    """The generated scene with one directional light in place of its own"""
    desc = with_flags(hs, flags)
    light = np.zeros(1, abi.LIGHT_DTYPE)
    light["type"], light["n_samples"], light["prim"] = abi.PT_LIGHT_DIRECTIONAL, 1, -1
    light["Lemit"], light["direction"] = (5, 5, 5), np.asarray(direction, np.float32)
    desc.lights, desc.num_lights = light.ctypes.data, 1
    desc._light = light
    return desc


@pytest.mark.parametrize("direction", [(0, -1, 0), (0.3, -0.8, 0.5)])
def test_large_trees_with_one_directional_light(direction):
    """4 M and 1.7 M nodes: the bounds in the light's frame climb the tree over many levels, and the record indices come
    from a prefix sum over hundreds of blocks"""
    with host.HostScene.soup(seed=5, triangles=2_000_000, edge=0.01, skybox_size=64) as hs:
        sizes = compare(lambda flags: lit_from(hs, flags, direction), f"soup 2M, light {direction}")
        assert sizes["light-space nodes"] >= 16 * (4_000_000 - 1)
    with host.HostScene.knot(seed=2, triangles=871414) as hs:
        sizes = compare(lambda flags: lit_from(hs, flags, direction), f"knot, light {direction}")
        assert sizes["light-space nodes"] > 0


def test_bounds_with_full_fences(monkeypatch):
    """The climb that makes the bounds in the light's frame with agent-scope fences around its arrival counter - what an
    upload falls back to should the fast climb ever fail its check - writes the same bytes"""
    monkeypatch.setenv("PTAMD_SHADOW_FULL_FENCES", "1")
    with host.HostScene.atrium(seed=4, triangles=60000, texture_size=64) as hs:
        sizes = compare(lambda flags: with_flags(hs, flags), "atrium 60 k, fenced climb")
        assert sizes["light-space nodes"] > 0


def test_films_agree(scenes):
    """and a render over each set of records gives the same film"""
    s = scenes["mixed"].with_image(64, 64)
    films = []
    for flags in (0, abi.PT_SCENE_HOST_LAYOUTS):
        with api.DeviceScene(s.to_desc(flags=flags)) as dev:
            for kw in ({}, {"ordered": True}, {"wide": True, "wide_shadows": True}):
                dev.clear_film()
                dev.render(4, seed=2, **kw)
                film, _, _ = dev.readback(abi.PT_TONEMAP_NONE, tonemapped=False)
                films.append(film.copy())
    for f in films[1:]:
        assert (f.view(np.uint32) == films[0].view(np.uint32)).all()
