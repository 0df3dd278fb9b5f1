"""Pins the CPU checker and the host BVH builder against the reference's own compiled code (oracle/_ref, built
by oracle/Makefile.ref from /root/reference).  Skipped where that library does not exist (the GPU box); the
golden vectors in tests/golden carry the same comparisons there."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from oracle import checker, ref
from pathtracer_amd import abi, host

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libptref.so not built (no /root/reference)")


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_radiance_bit_identical_to_reference_li(name, scenes):
    small = scenes[name].with_image(40, 30)
    desc = small.to_desc()
    want, rays = ref.RefScene(desc).li_image(6, seed=99)
    got, stats = checker.OracleScene(desc).li_image(6, seed=99)
    assert_bits_equal(got, want, "radiance")
    assert stats["rays_closest"] == rays["rays_closest"] and stats["rays_shadow"] == rays["rays_shadow"]


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_traversal_bit_identical_on_random_rays(name, scenes):
    s = scenes[name]
    desc = s.to_desc()
    rs, orc = ref.RefScene(desc), checker.OracleScene(desc)
    rng = np.random.default_rng(17)
    n = 60_000
    lo, hi = s.nodes["aabb_min"][0], s.nodes["aabb_max"][0]
    o = rng.uniform(lo - 0.3, hi + 0.3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:5000, 0] = 0.0
    want, got = rs.intersect(o, d), orc.intersect(o, d)
    assert np.array_equal(got["prim"], want["prim"])
    hit = want["prim"] >= 0
    for key in ("t", "nodes", "prim_tests", "material"):
        assert_bits_equal(got[key], want[key], key)
    for key in ("u", "v", "position", "normal", "tangent", "bitangent", "uv"):
        assert_bits_equal(got[key][hit], want[key][hit], key)
    tmax = rng.uniform(0, 3, n).astype(np.float32)
    assert np.array_equal(orc.occluded(o, d, tmax)["occluded"], rs.occluded(o, d, tmax))


def test_sample_loop_matches_reference_render(scenes):
    """The reference's own PathTracer::Render against the harness' sample loop + tonemap.

    Render reads `ray.position` before `ray` is constructed (src/path_tracer.cpp:189).  In this build of the
    reference that stack slot holds the previous pixel's colour at the first sample of a pixel and the camera
    position afterwards; with that emulated the two agree bit for bit everywhere but in the very first pixel,
    whose first sample reads uninitialised memory.  The product restates the evident intent (camera position).
    """
    small = scenes["cornell"].with_image(24, 24)
    rs = ref.RefScene(small.to_desc())
    for spp in (1, 5):
        rendered = rs.render([0.0, 0.0, 0.0], spp, seed=4)
        sums, _ = rs.li_image(spp, seed=4, emulate_stale_origin=True)
        mine, _ = ref.tonemap((sums / np.float32(spp)).reshape(-1, 3), float(small.camera["exposure"]), float(small.camera["gamma"]))
        mine = mine.reshape(24, 24, 3)
        same = (mine == rendered).all(axis=2)
        assert same.reshape(-1)[1:].all(), f"{(~same).sum()} pixels differ at {spp} spp"


def test_host_tonemap_pass_is_the_reference_pass_bit_for_bit():
    """host.tonemap (the command line's --host-tonemap) against the reference's Uncharted2Tonemap / ReinhardTonemap,
    GammaCorrect and clamp: same floats, including the 0.06 % of values where powf is not correctly rounded."""
    rng = np.random.default_rng(31)
    rgb = np.concatenate([rng.uniform(0, 30, (200000, 3)), np.exp(rng.uniform(-20, 3, (200000, 3))), np.zeros((1, 3))]).astype(np.float32)
    for exposure, gamma in ((1.0, 2.2), (6.0, 1.0), (0.3, 1.8)):
        u2, reinhard = ref.tonemap(rgb, exposure, gamma)
        assert host.tonemap(rgb, exposure, gamma, 0).tobytes() == u2.tobytes()
        assert host.tonemap(rgb, exposure, gamma, 1).tobytes() == reinhard.tobytes()


def test_tonemap_bit_identical():
    rng = np.random.default_rng(3)
    rgb = rng.uniform(0, 20, (5000, 3)).astype(np.float32)
    for exposure, gamma in ((1.0, 2.2), (6.0, 1.0)):
        a, b = ref.tonemap(rgb, exposure, gamma)
        assert_bits_equal(checker.tonemap(rgb, exposure, gamma, abi.PT_TONEMAP_UNCHARTED2), a, "uncharted2")
        assert_bits_equal(checker.tonemap(rgb, exposure, gamma, abi.PT_TONEMAP_REINHARD), b, "reinhard")


@pytest.mark.parametrize("name", ["cornell", "mixed"])
@pytest.mark.parametrize("method", [abi.PTH_SPLIT_SAH, abi.PTH_SPLIT_MIDDLE, abi.PTH_SPLIT_EQUAL_COUNTS])
def test_bvh_builder_matches_reference_build(name, method, scenes):
    """Node for node and shape for shape, the host builder equals BVH::Build (src/bvh.cpp:204-228)."""
    s = scenes[name]
    want_nodes, want_order = ref.bvh_build(s.to_desc(), method)
    boxes = shape_boxes(s)
    nodes, order = host.build_bvh(boxes, method)
    assert np.array_equal(order, want_order)
    assert len(nodes) == len(want_nodes)
    for field in ("aabb_min", "aabb_max", "offset", "num_shapes", "axis"):
        assert np.array_equal(nodes[field], want_nodes[field]), field


def shape_boxes(s):
    """World-space boxes of the shapes in their flattened order (Triangle / Sphere::WorldSpaceAABB)."""
    boxes = np.zeros((len(s.prims), 6), np.float32)
    for k, p in enumerate(s.prims):
        if p["type"] == abi.PT_PRIM_TRIANGLE:
            t = s.triangles[p["index"]]
            base = s.meshes[t["mesh"]]["vertex_offset"]
            v = s.vertices[[base + t["i0"], base + t["i1"], base + t["i2"]]]
            boxes[k, :3], boxes[k, 3:] = v.min(axis=0), v.max(axis=0)
        else:
            sp = s.spheres[p["index"]]
            boxes[k, :3], boxes[k, 3:] = sp["position"] - sp["radius"], sp["position"] + sp["radius"]
    return boxes


def test_camera_and_sphere_transform_match_glm(scenes):
    """Camera::UpdateOrientationVectors through the reference's Render (it refuses a camera whose vectors do not
    reproduce) for a rotated camera; the mixed scene's camera is rotated by 10 degrees about y."""
    small = scenes["mixed"].with_image(8, 8)
    rs = ref.RefScene(small.to_desc())
    rs.render(np.radians([0.0, 10.0, 0.0]).astype(np.float32), 1, seed=0)   # raises on a mismatch
    view, up, right = host.camera_vectors(np.radians([0.0, 10.0, 0.0]).astype(np.float32))
    assert_bits_equal(view, small.camera["view_dir"], "view")
    assert_bits_equal(right, small.camera["right_dir"], "right")


def test_benchmark_scene_bit_identical_to_reference_li():
    """The benchmark's scene (262,267 triangles, 24 textures, directional light, depth 5) through the reference's
    own Li / BVH::Intersect / Triangle::Intersect / Texture::GetPixel, against the checker: every pixel equal."""
    with host.HostScene.atrium(seed=1, triangles=262267, texture_size=1024) as hs:
        hs.set_image(240, 135, 240 / 135)
        desc = hs.desc()
        rs, orc = ref.RefScene(desc), checker.OracleScene(desc)
        want, rstats = rs.li_image(2, seed=5)
        got, ostats = orc.li_image(2, seed=5)
        assert_bits_equal(got, want, "atrium radiance sums")
        assert want.max() > 0
        assert ostats["rays_closest"] == rstats["rays_closest"] and ostats["rays_shadow"] == rstats["rays_shadow"]
        rng = np.random.default_rng(3)
        o = np.array([[-9.0, 2.0, 0.0]], np.float32) + rng.uniform(-1, 1, (20000, 3)).astype(np.float32)
        d = rng.normal(size=(20000, 3)).astype(np.float32)
        a, b = rs.intersect(o, d), orc.intersect(o, d)
        assert np.array_equal(a["prim"], b["prim"]) and (a["prim"] >= 0).mean() > 0.5
        assert_bits_equal(a["t"], b["t"], "t")
        assert np.array_equal(a["nodes"], b["nodes"]) and np.array_equal(a["prim_tests"], b["prim_tests"])
        rs.close()


def test_cornell_512_16spp_cpu_configuration(scenes):
    """The CPU-only configuration of the benchmark list: resources/scenes/cornell.json at 512 x 512, 16 samples
    per pixel, depth 5 (4.19 M pixel samples), the reference's own code against the checker, bit for bit."""
    scene = scenes["cornell"].with_image(512, 512)
    assert scene.max_depth == 5
    desc = scene.to_desc()
    want, rays = ref.RefScene(desc).li_image(16, seed=16)
    got, stats = checker.OracleScene(desc).li_image(16, seed=16)
    assert_bits_equal(got, want, "radiance sums")
    assert stats["samples"] == 512 * 512 * 16
    assert stats["rays_closest"] == rays["rays_closest"] and stats["rays_shadow"] == rays["rays_shadow"]
    assert 0.05 < float(got.mean()) / 16 < 1.0


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_random_feature_mixes_bit_identical_to_reference_li(name, seed, scenes):
    """The scene variations of the GPU parity test (several point / directional lights, area lights on triangles
    and spheres with 1-3 samples, emissive materials, depth 1-6) through the reference's own Li.  The cube map is
    replaced by a constant background here: the reference's Skybox does not build in this container."""
    from test_gpu_parity import mutated_scene
    scene = mutated_scene(scenes[name], 1000 * seed + len(name))
    scene.skybox, scene.skybox_width, scene.skybox_height = np.zeros(0, np.float32), 0, 0
    scene.background_radiance = np.float32([0.3, 0.2, 0.1])
    desc = scene.to_desc()
    want, rays = ref.RefScene(desc).li_image(5, seed=seed)
    got, stats = checker.OracleScene(desc).li_image(5, seed=seed)
    assert_bits_equal(got, want, "radiance")
    assert stats["rays_closest"] == rays["rays_closest"] and stats["rays_shadow"] == rays["rays_shadow"]


def test_tie_breaking_and_rounding_cases_bit_identical_to_reference():
    """Coincident triangles, shared edges and vertices, zero-area triangles, very large and very small
    coordinates (tests/geometry_cases.py): hit primitive, t, u, v and the per-ray counts of the reference's
    BVH::Intersect / Occluded against the checker's."""
    from geometry_cases import cases
    for what, (scene, o, d) in cases().items():
        desc = scene.to_desc()
        want, got = ref.RefScene(desc).intersect(o, d), checker.OracleScene(desc).intersect(o, d)
        hit = want["prim"] >= 0
        print(f"{what}: {len(scene.prims)} triangles, {hit.mean():.2f} of the rays hit")
        assert hit.mean() > 0.05, what
        assert np.array_equal(got["prim"], want["prim"]), what
        for key in ("t", "nodes", "prim_tests"):
            assert_bits_equal(got[key], want[key], f"{what}: {key}")
        for key in ("u", "v"):
            assert_bits_equal(got[key][hit], want[key][hit], f"{what}: {key}")
        tmax = np.where(hit, np.minimum(want["t"], np.float32(1e30)) * np.float32(1.5), np.float32(10)).astype(np.float32)
        tmax[::2] = want["t"][::2]        # exactly the hit distance: strict < must not report the hit itself
        assert np.array_equal(checker.OracleScene(desc).occluded(o, d, tmax)["occluded"], ref.RefScene(desc).occluded(o, d, tmax)), what
        img_ref, _ = ref.RefScene(desc).li_image(3, seed=5)      # paths bouncing between these triangles under a constant sky
        img, _ = checker.OracleScene(desc).li_image(3, seed=5)
        assert_bits_equal(img, img_ref, f"{what}: radiance")
        assert img.mean() > 0.01, what


def test_tga_decoding_equals_the_reference_texture_load():
    """Every TGA the reference ships (the Sponza textures: 24 and 32 bit, run-length encoded and plain), decoded
    by the host library and by the reference's Texture::Load (stb_image), flipped and unflipped: equal bytes."""
    import glob
    files = sorted(glob.glob("/root/reference/resources/**/*.tga", recursive=True))
    assert len(files) >= 30
    for f in files:
        for flip in (True, False):
            mine, theirs = host.read_image(f, flip), ref.load_texture(f, flip=flip)
            assert mine.shape == theirs.shape and np.array_equal(mine, theirs), (f, flip)


def test_jpeg_decoding_equals_the_reference_texture_load(tmp_path):
    """Baseline JPEG through the host library and through the reference's Texture::Load (stb_image): the seven
    files the reference ships (4:2:0) and generated ones - 4:4:4, 4:2:2, 4:2:0, grey, odd sizes down to 1 x 1,
    qualities 5 to 100, optimised Huffman tables, restart markers, and each of those as a progressive file (DC
    and AC scans with successive approximation, as libjpeg's standard script orders them), and four-component
    files (Adobe CMYK, YCCK, unmarked) - equal bytes."""
    import glob
    files = sorted(glob.glob("/root/reference/resources/**/*.jpg", recursive=True))
    assert len(files) >= 7
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:97, 0:131]
    smooth = np.stack([128 + 120 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 128 + 100 * np.cos(xx / 5.0 + yy / 11.0), (xx * 2 + yy) % 256], axis=-1)
    pictures = {"smooth": smooth, "noise": rng.integers(0, 256, (97, 131, 3)), "edges": (rng.integers(0, 2, (97, 131, 3)) * 255)}
    n = 0
    for name, pic in pictures.items():
        pic = np.clip(pic, 0, 255).astype(np.uint8)
        for (w, h) in ((131, 97), (64, 64), (17, 9), (8, 8), (1, 1), (33, 1), (1, 20)):
            img = PIL.fromarray(pic[:h, :w])
            for subsampling in (0, 1, 2):
                for quality, extra in ((75, {}), (100, {}), (5, {}), (90, {"optimize": True}), (60, {"restart_marker_blocks": 3}),
                                       (75, {"progressive": True}), (100, {"progressive": True}), (5, {"progressive": True}),
                                       (60, {"progressive": True, "restart_marker_blocks": 3})):
                    f = str(tmp_path / f"c{n}.jpg")
                    n += 1
                    img.save(f, "JPEG", quality=quality, subsampling=subsampling, **extra)
                    files.append(f)
            f = str(tmp_path / f"g{n}.jpg")
            n += 1
            img.convert("L").save(f, "JPEG", quality=80)
            files.append(f)
            f = str(tmp_path / f"gp{n}.jpg")
            n += 1
            img.convert("L").save(f, "JPEG", quality=80, progressive=True)
            files.append(f)
    # four components: CMYK as Adobe writes it, and the same data declared YCCK / YCbCr + unknown by the marker
    cmyk = PIL.fromarray(np.clip(np.dstack([pictures["smooth"], pictures["noise"][..., :1]]), 0, 255).astype(np.uint8), "CMYK")
    for subsampling in (0, 1, 2):
        for extra in ({}, {"progressive": True}):
            f = str(tmp_path / f"k{n}.jpg")
            n += 1
            cmyk.save(f, "JPEG", quality=85, subsampling=subsampling, **extra)
            files.append(f)
            data = bytearray(open(f, "rb").read())
            at = data.find(b"Adobe") + 11
            assert data[at] == 0
            for transform in (1, 2):
                data[at] = transform
                files.append(str(tmp_path / f"k{n}_{transform}.jpg"))
                open(files[-1], "wb").write(bytes(data))
    for f in files:
        for flip in (True, False):
            mine, theirs = host.read_image(f, flip), ref.load_texture(f, flip=flip)
            assert mine.shape == theirs.shape and np.array_equal(mine, theirs), (f, flip)
    assert sum(b"\xff\xc2" in open(f, "rb").read()[:1024] for f in files) >= 250   # progressive frame headers


def test_reference_texture_scene_loads_and_renders_bit_identically():
    """resources/scenes/texture.json (a sphere, a textured plane, the JPEG texture the reference ships) through
    the host loader: the texel pool equals the reference's decode, and the reference's Li over the flattened
    scene equals the checker's."""
    res = "/root/reference/resources/"
    with host.HostScene.from_file(res + "scenes/texture.json", res) as hs:
        scene = hs.to_data()
    assert len(scene.textures) == 1 and (int(scene.textures["width"][0]), int(scene.textures["height"][0])) == (220, 220)
    assert np.array_equal(scene.texels.reshape(220, 220, 4), ref.load_texture(res + "images/cockatoo.jpg", flip=True))
    # As shipped the file renders black: its light uses the stale key "color" and stays dark, and its sphere's
    # "transform" block is not a key of the parser, so the unit sphere sits around the camera.  Give the light
    # power and move the camera out to compare something; then the file as it is.
    lit = scene.with_image(64, 64)
    lit.lights = lit.lights.copy()
    lit.lights["Lemit"][0] = np.float32([3, 3, 3])
    lit.camera["position"] = np.float32([0, 0.5, 4])
    for s, nonzero in ((lit, True), (scene.with_image(64, 64), False)):
        desc = s.to_desc()
        want, rays = ref.RefScene(desc).li_image(4, seed=12)
        got, stats = checker.OracleScene(desc).li_image(4, seed=12)
        assert_bits_equal(got, want, "radiance")
        assert stats["rays_closest"] == rays["rays_closest"] and (got.mean() > 0.01) == nonzero


def test_cube_map_faces_from_the_reference_files(tmp_path):
    """The water cube map of resources/scenes/dragon.json (six 2048 x 2048 JPEG files): the host loader's float4
    faces equal byte / 255 of the reference's decode (stbi_loadf with gamma and scale 1, src/resource/skybox.cpp:34-37),
    in the face order right, left, top, bottom, back, front."""
    res = "/root/reference/resources/"
    scene_file = tmp_path / "sky.json"
    scene_file.write_text('''{ "Skybox": { "name": "water", "right": "images/water/right.jpg", "left": "images/water/left.jpg",
      "top": "images/water/top.jpg", "bottom": "images/water/bottom.jpg", "back": "images/water/back.jpg",
      "front": "images/water/front.jpg", "flipVertically": false },
      "Material": { "name": "grey", "albedo": [ 0.5, 0.5, 0.5 ] },
      "Sphere": { "material": "grey", "position": [ 0, 0, -5 ], "radius": 1 } }''')
    with host.HostScene.from_file(str(scene_file), res) as hs:
        scene = hs.to_data()
    assert (scene.skybox_width, scene.skybox_height) == (2048, 2048)
    faces = scene.skybox.reshape(6, 2048, 2048, 4)
    for k, name in enumerate(("right", "left", "top", "bottom", "back", "front")):
        want = ref.load_texture(res + f"images/water/{name}.jpg", flip=False).astype(np.float32) / np.float32(255)
        assert np.array_equal(faces[k], want), name


def test_png_decoding_equals_the_reference_texture_load(tmp_path):
    """PNG through the host library and through the reference's Texture::Load (stb_image): every colour type at
    every bit depth it allows, interlaced and not, each row filter, colour keys and palette transparency,
    several IDAT chunks, odd sizes down to 1 x 1 (tests/png_writer.py), and files written by Pillow with
    different deflate settings (stored, fixed and dynamic Huffman blocks) - equal bytes."""
    from png_writer import write_png, CHANNELS
    rng = np.random.default_rng(4)
    files = []

    def new():
        files.append(str(tmp_path / f"{len(files)}.png"))
        return files[-1]

    for (w, h) in ((37, 29), (8, 8), (1, 1), (5, 1), (1, 7), (9, 16)):
        for colour_type, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
            for depth in depths:
                top = min(1 << depth, 1 << 16)
                samples = rng.integers(0, top, (h, w, CHANNELS[colour_type]))
                samples[: h // 2] = (np.arange(w)[None, :, None] * 3 + np.arange(h // 2)[:, None, None]) % top   # smooth half: filters matter
                kw = {}
                if colour_type == 3:
                    kw["palette"] = rng.integers(0, 256, (top, 3))
                for interlace in (False, True):
                    write_png(new(), samples, colour_type, depth, interlace=interlace, idat_split=(0 if interlace else 40), **kw)
                    if colour_type == 3:
                        write_png(new(), samples, colour_type, depth, interlace=interlace, trns=rng.integers(0, 256, top // 2 + 1).astype(np.uint8), **kw)
                    elif colour_type in (0, 2):
                        key = samples[h - 1, w - 1]
                        write_png(new(), samples, colour_type, depth, interlace=interlace, trns=b"".join(int(v).to_bytes(2, "big") for v in key))
                if (w, h) == (37, 29):
                    for f in range(5):
                        write_png(new(), samples, colour_type, depth, row_filter=f, **kw)
    PIL = pytest.importorskip("PIL.Image")
    picture = rng.integers(0, 256, (40, 56, 4)).astype(np.uint8)
    picture[:20] = (np.arange(56)[None, :, None] * 4) % 256
    for mode in ("RGBA", "RGB", "L", "LA", "P", "1"):
        for kw in ({}, {"optimize": True}, {"compress_level": 0}, {"compress_level": 1}, {"compress_level": 9}):
            PIL.fromarray(picture).convert(mode).save(new(), **kw)
    assert len(files) > 400
    for f in files:
        for flip in (True, False):
            mine, theirs = host.read_image(f, flip), ref.load_texture(f, flip=flip)
            assert mine.shape == theirs.shape and np.array_equal(mine, theirs), (f, flip)


def writer_pictures():
    """Float pictures for the image writers: noise beyond [0, 1], gradients, long runs and runs of exactly three
    (run-length packets, deflate matches), flat black and white, a smooth shaded picture of rendered size, and
    sizes that leave partial 8 x 8 blocks, one-pixel rows and rows longer than a 128-pixel packet."""
    rng = np.random.default_rng(0)
    pics = []
    for (w, h) in ((53, 37), (1, 1), (2, 1), (1, 2), (3, 3), (130, 5), (257, 4), (300, 2), (8, 8), (16, 16), (64, 48)):
        p = rng.uniform(-0.2, 1.2, (h, w, 3)).astype(np.float32)
        pics.append(p)
        q = p.copy()
        q[: h // 2 + 1] = np.linspace(0, 1, w, dtype=np.float32)[None, :, None]
        pics.append(q)
        r = np.round(rng.uniform(0, 1, (h, w, 1)) * 3) / 3 * np.ones(3)
        r[:, : w // 2] = 0.5
        pics.append(r.astype(np.float32))
        pics.append(np.repeat(rng.integers(0, 2, (h, (w + 2) // 3, 3)), 3, axis=1)[:, :w].astype(np.float32))
    pics.append(np.zeros((19, 21, 3), np.float32))
    pics.append(np.ones((19, 21, 3), np.float32))
    y, x = np.mgrid[0:360, 0:640].astype(np.float32)
    shaded = np.stack([0.5 + 0.5 * np.sin(x / 47 + y / 91), (x / 640) * (y / 360), 0.3 + 0.2 * np.cos(y / 13)], axis=-1)
    shaded += rng.normal(0, 0.02, shaded.shape)            # the grain of a path-traced picture
    shaded[100:200, 200:400] = (0.9, 0.1, 0.1)             # a flat surface with hard edges
    pics.append(shaded.astype(np.float32))
    return pics


@pytest.mark.parametrize("ext", ["png", "tga", "bmp", "jpg", "jpeg"])
def test_image_writers_equal_the_reference_image_save(ext, tmp_path):
    """Every file format Image::Save writes (src/image.cpp:43-100, stb_image_write 1.14 underneath): the host
    library's file equals the reference's byte for byte - PNG row filter choice and deflate matches, TGA packets,
    BMP header fields, JPEG tables, coefficients and entropy coding - and reads back to the quantised pixels
    (exactly, but for JPEG)."""
    theirs, mine = str(tmp_path / f"theirs.{ext}"), str(tmp_path / f"mine.{ext}")
    for n, pic in enumerate(writer_pictures()):
        ref.save_image(theirs, pic)
        host.write_image(mine, pic)
        a, b = open(theirs, "rb").read(), open(mine, "rb").read()
        assert a == b, (ext, n, pic.shape, len(a), len(b), next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), -1))
        back = host.read_image(mine, False)[..., :3]
        quantised = (255.0 * np.minimum(np.maximum(pic, np.float32(0)), np.float32(1))).astype(np.uint8)
        if ext in ("jpg", "jpeg"):
            assert back.shape == quantised.shape
            assert np.abs(back.astype(int) - quantised).mean() < 6
        else:
            assert np.array_equal(back, quantised), (ext, n)


def test_cube_map_face_decoding_equals_the_reference_float_load(tmp_path):
    """A cube map face through the host library and through the calls Skybox::Load makes (stbi_loadf, four
    channels, gamma and scale 1, src/resource/skybox.cpp:33-37): Radiance .hdr files - run-length coded and flat,
    widths on both sides of the 8-pixel limit of the run-length coding, both identifier lines, black pixels with
    a zero exponent, radiance from 1e-6 to 1e6 - give the same floats bit for bit, flipped or not; so do 8-bit
    formats (byte / 255).  A scene whose Skybox names six .hdr faces loads them in the reference's order."""
    from hdr_writer import to_rgbe, write_hdr
    rng = np.random.default_rng(6)
    files = []
    for n, (w, h) in enumerate(((40, 12), (8, 5), (7, 9), (1, 1), (300, 3), (129, 4))):
        radiance = np.exp(rng.uniform(np.log(1e-6), np.log(1e6), (h, w, 1))) * rng.uniform(0, 1, (h, w, 3))
        radiance[h // 2:, : w // 2] = radiance[h // 2, 0]      # runs
        radiance[0, 0] = 0                                     # zero exponent byte
        rgbe = to_rgbe(radiance)
        for run_length in (True, False):
            for identifier in ("#?RADIANCE", "#?RGBE"):
                files.append(str(tmp_path / f"{n}_{int(run_length)}_{identifier[2:]}.hdr"))
                write_hdr(files[-1], rgbe, run_length, identifier)
    picture = rng.uniform(0, 1, (9, 14, 3)).astype(np.float32)
    for ext in ("png", "tga", "bmp", "jpg"):
        files.append(str(tmp_path / f"ldr.{ext}"))
        host.write_image(files[-1], picture)
    seen_large = False
    for f in files:
        for flip in (True, False):
            mine, theirs = host.read_image_float(f, flip), ref.load_skybox_face(f, flip)
            assert mine.shape == theirs.shape and mine.tobytes() == theirs.tobytes(), (f, flip)
            seen_large |= bool(mine[..., :3].max() > 100)
            # the same file asked for as a texture: gamma 1 / 2.2, times 255, rounded, clamped
            assert np.array_equal(host.read_image(f, flip), ref.load_texture(f, flip=flip)), (f, flip)
    assert seen_large
    with open(tmp_path / "truncated.hdr", "wb") as f:
        f.write(open(files[0], "rb").read()[:90])
    with pytest.raises(host.HostError):
        host.read_image_float(str(tmp_path / "truncated.hdr"))

    names = ("right", "left", "top", "bottom", "back", "front")
    for k, name in enumerate(names):
        write_hdr(str(tmp_path / f"{name}.hdr"), to_rgbe(np.full((16, 16, 3), 0.25 * (k + 1)) + rng.uniform(0, 30, (16, 16, 3))))
    (tmp_path / "sky.json").write_text('{ "Skybox": { "name": "sky", ' + ", ".join(f'"{n}": "{n}.hdr"' for n in names) + ''' },
      "Material": { "name": "grey", "albedo": [ 0.5, 0.5, 0.5 ] },
      "Sphere": { "material": "grey", "position": [ 0, 0, -5 ], "radius": 1 } }''')
    with host.HostScene.from_file(str(tmp_path / "sky.json"), str(tmp_path) + "/") as hs:
        scene = hs.to_data()
    faces = scene.skybox.reshape(6, 16, 16, 4)
    for k, name in enumerate(names):
        assert np.array_equal(faces[k], ref.load_skybox_face(str(tmp_path / f"{name}.hdr"), True)), name
    assert faces.max() > 20


def test_bmp_decoding_equals_the_reference_texture_load(tmp_path):
    """BMP through the host library and through the reference's Texture::Load (stb_image): 1, 4 and 8 bit
    palettes, 16 bit 5-5-5 and 5-6-5 / 4-4-4-4 masks, 24 bit, 32 bit with used, unused (all zero) and masked
    alpha, core (OS/2, 24 bit), info, V4 and V5 headers, bottom-up and top-down rows, odd widths - equal bytes.
    (A palette under a core header is outside the comparison: the reference's decoder miscounts its entries and
    returns uninitialised colours for the last of them.)"""
    import struct
    rng = np.random.default_rng(9)
    files = []

    def new():
        files.append(str(tmp_path / f"{len(files)}.bmp"))
        return files[-1]

    def bmp(w, h, bpp, rows, palette=b"", header=40, compression=0, masks=(), top_down=False):
        """rows: h byte strings, top row first, unpadded."""
        stride = (len(rows[0]) + 3) & ~3
        body = b"".join(r + bytes(stride - len(r)) for r in (rows if top_down else rows[::-1]))
        if header == 12:
            info = struct.pack("<IHHHH", 12, w, h, 1, bpp)
        else:
            info = struct.pack("<IiiHHIIiiII", header, w, -h if top_down else h, 1, bpp, compression, 0, 2835, 2835, 0, 0)
            if header == 40:
                info += b"".join(struct.pack("<I", m) for m in masks[:3])
            else:
                padded = (tuple(masks) + (0, 0, 0, 0))[:4]
                info += b"".join(struct.pack("<I", m) for m in padded)
                info += bytes(header - 40 - 16)
        offset = 14 + len(info) + len(palette)
        return b"BM" + struct.pack("<IHHI", offset + len(body), 0, 0, offset) + info + palette + body

    for (w, h) in ((13, 7), (8, 8), (1, 1), (5, 2), (33, 3)):
        for top_down in (False, True):
            # palettes
            for bpp in (1, 4, 8):
                n = 1 << bpp
                index = rng.integers(0, n, (h, w))
                per_byte = 8 // bpp
                packed = np.zeros((h, (w + per_byte - 1) // per_byte), np.uint8)
                for i in range(w):
                    packed[:, i // per_byte] |= (index[:, i] << (8 - bpp * (i % per_byte + 1))).astype(np.uint8)
                colours = rng.integers(0, 256, (n, 3)).astype(np.uint8)
                rows = [packed[r].tobytes() for r in range(h)]
                open(new(), "wb").write(bmp(w, h, bpp, rows, b"".join(bytes(c) + b"\0" for c in colours), top_down=top_down))
            # 16 bit
            words = rng.integers(0, 1 << 16, (h, w)).astype("<u2")
            rows = [words[r].tobytes() for r in range(h)]
            open(new(), "wb").write(bmp(w, h, 16, rows, top_down=top_down))
            open(new(), "wb").write(bmp(w, h, 16, rows, compression=3, masks=(0xF800, 0x07E0, 0x001F), top_down=top_down))
            open(new(), "wb").write(bmp(w, h, 16, rows, header=108, compression=3, masks=(0x0F00, 0x00F0, 0x000F, 0xF000), top_down=top_down))
            # 24 bit
            px = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            open(new(), "wb").write(bmp(w, h, 24, [px[r].tobytes() for r in range(h)], top_down=top_down))
            open(new(), "wb").write(bmp(w, h, 24, [px[r].tobytes() for r in range(h)], header=124, top_down=top_down))
            if not top_down:
                open(new(), "wb").write(bmp(w, h, 24, [px[r].tobytes() for r in range(h)], header=12))
            # 32 bit
            px = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
            rows = [px[r].tobytes() for r in range(h)]
            open(new(), "wb").write(bmp(w, h, 32, rows, top_down=top_down))
            px[..., 3] = 0
            zero_alpha = [px[r].tobytes() for r in range(h)]
            open(new(), "wb").write(bmp(w, h, 32, zero_alpha, top_down=top_down))                      # becomes opaque
            open(new(), "wb").write(bmp(w, h, 32, zero_alpha, header=56, top_down=top_down))
            open(new(), "wb").write(bmp(w, h, 32, zero_alpha, header=108, compression=3, masks=(0xFF0000, 0xFF00, 0xFF, 0xFF000000), top_down=top_down))   # stays transparent
            open(new(), "wb").write(bmp(w, h, 32, rows, compression=3, masks=(0xFF, 0xFF00, 0xFF0000), top_down=top_down))
            open(new(), "wb").write(bmp(w, h, 32, rows, header=124, compression=3, masks=(0xFF000000, 0x00FF0000, 0x0000FF00, 0x000000FF), top_down=top_down))
    PIL = pytest.importorskip("PIL.Image")
    picture = rng.integers(0, 256, (21, 30, 4)).astype(np.uint8)
    for mode in ("RGB", "RGBA", "L", "P", "1"):
        PIL.fromarray(picture).convert(mode).save(new())
    picture = rng.uniform(0, 1, (9, 14, 3)).astype(np.float32)
    host.write_image(new(), picture)
    assert len(files) > 140
    for f in files:
        for flip in (True, False):
            mine, theirs = host.read_image(f, flip), ref.load_texture(f, flip=flip)
            assert mine.shape == theirs.shape and np.array_equal(mine, theirs), (f, flip)
    open(tmp_path / "rle.bmp", "wb").write(bmp(4, 4, 8, [bytes(4)] * 4, bytes(1024), compression=1))
    with pytest.raises(host.HostError):
        host.read_image(str(tmp_path / "rle.bmp"))


def test_generated_tga_files_decode_like_the_reference(tmp_path):
    """TGA variants the reference's files do not have, through the host library and the reference's
    Texture::Load: colour-mapped (palettes of 8, 15, 16, 24, 32 bit entries, 8 and 16 bit indices, indices past
    the palette), 15 / 16 bit true colour, grey and grey + alpha, each raw and run-length coded with packets that
    cross rows, top and bottom origin, an image id field - equal bytes."""
    import struct
    rng = np.random.default_rng(10)
    files = []

    def new():
        files.append(str(tmp_path / f"{len(files)}.tga"))
        return files[-1]

    def packets(pixels, size):
        """random raw and run packets over the whole pixel stream (runs repeat the packet's first pixel)"""
        out, i, n = bytearray(), 0, len(pixels) // size
        while i < n:
            count = min(int(rng.integers(1, 129)), n - i)
            if rng.integers(0, 2):
                out += bytes((128 + count - 1,)) + pixels[i * size:(i + 1) * size]
            else:
                out += bytes((count - 1,)) + pixels[i * size:(i + count) * size]
            i += count
        return bytes(out)

    def tga(w, h, type_, bpp, data, palette=b"", palette_len=0, palette_bits=0, top=False, ident=b""):
        head = struct.pack("<BBBHHBHHHHBB", len(ident), 1 if palette_len else 0, type_, 0, palette_len, palette_bits, 0, 0, w, h, bpp, 0x20 if top else 0)
        return head + ident + palette + data

    for (w, h) in ((19, 11), (1, 1), (128, 3), (5, 40)):
        n = w * h
        for top in (False, True):
            for bpp in (15, 16, 24, 32):                                    # true colour
                size = (bpp + 7) // 8
                px = rng.integers(0, 256, n * size).astype(np.uint8).tobytes()
                open(new(), "wb").write(tga(w, h, 2, bpp, px, top=top, ident=b"id" if bpp == 24 else b""))
                open(new(), "wb").write(tga(w, h, 10, bpp, packets(px, size), top=top))
            for bpp in (8, 16):                                             # grey, grey + alpha
                size = bpp // 8
                px = rng.integers(0, 256, n * size).astype(np.uint8).tobytes()
                open(new(), "wb").write(tga(w, h, 3, bpp, px, top=top))
                open(new(), "wb").write(tga(w, h, 11, bpp, packets(px, size), top=top))
            for palette_bits in (8, 15, 16, 24, 32):                        # colour-mapped
                for index_bits, entries in ((8, 200), (16, 700)):
                    palette = rng.integers(0, 256, entries * ((palette_bits + 7) // 8)).astype(np.uint8).tobytes()
                    index = rng.integers(0, entries + 20, n).astype("<u2" if index_bits == 16 else np.uint8).tobytes()
                    open(new(), "wb").write(tga(w, h, 1, index_bits, index, palette, entries, palette_bits, top=top))
                    open(new(), "wb").write(tga(w, h, 9, index_bits, packets(index, index_bits // 8), palette, entries, palette_bits, top=top))
    PIL = pytest.importorskip("PIL.Image")
    picture = rng.integers(0, 4, (21, 30, 4)).astype(np.uint8) * 80
    for mode in ("RGB", "RGBA", "L", "LA", "P"):
        for kw in ({}, {"compression": "tga_rle"}):
            PIL.fromarray(picture).convert(mode).save(new(), **kw)
    assert len(files) > 250
    for f in files:
        for flip in (True, False):
            mine, theirs = host.read_image(f, flip), ref.load_texture(f, flip=flip)
            assert mine.shape == theirs.shape and np.array_equal(mine, theirs), (f, flip)


@pytest.mark.parametrize("name", ["cornell", "mixed"])
def test_reference_side_binding_flattens_a_scene_to_the_same_render(name, scenes):
    """oracle/pt_flatten.hpp is the file a maintainer of the reference adds: PT::Scene -> PtSceneDesc.  On the
    PT::Scene the harness builds from a fixture (meshes, spheres, textures, every kind of light), it returns the
    BVH nodes byte for byte, the shapes in bvh.shapes order, the lights in scene order and the camera that
    Camera::UpdateOrientationVectors derives; materials and meshes come out in order of first use, so the check
    is what the arrays render to: the checker's radiance and counters from the flattened arrays equal those from
    the original arrays and the reference's own Li, bit for bit."""
    rotation = np.array([0.1, -0.3, 0.0], np.float32)
    scene = scenes[name].with_image(20, 16, 20 / 16)
    view, up, right = host.camera_vectors(rotation)
    scene.camera = dict(scene.camera, view_dir=view, up_dir=up, right_dir=right)
    desc = scene.to_desc()
    rs = ref.RefScene(desc)
    flat = rs.flatten(rotation)
    assert flat.nodes.tobytes() == scene.nodes.tobytes()
    assert np.array_equal(flat.prims["type"], scene.prims["type"])
    assert len(flat.triangles) == len(scene.triangles) and len(flat.spheres) == len(scene.spheres)
    assert np.array_equal(flat.lights["type"], scene.lights["type"]) and np.array_equal(flat.lights["prim"], scene.lights["prim"])
    assert flat.lights["Lemit"].tobytes() == scene.lights["Lemit"].tobytes()
    for k in ("position", "view_dir", "up_dir", "right_dir"):
        assert np.array_equal(flat.camera[k], scene.camera[k]), k
    assert (flat.image_width, flat.image_height, flat.max_depth) == (20, 16, scene.max_depth)
    # world-space corners of every triangle, in shape order
    def corners(s):
        t = s.triangles[s.prims["index"][s.prims["type"] == abi.PT_PRIM_TRIANGLE]]
        base = s.meshes["vertex_offset"][t["mesh"]]
        return np.stack([s.vertices[base + t[i]] for i in ("i0", "i1", "i2")], axis=1)
    assert corners(flat).tobytes() == corners(scene).tobytes()
    want, want_stats = checker.OracleScene(desc).li_image(3, seed=2)
    got, got_stats = checker.OracleScene(flat.to_desc()).li_image(3, seed=2)
    theirs, _ = rs.li_image(3, seed=2)
    assert got.tobytes() == want.tobytes() == theirs.tobytes()
    assert got_stats == want_stats


def test_pnm_decoding_equals_the_reference_texture_load(tmp_path):
    """Binary PGM / PPM through the host library and the reference's Texture::Load: comments in the header, a
    maximum below 255 (samples are taken as stored), odd sizes."""
    rng = np.random.default_rng(13)
    files = []
    for n, (w, h) in enumerate(((7, 5), (1, 1), (32, 3))):
        for magic, channels in ((b"P5", 1), (b"P6", 3)):
            for maxv, comment in ((255, b""), (100, b"# made by a test\n"), (255, b"# one\n# two\n")):
                files.append(str(tmp_path / f"{n}_{magic.decode()}_{maxv}_{len(comment)}.pnm"))
                body = rng.integers(0, maxv + 1, w * h * channels).astype(np.uint8).tobytes()
                open(files[-1], "wb").write(magic + b"\n" + comment + f"{w} {h}\n{maxv}\n".encode() + body)
    for f in files:
        for flip in (True, False):
            mine, theirs = host.read_image(f, flip), ref.load_texture(f, flip=flip)
            assert mine.shape == theirs.shape and np.array_equal(mine, theirs), (f, flip)
