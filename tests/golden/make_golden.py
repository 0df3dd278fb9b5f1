"""Generates the golden fixtures in this directory from the reference itself.

Run in the build container only (needs /root/reference and oracle/_ref/libptref.so, see oracle/Makefile.ref):

    python tests/golden/make_golden.py

Every expected value below comes from the reference's own compiled code through oracle/ref.py - never from
this repository's restatement or kernels.  The scenes are flattened by this repository's host loader from the
reference's resource files (the reference's own loader needs Assimp, which cannot be built here).

Files written:
  cornell_scene.npz   resources/scenes/cornell.json, flattened (36 triangles, 2 area lights, 71 nodes)
  mixed_scene.npz     spheres + textured and untextured meshes + point and directional light; the texture is
                      resources/images/cockatoo.jpg decoded by the reference's Texture::Load
  golden_<scene>.npz  for each scene: radiance sums of Li at 48x48 (1 and 8 samples, Philox seed 7), a
                      sample-table run, closest-hit and any-hit results on seeded ray batches
  golden_leaf.npz     known-answer tables of the leaf functions incl. edge cases
  golden_bvh.npz      node arrays the reference's BVH::Build produces for the three split methods
  golden_converged.npz  64x64 mean radiance of the reference's Li over 8192 samples per pixel, both scenes
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402
from pathtracer_amd import abi, host  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
RES = "/root/reference/resources/"

MIXED_SCENE = '''{
 "OutputImageData": { "resolution": [ 48, 48 ], "filename": "mixed.png" },
 "Camera": { "position": [ 0, 0, 0 ], "rotation": [ 0, 10, 0 ], "vfov": 45, "aspectRatio": 1, "exposure": 1.0, "gamma": 2.2 },
 "BVH": { "splitMethod": "SAH" },
 "MaxDepth": 4,
 "BackgroundColor": { "color": [ 0.2, 0.3, 0.4 ] },
 "DirectionalLight": { "direction": [ 0.2, -0.5, -1 ], "Lemit": [ 2, 2, 2 ] },
 "PointLight": { "position": [ 0, 3, -8 ], "Lemit": [ 30, 20, 10 ] },
 "Texture": { "name": "cockatoo", "filename": "images/cockatoo.tga" },
 "Material": { "name": "cockatoo", "albedo": [ 1, 1, 1 ], "albedoTexture": "cockatoo" },
 "Material": { "name": "grey", "albedo": [ 0.7, 0.7, 0.7 ] },
 "Model": { "name": "plane", "filename": "models/plane.obj" },
 "Model": { "name": "cube", "filename": "models/cube.obj" },
 "ModelInstance": { "model": "plane", "material": "cockatoo", "transform": { "position": [ 2, 0, -10 ], "rotation": [ 90, 0, 0 ], "scale": [ 2, 2, 2 ] } },
 "ModelInstance": { "model": "plane", "material": "grey", "transform": { "position": [ 0, -2, -10 ], "scale": [ 8, 8, 8 ] } },
 "ModelInstance": { "model": "cube", "material": "cockatoo", "transform": { "position": [ 0.5, -1.2, -7 ], "rotation": [ 0, 30, 0 ], "scale": [ 0.6, 0.6, 0.6 ] } },
 "Sphere": { "material": "cockatoo", "position": [ -2, 0, -10 ], "rotation": [ 0, 20, 0 ], "radius": 1.5 },
 "Sphere": { "material": "grey", "position": [ 0.3, 0.8, -8 ], "radius": 0.5 }
}'''


def write_tga(path, rgba):
    h, w, _ = rgba.shape
    hdr = bytearray(18)
    hdr[2], hdr[12], hdr[13], hdr[14], hdr[15], hdr[16], hdr[17] = 2, w & 255, w >> 8, h & 255, h >> 8, 32, 0x28
    with open(path, "wb") as f:
        f.write(bytes(hdr) + rgba[:, :, [2, 1, 0, 3]].tobytes())


def make_scenes():
    cornell = host.HostScene.from_file(RES + "scenes/cornell.json", RES).to_data()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(tmp + "/res/images")
        os.symlink(RES + "models", tmp + "/res/models")
        # the reference's stb_image decodes the JPEG; rows as stored in the file (no flip), the loader flips
        write_tga(tmp + "/res/images/cockatoo.tga", ref.load_texture(RES + "images/cockatoo.jpg", flip=False))
        with open(tmp + "/mixed.json", "w") as f:
            f.write(MIXED_SCENE)
        mixed = host.HostScene.from_file(tmp + "/mixed.json", tmp + "/res/").to_data()
    return {"cornell": cornell, "mixed": mixed}


def ray_batches(scene, rng, n):
    """Primary-like, random and grazing rays around the scene's bounding box."""
    lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
    centre, extent = 0.5 * (lo + hi), 0.5 * (hi - lo)
    cam = scene.camera["position"]
    o1 = np.tile(cam, (n, 1)).astype(np.float32)
    d1 = (centre + extent * rng.uniform(-1.2, 1.2, (n, 3)) - cam).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True).astype(np.float32)
    o2 = (centre + extent * rng.uniform(-1.0, 1.0, (n, 3))).astype(np.float32)
    d2 = rng.normal(size=(n, 3)).astype(np.float32)   # not normalised, like bounce rays
    o3 = o2.copy()
    d3 = rng.normal(size=(n, 3)).astype(np.float32)
    d3[:, rng.integers(0, 3)] = 0.0                     # axis-parallel components: infinite slabs
    d3[: n // 4, 1] = 0.0
    return np.concatenate([o1, o2, o3]), np.concatenate([d1, d2, d3])


def golden_for(name, scene):
    rng = np.random.default_rng(1234)
    small = scene.with_image(48, 48)
    desc = small.to_desc()
    rs = ref.RefScene(desc)
    out = {}
    for spp in (1, 8):
        img, rays = rs.li_image(spp, seed=7)
        out[f"li_sum_spp{spp}"] = img
        out[f"rays_spp{spp}"] = np.array([rays["rays_closest"], rays["rays_shadow"]], np.uint64)
    draws = small.draws_per_sample()
    table = rng.uniform(0, 1, 4099).astype(np.float32)   # prime length: exercises the modulo wrap
    img, _ = rs.li_image(2, seed=0, table=table, draws_per_sample=draws)
    out["table"], out["li_sum_table_spp2"] = table, img
    o, d = ray_batches(scene, rng, 4000)
    hit = rs.intersect(o, d)
    out["ray_o"], out["ray_d"] = o, d
    for k in ("prim", "t", "u", "v", "position", "normal", "tangent", "bitangent", "uv", "material", "nodes", "prim_tests"):
        out["hit_" + k] = hit[k]
    tmax = np.where(rng.uniform(size=len(o)) < 0.2, np.float32(3.4028235e38), rng.uniform(0, 4, len(o))).astype(np.float32)
    out["shadow_tmax"], out["shadow_occluded"] = tmax, rs.occluded(o, d, tmax)
    if len(scene.textures):
        u, v = rng.uniform(-3, 3, 2000).astype(np.float32), rng.uniform(-3, 3, 2000).astype(np.float32)
        u[:8] = [0, 1, -1, 0.5, -0.0, 2, 0.999999, -1e-9]
        out["tex_u"], out["tex_v"], out["tex_rgba"] = u, v, rs.texture_fetch(0, u, v)
    np.savez_compressed(os.path.join(HERE, f"golden_{name}.npz"), **out)
    print(name, "hit fraction", (hit["prim"] >= 0).mean(), "occluded fraction", out["shadow_occluded"].mean())


def golden_leaf():
    rng = np.random.default_rng(99)
    n = 4096
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    out = {}
    # slab test: random boxes and rays plus axis-parallel rays (1/0 = inf, 0 * inf = NaN) and flat boxes
    o = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:512, 0] = 0.0
    d[256:768, 1] = -0.0
    bmin = rng.uniform(-1.5, 0.5, (n, 3)).astype(np.float32)
    bmax = (bmin + rng.uniform(0, 1.5, (n, 3))).astype(np.float32)
    aimed = slice(768, 3072)                           # rays aimed at a point of their box: mostly hits
    d[aimed] = (bmin[aimed] + (bmax - bmin)[aimed] * rng.uniform(0, 1, (2304, 3)) - o[aimed]).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = (np.float32(1.0) / d).astype(np.float32)
    bmax[1024:1536, 2] = bmin[1024:1536, 2]            # flat boxes
    o[:128, 0] = bmin[:128, 0]                         # origin on a slab plane with zero direction: NaN
    maxt = np.where(rng.uniform(size=n) < 0.3, np.float32(3.4028235e38), rng.uniform(0, 5, n)).astype(np.float32)
    out.update(aabb_o=o, aabb_inv=inv, aabb_min=bmin, aabb_max=bmax, aabb_maxt=maxt, aabb_hit=ref.ray_aabb(o, inv, bmin, bmax, maxt))
    # triangles: random, edge-on (det ~ 0), hits through vertices and edges
    v0, v1, v2 = (rng.uniform(-1, 1, (n, 3)).astype(np.float32) for _ in range(3))
    to = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    bary = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    bary[:256] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 256)]          # exactly through a vertex
    bary[256:512, 0] = 0
    bary[256:512, 1:] /= bary[256:512, 1:].sum(axis=1, keepdims=True)           # on an edge
    target = (bary[:, :1] * v0 + bary[:, 1:2] * v1 + bary[:, 2:] * v2).astype(np.float32)
    td = (target - to).astype(np.float32)
    td[512:768] = (v1 - v0)[512:768]                                            # parallel to the plane
    tmaxt = np.where(rng.uniform(size=n) < 0.5, np.float32(3.4028235e38), rng.uniform(0, 2, n)).astype(np.float32)
    hit, t, u, v = ref.ray_triangle(to, td, v0, v1, v2, tmaxt)
    out.update(tri_o=to, tri_d=td, tri_v0=v0, tri_v1=v1, tri_v2=v2, tri_maxt=tmaxt, tri_hit=hit, tri_t=t, tri_u=u, tri_v=v)
    # spheres: outside, inside (far root), tangent, behind
    c = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    r = rng.uniform(0.1, 1.5, n).astype(np.float32)
    so = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    so[:512] = (c[:512] + 0.3 * r[:512, None] * rng.uniform(-1, 1, (512, 3))).astype(np.float32)   # origin inside
    sd = rng.normal(size=(n, 3)).astype(np.float32)
    smaxt = np.where(rng.uniform(size=n) < 0.5, np.float32(3.4028235e38), rng.uniform(0, 3, n)).astype(np.float32)
    hit, t = ref.ray_sphere(so, sd, c, r, smaxt)
    out.update(sph_o=so, sph_d=sd, sph_c=c, sph_r=r, sph_maxt=smaxt, sph_hit=hit, sph_t=t)
    # sampling warps incl. the centre of the square, the diagonals and the borders
    u1, u2 = rng.uniform(0, 1, n).astype(np.float32), rng.uniform(0, 1, n).astype(np.float32)
    u1[:8] = [0.5, 0.0, 1.0, 0.25, 0.75, 0.5, 0.0, 0.99999994]
    u2[:8] = [0.5, 0.0, 1.0, 0.75, 0.25, 0.0, 0.5, 0.99999994]
    u2[8:72] = u1[8:72]
    u2[72:136] = 1 - u1[72:136]
    s = ref.sampling(u1, u2)
    out.update(smp_u1=u1, smp_u2=u2, **{"smp_" + k: v for k, v in s.items()})
    # tonemap: Uncharted2 and Reinhard, then gamma and clamp, for several exposure / gamma pairs
    rgb = np.concatenate([rng.uniform(0, 4, (1500, 3)), rng.uniform(0, 0.01, (300, 3)), rng.uniform(5, 40, (245, 3)),
                          [[0, 0, 0]], [[1, 1, 1]], [[11.2, 11.2, 11.2]]]).astype(np.float32)
    out["tm_rgb"] = rgb
    for i, (exposure, gamma) in enumerate([(1.0, 2.2), (6.0, 1.0), (0.5, 2.4)]):
        a, b = ref.tonemap(rgb, exposure, gamma)
        out[f"tm_params{i}"] = np.array([exposure, gamma], np.float32)
        out[f"tm_uncharted2_{i}"], out[f"tm_reinhard_{i}"] = a, b
    np.savez_compressed(os.path.join(HERE, "golden_leaf.npz"), **out)
    _ = inf, nan


def golden_bvh(scenes):
    """The reference's BVH::Build over the shapes of each scene taken in their flattened order."""
    out = {}
    for name, scene in scenes.items():
        desc = scene.to_desc()
        for method, tag in ((0, "sah"), (1, "middle"), (2, "equal")):
            nodes, order = ref.bvh_build(desc, method)
            out[f"{name}_{tag}_nodes"], out[f"{name}_{tag}_order"] = nodes, order
    np.savez_compressed(os.path.join(HERE, "golden_bvh.npz"), **out)


CONVERGED_SPP, CONVERGED_SEED, CONVERGED_SIZE = 8192, 2024, 64


def golden_converged(scenes):
    """Mean radiance per pixel of the reference's Li over many samples: what a render has to converge to."""
    out = {"spp": np.array([CONVERGED_SPP], np.int64), "seed": np.array([CONVERGED_SEED], np.int64)}
    for name, scene in scenes.items():
        rs = ref.RefScene(scene.with_image(CONVERGED_SIZE, CONVERGED_SIZE).to_desc())
        img, _ = rs.li_image(CONVERGED_SPP, seed=CONVERGED_SEED)
        out[f"{name}_mean"] = (img / np.float32(CONVERGED_SPP)).astype(np.float32)
        print(name, "converged mean per channel", out[f"{name}_mean"].reshape(-1, 3).mean(axis=0))
    np.savez_compressed(os.path.join(HERE, "golden_converged.npz"), **out)


def golden_images():
    """Picture files of every kind the host library reads, with the reference's decode of each (Texture::Load for
    8-bit texels, the Skybox::Load calls for cube map faces), and the files the reference's Image::Save writes
    for one float picture.  File bytes are stored, so the test needs neither Pillow nor the reference."""
    import tempfile
    import PIL.Image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hdr_writer import to_rgbe, write_hdr
    from png_writer import write_png
    rng = np.random.default_rng(77)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        yy, xx = np.mgrid[0:23, 0:37]
        picture = np.stack([128 + 120 * np.sin(xx / 5.0) * np.cos(yy / 4.0), (xx * 7 + yy * 3) % 256, rng.integers(0, 256, (23, 37))], axis=-1).astype(np.uint8)
        img = PIL.Image.fromarray(picture)
        files = {}

        def path(name):
            files[name] = os.path.join(tmp, name)
            return files[name]

        img.save(path("baseline_420.jpg"), quality=80, subsampling=2)
        img.save(path("baseline_422_restart.jpg"), quality=60, subsampling=1, restart_marker_blocks=2)
        img.save(path("progressive_420.jpg"), quality=75, subsampling=2, progressive=True)
        img.save(path("progressive_444.jpg"), quality=95, subsampling=0, progressive=True)
        img.convert("L").save(path("grey_progressive.jpg"), quality=70, progressive=True)
        img.save(path("deflated.png"), optimize=True)
        img.convert("P").save(path("palette.png"))
        write_png(path("rgba16_interlaced.png"), rng.integers(0, 1 << 16, (11, 13, 4)), 6, 16, interlace=True)
        write_png(path("grey2_key.png"), rng.integers(0, 4, (9, 10, 1)), 0, 2, trns=(2).to_bytes(2, "big"))
        img.save(path("rle.tga"), compression="tga_rle")
        img.convert("P").save(path("colour_mapped.tga"))
        img.convert("LA").save(path("grey_alpha.tga"))
        img.convert("P").save(path("palette.bmp"))
        img.convert("1").save(path("one_bit.bmp"))
        img.save(path("rgb.bmp"))
        radiance = np.exp(rng.uniform(-8, 8, (9, 20, 1))) * rng.uniform(0, 1, (9, 20, 3))
        radiance[4:, :10] = radiance[4, 0]
        write_hdr(path("run_length.hdr"), to_rgbe(radiance), True)
        write_hdr(path("flat.hdr"), to_rgbe(radiance[:, :7]), False, "#?RGBE")
        for name, f in files.items():
            out[f"file_{name}"] = np.frombuffer(open(f, "rb").read(), np.uint8)
            if name.endswith(".hdr"):
                out[f"faces_{name}"] = ref.load_skybox_face(f, True)
            else:
                out[f"texels_{name}"] = ref.load_texture(f, True)
                out[f"faces_{name}"] = ref.load_skybox_face(f, False)
        shaded = np.stack([0.5 + 0.6 * np.sin(xx / 6.0 + yy / 9.0), (xx / 37.0) * (yy / 23.0), 0.3 + 0.2 * np.cos(yy / 2.0)], axis=-1)
        shaded = (shaded + rng.normal(0, 0.03, shaded.shape)).astype(np.float32)
        shaded[5:12, 8:30] = (0.9, 0.1, 0.1)
        out["rendered"] = shaded
        for ext in ("png", "tga", "bmp", "jpg"):
            f = os.path.join(tmp, "saved." + ext)
            ref.save_image(f, shaded)
            out[f"saved_{ext}"] = np.frombuffer(open(f, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "golden_images.npz"), **out)


def main():
    if not ref.available():
        sys.exit("oracle/_ref/libptref.so is missing: run `make ref` in the container that holds /root/reference")
    scenes = make_scenes()
    if "--only-converged" in sys.argv:   # the other files stay as they are
        golden_converged(scenes)
        return
    if "--only-images" in sys.argv:
        golden_images()
        return
    for name, scene in scenes.items():
        scene.save(os.path.join(HERE, f"{name}_scene.npz"))
        golden_for(name, scene)
    golden_leaf()
    golden_bvh(scenes)
    golden_converged(scenes)
    golden_images()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
