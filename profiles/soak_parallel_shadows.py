"""Soak for the light-space shadow rays: random light directions (a share of them almost along an axis or with tiny
components), random translations and scales of the fixture scenes; the film front to back, and the film over the wide
layout (whose shadow rays run over the wide nodes in the light's frame), must equal the film in the reference's order bit
for bit.  usage: python3 profiles/soak_parallel_shadows.py [--cases N]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import abi, api
from pathtracer_amd.scene import SceneData

p = argparse.ArgumentParser()
p.add_argument("--cases", type=int, default=300)
a = p.parse_args()
rng = np.random.default_rng(20261111)
scenes = {n: SceneData.load(os.path.join(ROOT, "tests", "golden", f"{n}_scene.npz")) for n in ("cornell", "mixed")}
bad = with_wide = 0
for case in range(a.cases):
    name = ("cornell", "mixed")[case % 2]
    s = scenes[name].with_image(32, 32, 1.0)
    s.max_depth = int(rng.integers(1, 5))
    d = rng.normal(size=3)
    kind = case % 5
    if kind == 1:
        d[rng.integers(0, 3)] *= 1e-6
    elif kind == 2:
        d[rng.integers(0, 3)] = 0.0
    elif kind == 3:
        k = rng.integers(0, 3); d[:] = rng.normal(size=3) * 1e-4; d[k] = rng.choice([-1.0, 1.0])
    rec = np.zeros(1, abi.LIGHT_DTYPE)
    rec["type"], rec["n_samples"], rec["prim"] = abi.PT_LIGHT_DIRECTIONAL, 1, -1
    rec["Lemit"] = (9, 8, 7)
    rec["direction"] = (d / np.linalg.norm(d)).astype(np.float32) if case % 7 else d.astype(np.float32)
    s.lights = rec
    if name == "cornell" and case % 3 == 0:   # triangles only: move and scale the whole scene
        scale = np.float32(10.0 ** rng.uniform(-3, 3))
        t = (rng.normal(size=3) * 10.0 ** rng.uniform(-2, 4)).astype(np.float32)
        s.vertices = s.vertices * scale + t
        s.nodes = s.nodes.copy()
        s.nodes["aabb_min"] = s.nodes["aabb_min"] * scale + t
        s.nodes["aabb_max"] = s.nodes["aabb_max"] * scale + t
        # boxes recomputed so that they still contain their triangles after the rounding of the move
        tri = s.triangles
        off = s.meshes["vertex_offset"][tri["mesh"]]
        P = np.stack([s.vertices[off + tri["i0"]], s.vertices[off + tri["i1"]], s.vertices[off + tri["i2"]]], 1)
        n = s.nodes
        for i in range(len(n) - 1, -1, -1):
            if n["num_shapes"][i] > 0:
                idx = s.prims["index"][n["offset"][i]:n["offset"][i] + n["num_shapes"][i]]
                n["aabb_min"][i] = P[idx].reshape(-1, 3).min(0); n["aabb_max"][i] = P[idx].reshape(-1, 3).max(0)
            else:
                l, r = i + 1, int(n["offset"][i])
                n["aabb_min"][i] = np.minimum(n["aabb_min"][l], n["aabb_min"][r]); n["aabb_max"][i] = np.maximum(n["aabb_max"][l], n["aabb_max"][r])
        s.camera = dict(s.camera)
        s.camera["position"] = s.camera["position"] * scale + t
    with api.DeviceScene(s.to_desc()) as dev:
        dev.render(4, seed=case)
        ref = dev.download_film()
        dev.render(4, seed=case, ordered=True)
        fast = dev.download_film()
        st = dev.stats()
        dev.render(4, seed=case, wide=True, wide_shadows=True)   # shadow rays over the wide nodes in the light's frame
        wide = dev.download_film()
        have_wide = len(dev.read_layout(abi.PT_LAYOUT_LIGHT_WIDE_NODES)) > 0
    with_wide += have_wide
    for what, film in (("front to back", fast), ("wide", wide)):
        if not np.array_equal(ref.view(np.uint32), film.view(np.uint32)):
            bad += 1
            print(f"case {case} ({name}, light {rec['direction'][0].tolist()}), {what}: {(ref.view(np.uint32) != film.view(np.uint32)).any(-1).sum()} pixels differ", flush=True)
print(f"{a.cases} cases ({with_wide} with the wide nodes in the light's frame), {bad} differing films", flush=True)
sys.exit(1 if bad else 0)
