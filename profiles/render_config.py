"""Renders one of the procedural configurations once, for the profiler.
usage: python3 profiles/render_config.py soup|knot|atrium|cornell [--traversal wide|ordered|reference] [--triangles N] [--spp N] [--width W --height H] [--depth D]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pathtracer_amd import api, host  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("scene", choices=["soup", "knot", "atrium", "cornell"])
p.add_argument("--triangles", type=int, default=0)
p.add_argument("--spp", type=int, default=4)
p.add_argument("--width", type=int, default=0)
p.add_argument("--height", type=int, default=0)
p.add_argument("--depth", type=int, default=-1)
p.add_argument("--traversal", choices=("wide", "ordered", "reference"), default="ordered")
p.add_argument("--host-layouts", action="store_true", help="PT_SCENE_HOST_LAYOUTS: records written on the host")
a = p.parse_args()

if a.scene == "cornell":
    # configs[1]: the Cornell box of the golden fixtures at 1024 x 1024 (small tree: the traversal flags change nothing but the layout read)
    from pathtracer_amd.scene import SceneData  # noqa: E402
    sd = SceneData.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz")).with_image(a.width or 1024, a.height or 1024)
    with api.DeviceScene(sd.to_desc()) as dev:
        dev.render(a.spp, seed=0, ordered=a.traversal != "reference", wide=a.traversal == "wide", wide_shadows=a.traversal == "wide")
        _, _, st = dev.readback()
        rays = st.rays_closest + st.rays_shadow
        print(json.dumps({"scene": a.scene, "spp": a.spp, "rays": int(rays), "render_ms": st.render_ms, "mrays_per_s": rays / st.render_ms * 1e-3}))
    sys.exit(0)
if a.scene == "soup":
    hs = host.HostScene.soup(seed=3, triangles=a.triangles or 20_000_000, edge=0.01, skybox_size=1024)
elif a.scene == "knot":
    hs = host.HostScene.knot(seed=2, triangles=a.triangles or 871414)
else:
    hs = host.HostScene.atrium(seed=1, triangles=a.triangles or 262267, texture_size=1024)
with hs:
    if a.width and a.height:
        hs.set_image(a.width, a.height, a.width / a.height)
    if a.depth >= 0:
        hs.set_max_depth(a.depth)
    desc = hs.desc()
    if a.host_layouts:
        desc.flags |= 32  # PT_SCENE_HOST_LAYOUTS
    with api.DeviceScene(desc) as dev:
        dev.render(a.spp, seed=0, ordered=a.traversal != "reference", wide=a.traversal == "wide", wide_shadows=a.traversal == "wide")
        _, _, st = dev.readback()
        rays = st.rays_closest + st.rays_shadow
        print(json.dumps({"scene": a.scene, "spp": a.spp, "rays": int(rays), "render_ms": st.render_ms,
                          "mrays_per_s": rays / st.render_ms * 1e-3,
                          "nodes_closest": int(st.nodes_closest), "nodes_shadow": int(st.nodes_shadow),
                          "tri_tests_closest": int(st.tri_tests_closest), "tri_tests_shadow": int(st.tri_tests_shadow),
                          "rays_closest": int(st.rays_closest), "rays_shadow": int(st.rays_shadow)}))
