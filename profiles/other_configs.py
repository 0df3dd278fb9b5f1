"""Throughput of the configurations other than the benchmark's on one GPU: Cornell box 1024^2, the 871 k
triangle knot at 4K with primary + shadow rays only, the 20 M triangle soup at 4K.  Prints one JSON line per
configuration (rays, samples, milliseconds inside pt_render, rates and per-ray traversal counters).

usage, from the repository root on the GPU box:  python3 profiles/other_configs.py [--soup-triangles N]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pathtracer_amd import api, host  # noqa: E402
from pathtracer_amd.scene import SceneData  # noqa: E402


def measure(name, desc, spp, repeats=3, mode="ordered", flags=0):
    ordered, wide = mode != "reference", mode == "wide"
    desc.flags |= flags
    t_upload = time.time()
    with api.DeviceScene(desc) as dev:
        t_upload = time.time() - t_upload
        dev.render(min(spp, 2), seed=1, ordered=ordered, wide=wide, wide_shadows=wide)  # warm-up: allocations, code upload
        dev.readback()
        best = None
        for r in range(repeats):
            dev.render(spp, seed=r, ordered=ordered, wide=wide, wide_shadows=wide, profile=True)
            _, _, st = dev.readback()
            if best is None or st.render_ms < best.render_ms:
                best = st
        rays = best.rays_closest + best.rays_shadow
        print(json.dumps({
            "config": name, "scene_flags": int(desc.flags), "traversal": mode, "width": desc.image_width, "height": desc.image_height, "depth": desc.max_depth, "spp": spp,
            "triangles": int(desc.num_triangles), "nodes": int(desc.num_nodes),
            "samples": int(best.samples), "rays": int(rays), "render_ms": round(best.render_ms, 3),
            "mrays_per_s": round(rays / best.render_ms * 1e-3, 1),
            "msamples_per_s": round(best.samples / best.render_ms * 1e-3, 1),
            "extend_ms": round(best.kernel_ms[1], 3), "shade_ms": round(best.kernel_ms[2], 3), "connect_ms": round(best.kernel_ms[3], 3),
            "rays_redone": int(best.rays_redone), "shadow_rays": int(best.rays_shadow),
            "nodes_per_shadow_ray": round(best.nodes_shadow / max(1, best.rays_shadow), 2),
            "tests_per_shadow_ray": round((best.tri_tests_shadow + best.sphere_tests_shadow) / max(1, best.rays_shadow), 3), "upload_s": round(t_upload, 3),
            "nodes_per_ray": round((best.nodes_closest + best.nodes_shadow) / rays, 2),
            "prim_tests_per_ray": round((best.tri_tests_closest + best.tri_tests_shadow + best.sphere_tests_closest
                                         + best.sphere_tests_shadow) / rays, 3),
        }), flush=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--soup-triangles", type=int, default=20_000_000)
    p.add_argument("--traversal", choices=("wide", "ordered", "reference", "both", "all"), default="all")
    p.add_argument("--scene-flags", type=int, default=0, help="PT_SCENE_* measurement switches for every upload")
    a = p.parse_args()
    modes = {"wide": ["wide"], "ordered": ["ordered"], "reference": ["reference"], "both": ["reference", "ordered"],
             "all": ["reference", "ordered", "wide"]}[a.traversal]

    cornell = SceneData.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz")).with_image(1024, 1024)
    for m in modes:
        measure("cornell 1024^2 (configs[1])", cornell.to_desc(), 256, mode=m, flags=a.scene_flags)

    t = time.time()
    with host.HostScene.knot(seed=2, triangles=871414) as hs:
        hs.set_image(3840, 2160, 3840 / 2160)
        hs.set_max_depth(1)
        print(f"knot scene: {time.time() - t:.1f} s", file=sys.stderr)
        for m in modes:
            measure("knot 871k 4K primary + shadow (configs[3])", hs.desc(), 16, mode=m, flags=a.scene_flags)

    t = time.time()
    with host.HostScene.soup(seed=3, triangles=a.soup_triangles, edge=0.01, skybox_size=1024) as hs:
        print(f"soup scene: {time.time() - t:.1f} s", file=sys.stderr)
        for m in modes:
            measure("soup 4K depth 5 (configs[4])", hs.desc(), 8, mode=m, flags=a.scene_flags)


if __name__ == "__main__":
    main()
