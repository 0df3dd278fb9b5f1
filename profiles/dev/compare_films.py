"""Films of the three traversal modes for one pass of the benchmark scene at full size, compared bit for bit.
usage (GPU box): python3 profiles/dev/compare_films.py [--spp 128]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import abi, api, host  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--spp", type=int, default=128)
p.add_argument("--seeds", type=int, default=2)
a = p.parse_args()
with host.HostScene.atrium(seed=1, triangles=262267, texture_size=1024) as hs:
    hs.set_image(1920, 1080, 1920 / 1080)
    with api.DeviceScene(hs.desc()) as dev:
        for seed in range(a.seeds):
            films = {}
            for name, kw in (("reference", {}), ("ordered", {"ordered": True}), ("wide", {"wide": True, "wide_shadows": True})):
                dev.clear_film()
                dev.render(a.spp, seed=seed, **kw)
                film, _, st = dev.readback(abi.PT_TONEMAP_NONE, tonemapped=False)
                films[name] = (film.copy(), st)
            ref, st = films["reference"]
            for name in ("ordered", "wide"):
                differing = int((films[name][0].view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
                print(f"seed {seed}, {a.spp} spp at 1920x1080 ({st.rays_closest / 1e6:.0f} M closest-hit rays, {st.rays_shadow / 1e6:.0f} M shadow rays): "
                      f"{name} differs from the reference order in {differing} pixels; rays traced again: {films[name][1].rays_redone}", flush=True)
