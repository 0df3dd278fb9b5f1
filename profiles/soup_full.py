"""The 20 M triangle soup at its full size on one GPU: 3840x2160, depth 5, 4096 samples per pixel (3.4e10 samples),
rendered as 16 sample ranges of 256 that accumulate into one film (a line of progress per range).  Prints one JSON
line.  usage, from the repository root on the GPU box:  python3 profiles/soup_full.py [--spp N] [--ranges K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from pathtracer_amd import api, host  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--spp", type=int, default=4096)
p.add_argument("--ranges", type=int, default=16)
p.add_argument("--traversal", choices=("ordered", "reference"), default="ordered")
args = p.parse_args()

t0 = time.time()
with host.HostScene.soup(seed=3, triangles=20_000_000, edge=0.01, skybox_size=1024) as hs:
    hs.set_image(3840, 2160, 3840 / 2160)
    hs.set_max_depth(5)
    desc = hs.desc()
    print(f"scene built in {time.time() - t0:.1f} s", flush=True)
    t_up = time.time()
    with api.DeviceScene(desc) as dev:
        print(f"uploaded in {time.time() - t_up:.1f} s", flush=True)
        step = args.spp // args.ranges
        t1 = time.time()
        for k in range(args.ranges):
            dev.render(args.spp, k * step, (k + 1) * step, seed=3, accumulate=k > 0, ordered=args.traversal == "ordered")
            print(f"samples {(k + 1) * step} of {args.spp} per pixel, {time.time() - t1:.1f} s", flush=True)
        lin, _, st = dev.readback()
        rays = st.rays_closest + st.rays_shadow
        print(json.dumps({"config": "soup 20M triangles 3840x2160 depth 5", "traversal": args.traversal, "wall_s_from_start": round(time.time() - t0, 1), "spp": args.spp, "samples": int(st.samples), "rays": int(rays),
                          "render_ms": round(st.render_ms, 1), "mrays_per_s": round(rays / st.render_ms * 1e-3, 1),
                          "msamples_per_s": round(st.samples / st.render_ms * 1e-3, 1), "finite": bool(np.isfinite(lin).all()),
                          "mean_radiance": [float(x) for x in lin.reshape(-1, 3).mean(axis=0)]}), flush=True)
