"""How much would ordering a unit's hits by primitive save the shading kernel?

ShadeKernel works on units of 256 queue entries and gathers a 128-byte attribute record (two 64-byte lines) per hit.
Entries of a unit that hit the same primitive share those lines; entries that hit different primitives cannot, however
they are ordered, because a record fills its lines.  So the most that ordering can save is the share of a unit's
fetches that are repeats of a primitive already fetched - and only if those repeats are not already served by the L1,
which holds a whole unit's records (256 x 128 B = 32 KB).  This script measures that share on the atrium, per bounce:
camera rays in the render's own order (one unit = one sample of a 16 x 16 tile), then cosine-distributed bounces off the
hit points.

usage (GPU box): python3 profiles/dev/shade_locality.py [--tiles 256]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from pathtracer_amd import abi, api, host  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--tiles", type=int, default=256)
p.add_argument("--bounces", type=int, default=4)
a = p.parse_args()

rs = np.random.RandomState(7)
with host.HostScene.atrium(seed=1, triangles=262267, texture_size=256) as hs:
    desc = hs.desc()
    cam = desc.camera
    W, H = desc.image_width, desc.image_height
    pos = np.array(cam.position[:], np.float32)
    view, up, right = (np.array(v[:], np.float32) for v in (cam.view_dir, cam.up_dir, cam.right_dir))
    half_h = np.tan(cam.vfov / 2)
    half_w = half_h * cam.aspect_ratio
    tiles_x = W // 16
    tile_ids = rs.choice(tiles_x * (H // 16), a.tiles, replace=False)
    rows = (tile_ids // tiles_x)[:, None] * 16 + np.arange(256)[None, :] // 16
    cols = (tile_ids % tiles_x)[:, None] * 16 + np.arange(256)[None, :] % 16
    px = (cols + rs.rand(*cols.shape)) / W * 2 - 1
    py = 1 - (rows + rs.rand(*rows.shape)) / H * 2
    d = view[None, None, :] + px[..., None] * half_w * right + py[..., None] * half_h * up
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32).reshape(-1, 3)
    o = np.broadcast_to(pos, d.shape).copy()
    unit = np.repeat(np.arange(a.tiles), 256)  # unit of each live path; a unit keeps its survivors, like the queues do
    out = []
    with api.DeviceScene(desc) as dev:
        for bounce in range(a.bounces + 1):
            res = dev.trace_closest(o, d, traversal=abi.PT_TRAVERSAL_WIDE)
            t, prim = res["t"], res["prim"]
            hit = prim >= 0
            distinct = repeats = 0
            for u in range(a.tiles):
                pr = prim[(unit == u) & hit]
                distinct += len(np.unique(pr))
                repeats += len(pr) - len(np.unique(pr))
            out.append({"bounce": bounce, "paths": int(len(prim)), "hits": int(hit.sum()), "distinct_primitives_per_unit": round(distinct / a.tiles, 1),
                        "repeat_share_of_fetches": round(repeats / max(1, int(hit.sum())), 4)})
            # next bounce: a direction on the hemisphere about the side of the surface the ray came from (the true normal is
            # not needed for a statistic about where diffuse rays land)
            o = (o + d * t[:, None])[hit]
            back = -d[hit]
            v = rs.normal(size=back.shape).astype(np.float32)
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            nd = back + v  # cosine-like lobe about the incoming direction's reverse
            nd /= np.maximum(np.linalg.norm(nd, axis=1, keepdims=True), 1e-6)
            o = (o + 1e-3 * back).astype(np.float32)
            d = nd.astype(np.float32)
            unit = unit[hit]
    for r in out:
        print(json.dumps(r))
