"""Node visits of closest-hit rays over the wide layout (camera rays of the benchmark scene, then scattered rays from their hit
points), through the batch query.  With a library whose TraceClosestWide traverses a second time starting from the distance
it found (an experiment build, PTAMD_LIB), the counts are those of the second traversal: what a known hit distance saves."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from pathtracer_amd import abi, api, host
hs = host.HostScene.atrium(seed=1, triangles=262267, texture_size=64)
hs.set_image(1920, 1080, 1920 / 1080)
desc = hs.desc()
cam = desc.camera
pos = np.array(list(cam.position), np.float32); view = np.array(list(cam.view_dir), np.float32)
up = np.array(list(cam.up_dir), np.float32); right = np.array(list(cam.right_dir), np.float32)
hh = np.tan(cam.vfov / 2); hw = hh * cam.aspect_ratio
rng = np.random.default_rng(1)
n = 400000
x = rng.uniform(-1, 1, n).astype(np.float32)[:, None]; y = rng.uniform(-1, 1, n).astype(np.float32)[:, None]
d = view[None, :] + right[None, :] * (x * hw) + up[None, :] * (y * hh)
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.repeat(pos[None, :], n, 0)
with api.DeviceScene(desc) as dev:
    r = dev.trace_closest(o, d.astype(np.float32), traversal=abi.PT_TRAVERSAL_WIDE)
    hit = r["prim"] >= 0
    print(os.environ.get("PTAMD_LIB", "default"), "camera rays: hits", hit.mean(), "nodes per ray", r["nodes"][hit].mean(), "prim tests", r["prim_tests"][hit].mean())
    # second bounce: from the hit points into random directions
    p2 = o[hit] + d[hit] * r["t"][hit, None] * 0.999
    d2 = rng.normal(size=p2.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    r2 = dev.trace_closest(p2.astype(np.float32), d2, traversal=abi.PT_TRAVERSAL_WIDE)
    h2 = r2["prim"] >= 0
    print(os.environ.get("PTAMD_LIB", "default"), "scattered rays: hits", h2.mean(), "nodes per ray", r2["nodes"][h2].mean(), "prim tests", r2["prim_tests"][h2].mean())
