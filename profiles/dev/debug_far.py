import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import checker
from pathtracer_amd import abi, api
from pathtracer_amd.scene import SceneData
scene = SceneData.load(os.path.join(ROOT, "tests", "golden", "mixed_scene.npz")).with_image(48, 48)
desc = scene.to_desc()
rng = np.random.default_rng(9)
n = 120_000
lo, hi = scene.nodes["aabb_min"][0], scene.nodes["aabb_max"][0]
o = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
scale = np.float32(10.0) ** rng.integers(-38, 31, (n, 3)).astype(np.float32)
d = (d * np.where(rng.integers(0, 3, (n, 3)) == 0, scale, np.float32(1.0))).astype(np.float32)
o = (o + rng.normal(size=(n, 3)).astype(np.float32) * np.float32(1e4)).astype(np.float32)
centre = ((lo + hi) / 2).astype(np.float32)
d = (centre - o + rng.normal(size=(n, 3)).astype(np.float32) * (hi - lo).astype(np.float32)).astype(np.float32)
orc = checker.OracleScene(desc)
want = orc.intersect(o, d)
with api.DeviceScene(desc) as dev:
    for mode, name in ((abi.PT_TRAVERSAL_ORDERED, "ordered"), (abi.PT_TRAVERSAL_WIDE, "wide")):
        got = dev.trace_closest(o, d, mode)
        bad = (got["prim"] != want["prim"]) | (got["t"].view(np.uint32) != want["t"].view(np.uint32))
        print(name, int(bad.sum()), "mismatches")
        for i in np.argwhere(bad)[:12, 0]:
            pt = scene.prims["type"]
            print("  ray", i, "want prim", want["prim"][i], "type", pt[want["prim"][i]] if want["prim"][i] >= 0 else -1, "t", want["t"][i],
                  "got prim", got["prim"][i], "type", pt[got["prim"][i]] if got["prim"][i] >= 0 else -1, "t", got["t"][i])
