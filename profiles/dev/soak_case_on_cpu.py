"""One case of profiles/soak_traversal.py replayed on the CPU checker: the same scene and rays (the generator is replayed
up to the case), reference order against the checker's front-to-back mode.  usage: soak_case_on_cpu.py SEED CASE [RAYS]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import soak_traversal as soak  # noqa: E402
from geometry_cases import triangle_scene  # noqa: E402
from oracle import checker  # noqa: E402
from pathtracer_amd import abi  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
rays = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
rng = np.random.default_rng(seed)
for case in range(want + 1):
    tris, kind = soak.random_triangles(rng)
    scene = triangle_scene(tris)
    spheres = rng.uniform() < 0.3
    if spheres:
        scene = soak.add_spheres(scene, rng)
    o, d = soak.random_rays(rng, scene, rays)
    with np.errstate(over="ignore"):
        tmax = (np.abs(rng.normal(size=len(o))) * float(np.abs(scene.nodes["aabb_max"][0] - scene.nodes["aabb_min"][0]).max()) * 2).astype(np.float32)
    tmax[rng.uniform(size=len(o)) < 0.2] = np.float32(3.4e38)
desc = scene.to_desc()
ref, ordered = checker.OracleScene(desc), checker.OracleScene(desc).set_traversal(abi.PT_TRAVERSAL_ORDERED)
a, b = ref.intersect(o, d), ordered.intersect(o, d)
differ = np.zeros(len(o), bool)
for k in ("prim", "t", "u", "v"):
    differ |= a[k].view(np.uint32) != b[k].view(np.uint32)
print(f"case {want}: kind {kind}, {len(tris)} triangles, spheres {spheres}, root box {scene.nodes['aabb_min'][0]} .. {scene.nodes['aabb_max'][0]}: {int(differ.sum())} rays differ on the CPU")
for i in np.flatnonzero(differ)[:5]:
    print(f"  ray {i}: o {o[i]} d {d[i]}: reference prim {a['prim'][i]} t {a['t'][i]!r}, front to back prim {b['prim'][i]} t {b['t'][i]!r}")
np.savez("/tmp/soak_case.npz", tris=tris, o=o, d=d)
