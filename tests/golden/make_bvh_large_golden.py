"""Writes tests/golden/golden_bvh_large.npz: 4,000 triangles and the node array and shape order that the reference's
own BVH::Build (SplitMethod::SAH, src/bvh.cpp:204-228) produces for them - big enough for every branch of the
builder (bucketed splits over thousands of shapes, the partitions they imply, nth_element on two to four
shapes, ties between centroids along an axis, boxes without thickness).  Runs only where /root/reference is
(through oracle/_ref); the tests read the file.

    python3 tests/golden/make_bvh_large_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from geometry_cases import triangle_scene  # noqa: E402
from oracle import ref  # noqa: E402
from pathtracer_amd import abi  # noqa: E402


def triangles(seed=17):
    rng = np.random.default_rng(seed)
    parts = []
    centre = rng.uniform(-1, 1, (2500, 1, 3))
    parts.append(centre + rng.uniform(-0.03, 0.03, (2500, 3, 3)))                      # a soup
    cells = rng.permutation(1000)[:900]                                               # no two alike: the reference never returns from
    cell = np.stack([cells % 10, cells // 10 % 10, cells // 100], axis=1)[:, None, :].astype(np.float64)   # a range of coincident centroids
    shape = np.array([[0, 0, 0], [0.5, 0, 0], [0, 0.5, 0]])
    parts.append(cell * 0.5 + shape - 6.0)                                            # a grid: centroids that tie along every axis
    flat = rng.uniform(-1, 1, (400, 1, 3)) + rng.uniform(-0.1, 0.1, (400, 3, 3))
    flat[:, :, 1] = 2.0
    parts.append(flat)                                                                # one plane: boxes without thickness
    cluster = rng.normal(0, 0.002, (200, 3, 3)) + np.array([3.0, 3.0, 3.0])
    parts.append(cluster)                                                             # a tight cluster far from the rest
    tris = np.concatenate(parts).astype(np.float32)
    return tris[rng.permutation(len(tris))]


def main():
    tris = triangles()
    scene = triangle_scene(tris)
    scene.prims["index"] = np.arange(len(tris), dtype=np.uint32)   # the reference builds over the shapes in this order
    nodes, order = ref.bvh_build(scene.to_desc(), abi.PTH_SPLIT_SAH)
    assert len(nodes) == 2 * len(tris) - 1 and (np.sort(order) == np.arange(len(tris))).all()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_bvh_large.npz")
    np.savez_compressed(out, triangles=tris, nodes=nodes, order=order)
    print(out, os.path.getsize(out) // 1024, "KiB,", len(nodes), "nodes, depth-first order of", len(order), "shapes")


if __name__ == "__main__":
    main()
