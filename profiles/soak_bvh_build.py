"""Soak of the GPU tree builder (pt_build_bvh) against the host builder (pth_build_bvh): random box sets of random
sizes and kinds - uniform, clustered, on grids (tied and coincident centroids), flat, spanning many orders of
magnitude, with repeated shapes - shape order and node words bit-equal, boxes equal as floats.

    python3 profiles/soak_bvh_build.py [--cases 2000] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import abi, api, host  # noqa: E402


def random_boxes(rng):
    n = int(rng.choice([rng.integers(1, 40), rng.integers(1, 600), rng.integers(1, 6000), rng.integers(1, 60000)], p=[0.3, 0.4, 0.25, 0.05]))
    kind = rng.integers(0, 7)
    if kind == 0:      # uniform soup
        lo = rng.uniform(-1, 1, (n, 3))
        ext = rng.uniform(0, rng.choice([0.001, 0.05, 0.5]), (n, 3))
    elif kind == 1:    # clusters
        k = int(rng.integers(1, 8))
        centres = rng.uniform(-10, 10, (k, 3))
        lo = centres[rng.integers(0, k, n)] + rng.normal(0, rng.choice([1e-4, 0.01, 1.0]), (n, 3))
        ext = np.abs(rng.normal(0, 0.01, (n, 3)))
    elif kind == 2:    # grid: ties along every axis, coincident shapes if the grid is small
        g = int(rng.integers(1, 12))
        lo = rng.integers(0, g, (n, 3)).astype(np.float64)
        ext = rng.integers(0, 3, (n, 3)) * 0.5
    elif kind == 3:    # flat along one or two axes
        lo = rng.uniform(-1, 1, (n, 3))
        ext = rng.uniform(0, 0.1, (n, 3))
        for axis in rng.permutation(3)[:rng.integers(1, 3)]:
            lo[:, axis] = rng.uniform(-1, 1)
            ext[:, axis] = 0
    elif kind == 4:    # many orders of magnitude
        lo = rng.normal(0, 1, (n, 3)) * 10.0 ** rng.uniform(-20, 20, (n, 1))
        ext = np.abs(lo) * rng.uniform(0, 0.5, (n, 3))
    elif kind == 5:    # few distinct shapes, repeated
        k = int(rng.integers(1, 6))
        base_lo, base_ext = rng.uniform(-1, 1, (k, 3)), rng.uniform(0, 0.3, (k, 3))
        pick = rng.integers(0, k, n)
        lo, ext = base_lo[pick], base_ext[pick]
    else:              # far from the origin: centroids collapse onto few floats
        lo = rng.uniform(-1, 1, (n, 3)) + rng.choice([1e5, 1e7, -3e6])
        ext = rng.uniform(0, 0.2, (n, 3))
    lo = lo.astype(np.float32)
    hi = (lo + ext.astype(np.float32)).astype(np.float32)
    boxes = np.concatenate([lo, hi], axis=1)
    if rng.random() < 0.2:
        boxes[rng.random(boxes.shape) < 0.05] *= np.float32(-0.0)   # zeros of either sign
        boxes[:, :3], boxes[:, 3:] = np.minimum(boxes[:, :3], boxes[:, 3:]), np.maximum(boxes[:, :3], boxes[:, 3:])
    return kind, boxes


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--cases", type=int, default=2000)
    p.add_argument("--seed", type=int, default=0)
    a = p.parse_args()
    rng = np.random.default_rng(a.seed)
    t0, shapes, by_kind = time.time(), 0, [0] * 7
    for case in range(a.cases):
        kind, boxes = random_boxes(rng)
        want_nodes, want_order = host.build_bvh(boxes, abi.PTH_SPLIT_SAH)
        nodes, order = api.build_bvh(boxes)
        ok = (np.array_equal(order, want_order) and all(np.array_equal(nodes[f], want_nodes[f]) for f in ("offset", "num_shapes", "axis"))
              and (nodes["aabb_min"] == want_nodes["aabb_min"]).all() and (nodes["aabb_max"] == want_nodes["aabb_max"]).all())
        if not ok:
            np.save("gpurun_out/soak_bvh_failing_boxes.npy", boxes)
            print(f"case {case} (kind {kind}, {len(boxes)} shapes): trees differ; boxes saved", flush=True)
            sys.exit(1)
        shapes += len(boxes)
        by_kind[kind] += 1
        if case % 250 == 249:
            print(f"{case + 1} cases, {shapes} shapes, {time.time() - t0:.0f} s", flush=True)
    print(f"{a.cases} cases ({by_kind} per kind), {shapes} shapes in all: every tree equal to the host builder's")


if __name__ == "__main__":
    main()
