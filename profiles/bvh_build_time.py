"""Time of BVH::Build (SAH) on the GPU (pt_build_bvh) beside the host builder (pth_build_bvh, threaded) on the boxes
of a triangle soup like the 20 M triangle configuration's: centres uniform in [-1, 1]^3, half extents up to `edge`.
Checks on the way that the two trees are the same.  One JSON line per size.

    python3 profiles/bvh_build_time.py [--sizes 262267,871414,2000000,20000000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import abi, api, host  # noqa: E402


def soup_boxes(n, edge, seed):
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    half = rng.uniform(0, edge, (n, 3)).astype(np.float32)
    return np.concatenate([centre - half, centre + half], axis=1).astype(np.float32)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--sizes", default="262267,871414,2000000,20000000")
    p.add_argument("--edge", type=float, default=0.01)
    p.add_argument("--repeats", type=int, default=3)
    a = p.parse_args()
    api.build_bvh(soup_boxes(1000, a.edge, 0))   # loads the library, creates the context
    for n in (int(s) for s in a.sizes.split(",")):
        boxes = soup_boxes(n, a.edge, 3)
        t = time.perf_counter()
        want_nodes, want_order = host.build_bvh(boxes, abi.PTH_SPLIT_SAH)
        host_s = time.perf_counter() - t
        walls, devs = [], []
        for _ in range(a.repeats):
            t = time.perf_counter()
            nodes, order, ms = api.build_bvh(boxes, return_ms=True)
            walls.append(time.perf_counter() - t)
            devs.append(ms)
        same = bool((order == want_order).all() and (nodes["offset"] == want_nodes["offset"]).all()
                    and (nodes["axis"] == want_nodes["axis"]).all() and (nodes["num_shapes"] == want_nodes["num_shapes"]).all()
                    and (nodes["aabb_min"] == want_nodes["aabb_min"]).all() and (nodes["aabb_max"] == want_nodes["aabb_max"]).all())
        leaves = nodes["num_shapes"] > 0
        print(json.dumps({"shapes": n, "nodes": len(nodes), "same_tree": same, "host_builder_s": round(host_s, 3),
                          "host_threads": min(16, os.cpu_count() or 1),
                          "gpu_build_ms": round(min(devs), 2), "gpu_call_s_with_copies": round(min(walls), 3),
                          "mshapes_per_s_gpu": round(n / min(devs) / 1e3, 1), "leaves": int(leaves.sum())}), flush=True)
        assert same


if __name__ == "__main__":
    main()
