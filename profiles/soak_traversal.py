"""Soak of the front-to-back and wide traversal modes against the reference's order, all on the GPU: random scenes of
random kinds - triangle soups over many scales, clusters, grids of coplanar quads with shared edges, slivers, stacks of
coincident triangles, geometry far from the origin or tiny, with and without spheres - and rays of random kinds
(from outside, from inside, along axes with zero direction components, aimed at vertices and edges, grazing along
triangle planes, from far away).  Closest hits: primitive, t, u, v bit-equal in all three modes, but for one class that is
counted apart (see main): rays whose REFERENCE answer is a hit in front of its own primitive's bounding box; any-hit answers
equal for random tMax.  The reference-order mode itself is pinned to the CPU checker, and that to the reference's code, by the
test suites; this script widens the input space of the step between the modes.

    python3 profiles/soak_traversal.py [--cases 300] [--seed 0] [--rays 60000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pathtracer_amd import abi, api, host  # noqa: E402
from geometry_cases import triangle_scene  # noqa: E402


def random_triangles(rng):
    n = int(rng.choice([rng.integers(1, 30), rng.integers(30, 800), rng.integers(800, 20000)], p=[0.25, 0.5, 0.25]))
    kind = int(rng.integers(0, 8))
    if kind == 0:      # soup, edge length over orders of magnitude
        c = rng.uniform(-1, 1, (n, 1, 3))
        tris = c + rng.normal(size=(n, 3, 3)) * 10.0 ** rng.uniform(-4, 0, (n, 1, 1))
    elif kind == 1:    # clusters
        k = int(rng.integers(1, 6))
        centres = rng.uniform(-10, 10, (k, 3))
        c = centres[rng.integers(0, k, n)][:, None, :]
        tris = c + rng.normal(size=(n, 3, 3)) * rng.choice([1e-3, 0.05, 0.5])
    elif kind == 2:    # grid of coplanar quads: shared edges and vertices, axis-aligned boxes of zero thickness
        g = max(1, int(np.sqrt(n / 2)))
        xs, ys = np.meshgrid(np.arange(g, dtype=np.float64), np.arange(g, dtype=np.float64))
        p00 = np.stack([xs, ys, np.zeros_like(xs)], -1).reshape(-1, 3)
        a = np.stack([p00, p00 + [1, 0, 0], p00 + [1, 1, 0]], 1)
        b = np.stack([p00, p00 + [1, 1, 0], p00 + [0, 1, 0]], 1)
        tris = np.concatenate([a, b])
        axis = int(rng.integers(0, 3))
        tris = np.roll(tris, axis, axis=2)
    elif kind == 3:    # slivers
        c = rng.uniform(-1, 1, (n, 1, 3))
        d = rng.normal(size=(n, 1, 3))
        tris = c + d * rng.uniform(-1, 1, (n, 3, 1)) + rng.normal(size=(n, 3, 3)) * 1e-5
    elif kind == 4:    # stacks of coincident and nearly coincident triangles
        k = max(1, n // 8)
        base = rng.uniform(-1, 1, (k, 3, 3))
        pick = rng.integers(0, k, n)
        tris = base[pick] + rng.choice([0.0, 1e-7, 1e-4], (n, 1, 1)) * rng.normal(size=(n, 1, 3))
    elif kind == 5:    # far from the origin
        tris = rng.uniform(-1, 1, (n, 3, 3)) * 0.3 + rng.uniform(-1, 1, (n, 1, 3)) + rng.choice([1e3, 1e5, -3e4])
    elif kind == 6:    # tiny
        tris = (rng.uniform(-1, 1, (n, 1, 3)) + rng.normal(size=(n, 3, 3)) * 0.05) * 1e-4
    else:              # a few huge triangles among many small ones
        c = rng.uniform(-1, 1, (n, 1, 3))
        tris = c + rng.normal(size=(n, 3, 3)) * 0.02
        big = rng.integers(0, n, max(1, n // 50))
        tris[big] = rng.uniform(-30, 30, (len(big), 3, 3))
    return np.ascontiguousarray(tris, np.float32), kind


def add_spheres(scene, rng):
    """A few spheres as extra primitives: boxes centre +- radius, the tree rebuilt over triangles and spheres"""
    k = int(rng.integers(1, 6))
    lo, hi = scene.vertices.min(axis=0), scene.vertices.max(axis=0)
    ext = float(max((hi - lo).max(), 1e-6))
    sph = np.zeros(k, abi.SPHERE_DTYPE)
    for i in range(k):
        pos = rng.uniform(lo, hi).astype(np.float32)
        radius = np.float32(ext * 10.0 ** rng.uniform(-2.5, -0.5))
        sph["position"][i], sph["radius"][i] = pos, radius
        sph["world_to_local"][i] = host.sphere_world_to_local(pos, rng.uniform(-3, 3, 3).astype(np.float32), radius)
        sph["material"][i] = 0
    scene.spheres = sph
    tris = scene.vertices.reshape(-1, 3, 3)
    boxes = np.concatenate([np.concatenate([tris.min(axis=1), tris.max(axis=1)], axis=1),
                            np.concatenate([sph["position"] - sph["radius"][:, None], sph["position"] + sph["radius"][:, None]], axis=1)]).astype(np.float32)
    nodes, order = host.build_bvh(boxes, abi.PTH_SPLIT_SAH)
    n = len(tris)
    scene.nodes = nodes
    scene.prims = np.zeros(len(order), abi.PRIM_DTYPE)
    is_sphere = order >= n
    scene.prims["type"] = np.where(is_sphere, abi.PT_PRIM_SPHERE, abi.PT_PRIM_TRIANGLE)
    scene.prims["index"] = np.where(is_sphere, order - n, order)
    return scene


def random_rays(rng, scene, n):
    lo, hi = scene.nodes["aabb_min"][0].astype(np.float64), scene.nodes["aabb_max"][0].astype(np.float64)
    centre, ext = 0.5 * (lo + hi), max(float((hi - lo).max()), 1e-9)
    verts = scene.vertices.astype(np.float64)
    parts_o, parts_d = [], []

    def add(o, d):
        parts_o.append(o)
        parts_d.append(d)

    m = n // 6
    # from outside towards the geometry
    t = verts[rng.integers(0, len(verts), m)] + rng.normal(size=(m, 3)) * ext * 0.01
    o = centre + rng.normal(size=(m, 3)) * ext * 2
    add(o, t - o)
    # from inside, any direction
    add(rng.uniform(lo, hi, (m, 3)), rng.normal(size=(m, 3)))
    # along axes: zero direction components, origins on a lattice that shares coordinates with the vertices
    o = verts[rng.integers(0, len(verts), m)] + rng.choice([0.0, 0.0, 1e-3], (m, 3)) * ext
    d = np.zeros((m, 3))
    axis = rng.integers(0, 3, m)
    d[np.arange(m), axis] = rng.choice([-1.0, 1.0], m)
    second = rng.uniform(size=m) < 0.3
    d[second, (axis[second] + 1) % 3] = rng.normal(size=int(second.sum()))
    o = o - d * ext * rng.uniform(0, 2, (m, 1))
    add(o, d)
    # exactly at vertices and edge points
    tri = verts.reshape(-1, 3, 3)
    pick = rng.integers(0, len(tri), m)
    w = rng.uniform(size=(m, 1))
    t = np.where(rng.uniform(size=(m, 1)) < 0.5, tri[pick, 0], tri[pick, 0] * w + tri[pick, 1] * (1 - w))
    o = centre + rng.normal(size=(m, 3)) * ext * 1.5
    add(o, t - o)
    # grazing along triangle planes
    e1, e2 = tri[pick, 1] - tri[pick, 0], tri[pick, 2] - tri[pick, 0]
    nrm = np.cross(e1, e2)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    inplane = e1 * rng.normal(size=(m, 1)) + e2 * rng.normal(size=(m, 1))
    inplane /= np.maximum(np.linalg.norm(inplane, axis=1, keepdims=True), 1e-300)
    d = inplane + nrm * 10.0 ** rng.uniform(-7, -2, (m, 1)) * rng.choice([-1.0, 1.0], (m, 1))
    t = tri[pick].mean(axis=1)
    add(t - d * ext * 10.0 ** rng.uniform(-1, 3, (m, 1)), d)
    # from far away
    k = n - 5 * m
    t = verts[rng.integers(0, len(verts), k)]
    o = centre + rng.normal(size=(k, 3)) * ext * 10.0 ** rng.uniform(1, 5, (k, 1))
    add(o, (t - o) * 10.0 ** rng.uniform(-3, 3, (k, 1)))
    return np.concatenate(parts_o).astype(np.float32), np.concatenate(parts_d).astype(np.float32)


K_BAND = np.float32(1.0 + 2.0 ** -12)


def leaf_box_entry(scene, prim, o, d):
    """Distance at which the reference's slab test (src/intersection_tests.cpp:91-110, single precision) has the ray enter
    the box of the leaf that holds primitive `prim`, per ray."""
    nodes = scene.nodes
    leaf_of = np.zeros(len(scene.prims), np.int64)
    leaves = np.flatnonzero(nodes["num_shapes"] > 0)
    for i in leaves:
        leaf_of[nodes["offset"][i]: nodes["offset"][i] + nodes["num_shapes"][i]] = i
    lo, hi = nodes["aabb_min"][leaf_of[prim]], nodes["aabb_max"][leaf_of[prim]]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = np.float32(1) / d
        t0, t1 = (lo - o) * inv, (hi - o) * inv
        near = np.where(inv < 0, t1, t0)
        entry = near[:, 0]
        entry = np.where(near[:, 1] > entry, near[:, 1], entry)
        entry = np.where(near[:, 2] > entry, near[:, 2], entry)
    return entry.astype(np.float32)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--cases", type=int, default=300)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--rays", type=int, default=60000)
    a = p.parse_args()
    bad, _, _ = run(a.cases, a.seed, a.rays)
    sys.exit(1 if bad else 0)


def run(cases, seed, rays_per_scene):
    """-> (failing (scene, mode) pairs, rays of the class counted apart per mode, rays traced per mode)"""
    a = argparse.Namespace(cases=cases, seed=seed, rays=rays_per_scene)
    rng = np.random.default_rng(a.seed)
    t0, rays, explained = time.time(), 0, {"ordered": 0, "wide": 0}
    kinds = np.zeros(8, np.int64)
    bad = 0
    for case in range(a.cases):
        tris, kind = random_triangles(rng)
        kinds[kind] += 1
        scene = triangle_scene(tris)
        spheres = rng.uniform() < 0.3
        if spheres:
            scene = add_spheres(scene, rng)
        o, d = random_rays(rng, scene, a.rays)
        with np.errstate(over="ignore"):
            tmax = (np.abs(rng.normal(size=len(o))) * float(np.abs(scene.nodes["aabb_max"][0] - scene.nodes["aabb_min"][0]).max()) * 2).astype(np.float32)
        tmax[rng.uniform(size=len(o)) < 0.2] = np.float32(3.4e38)
        with api.DeviceScene(scene.to_desc()) as dev:
            ref = dev.trace_closest(o, d, traversal=abi.PT_TRAVERSAL_REFERENCE)
            ref_any = dev.trace_any(o, d, tmax, traversal=abi.PT_TRAVERSAL_REFERENCE)["occluded"]
            for name, mode in (("ordered", abi.PT_TRAVERSAL_ORDERED), ("wide", abi.PT_TRAVERSAL_WIDE)):
                got = dev.trace_closest(o, d, traversal=mode)
                differ = np.zeros(len(o), bool)
                for k in ("prim", "t", "u", "v"):
                    differ |= got[k].view(np.uint32) != ref[k].view(np.uint32)
                anyd = dev.trace_any(o, d, tmax, traversal=mode)["occluded"] != ref_any
                if differ.any():
                    # The one class of rays on which an order other than the reference's can differ: the reference's own
                    # answer is a primitive whose computed t lies IN FRONT of the primitive's bounding box by more than the
                    # band (an ill-conditioned test: grazing incidence from far away, sliver triangles) - a hit that is not
                    # on the primitive.  A traversal that culls boxes by distance never tests that primitive once it holds
                    # a hit nearer than the box.
                    idx = np.flatnonzero(differ & (ref["prim"] >= 0))
                    entry = leaf_box_entry(scene, ref["prim"][idx], o[idx], d[idx])
                    in_front = entry > ref["t"][idx] * K_BAND
                    explained[name] += int(in_front.sum())
                    differ[idx[in_front]] = False
                if differ.any() or anyd.any():
                    bad += 1
                    i = int(np.flatnonzero(differ | anyd)[0])
                    print(f"case {case} (kind {kind}, {len(tris)} triangles, spheres {spheres}), {name}: {int(differ.sum())} closest hits and {int(anyd.sum())} any-hit answers "
                          f"differ; first: ray {i} o {o[i]} d {d[i]} reference prim {ref['prim'][i]} t {ref['t'][i]!r} got prim {got['prim'][i]} t {got['t'][i]!r}", flush=True)
        rays += len(o)
        if (case + 1) % 25 == 0:
            print(f"{case + 1} scenes, {rays / 1e6:.1f} M rays x 3 modes x 2 queries, {bad} failing (scene, mode) pairs, "
                  f"reference hits in front of their own box that the other order does not return: {explained}, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {a.cases} scenes (kinds {kinds.tolist()}), {rays / 1e6:.1f} M rays, seed {a.seed}: {bad} failing (scene, mode) pairs; "
          f"rays whose reference answer lies in front of its own box by more than the band and differs: {explained}")
    return bad, explained, rays


if __name__ == "__main__":
    main()
