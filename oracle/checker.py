"""Binding of oracle/libptoracle.so, the CPU restatement of the reference's hot path (oracle/pt_oracle.c).

TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py,
never by the product package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libptoracle.so")
_LIB = None


class PtoStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "samples", "rays_closest", "rays_shadow", "nodes_closest", "nodes_shadow", "tri_tests_closest",
        "tri_tests_shadow", "sphere_tests_closest", "sphere_tests_shadow", "shaded_hits", "draws", "rays_redone")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class OracleError(RuntimeError):
    pass


def available():
    return os.path.exists(_PATH)


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise OracleError(f"{_PATH} is missing: run `make oracle`")
        L = C.CDLL(_PATH)
        L.pto_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise OracleError(lib().pto_last_error().decode())


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def max_threads():
    return int(lib().pto_max_threads())


def ray_aabb(o, inv_dir, bmin, bmax, max_t):
    o, inv_dir, bmin, bmax, max_t = map(_f32, (o, inv_dir, bmin, bmax, max_t))
    hit = np.zeros(len(max_t), np.uint8)
    _check(lib().pto_ray_aabb(C.c_uint64(len(max_t)), _p(o), _p(inv_dir), _p(bmin), _p(bmax), _p(max_t), _p(hit)))
    return hit


def ray_triangle(o, d, v0, v1, v2, max_t):
    o, d, v0, v1, v2, max_t = map(_f32, (o, d, v0, v1, v2, max_t))
    n = len(max_t)
    hit = np.zeros(n, np.uint8)
    t, u, v = (np.zeros(n, np.float32) for _ in range(3))
    _check(lib().pto_ray_triangle(C.c_uint64(n), _p(o), _p(d), _p(v0), _p(v1), _p(v2), _p(max_t), _p(hit), _p(t), _p(u), _p(v)))
    return hit, t, u, v


def ray_sphere(o, d, centre, radius, max_t):
    o, d, centre, radius, max_t = map(_f32, (o, d, centre, radius, max_t))
    n = len(max_t)
    hit = np.zeros(n, np.uint8)
    t = np.zeros(n, np.float32)
    _check(lib().pto_ray_sphere(C.c_uint64(n), _p(o), _p(d), _p(centre), _p(radius), _p(max_t), _p(hit), _p(t)))
    return hit, t


def sampling(u1, u2):
    u1, u2 = _f32(u1), _f32(u2)
    n = len(u1)
    cos_h, disk = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
    sphere, tri = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
    _check(lib().pto_sampling(C.c_uint64(n), _p(u1), _p(u2), _p(cos_h), _p(disk), _p(sphere), _p(tri)))
    return {"cosine_hemisphere": cos_h, "concentric_disk": disk, "uniform_sphere": sphere, "uniform_triangle": tri}


def tonemap(rgb, exposure, gamma, op=0):
    rgb = _f32(rgb)
    out = np.zeros_like(rgb)
    _check(lib().pto_tonemap(C.c_uint64(rgb.size // 3), _p(rgb), C.c_float(exposure), C.c_float(gamma), C.c_uint32(op), _p(out)))
    return out


class OracleScene:
    def __init__(self, desc):
        self._h = C.c_void_p()
        _check(lib().pto_scene_create(C.byref(desc), C.byref(self._h)))
        self.width, self.height = desc.image_width, desc.image_height

    def set_traversal(self, traversal):
        """0 = the reference's first-child-first order (default), 1 = front to back (checks the GPU's second mode)."""
        _check(lib().pto_scene_set_traversal(self._h, C.c_uint32(traversal)))
        return self

    def close(self):
        if self._h:
            lib().pto_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def intersect(self, origins, dirs):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        out = {"prim": np.zeros(n, np.int32), "t": np.zeros(n, np.float32), "u": np.zeros(n, np.float32),
               "v": np.zeros(n, np.float32), "position": np.zeros((n, 3), np.float32),
               "normal": np.zeros((n, 3), np.float32), "tangent": np.zeros((n, 3), np.float32),
               "bitangent": np.zeros((n, 3), np.float32), "uv": np.zeros((n, 2), np.float32),
               "material": np.zeros(n, np.int32), "nodes": np.zeros(n, np.uint32), "prim_tests": np.zeros(n, np.uint32)}
        _check(lib().pto_intersect(self._h, C.c_uint64(n), _p(o), _p(d), _p(out["prim"]), _p(out["t"]), _p(out["u"]),
                                   _p(out["v"]), _p(out["position"]), _p(out["normal"]), _p(out["tangent"]),
                                   _p(out["bitangent"]), _p(out["uv"]), _p(out["material"]), _p(out["nodes"]),
                                   _p(out["prim_tests"])))
        return out

    def occluded(self, origins, dirs, t_max):
        o, d, tm = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3), _f32(t_max)
        n = len(o)
        occ, nodes, tests = np.zeros(n, np.uint8), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        _check(lib().pto_occluded(self._h, C.c_uint64(n), _p(o), _p(d), _p(tm), _p(occ), _p(nodes), _p(tests)))
        return {"occluded": occ, "nodes": nodes, "prim_tests": tests}

    def texture_fetch(self, tex, u, v):
        u, v = _f32(u), _f32(v)
        out = np.zeros((len(u), 4), np.float32)
        _check(lib().pto_texture_fetch(self._h, int(tex), C.c_uint64(len(u)), _p(u), _p(v), _p(out)))
        return out

    def skybox_fetch(self, dirs):
        d = _f32(dirs).reshape(-1, 3)
        out = np.zeros((len(d), 4), np.float32)
        _check(lib().pto_skybox_fetch(self._h, C.c_uint64(len(d)), _p(d), _p(out)))
        return out

    def li_image(self, spp_total, s0=0, s1=None, seed=0, table=None, draws_per_sample=0, threads=0):
        """Sum over samples [s0, s1) of Li per pixel, (H, W, 3) float32, plus the counters."""
        s1 = spp_total if s1 is None else s1
        out = np.zeros((self.height, self.width, 3), np.float32)
        stats = PtoStats()
        tab = _f32(table) if table is not None else None
        _check(lib().pto_li_image(self._h, int(spp_total), int(s0), int(s1), 1 if tab is not None else 0,
                                  C.c_uint64(seed), _p(tab), C.c_uint64(0 if tab is None else tab.size),
                                  int(draws_per_sample), int(threads), _p(out), C.byref(stats)))
        return out, stats.as_dict()
