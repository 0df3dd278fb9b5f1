"""Binding of oracle/_ref/libptref.so: the reference's own hot-path code behind a harness.

TEST INFRASTRUCTURE ONLY - imported by tests/, by the fixture generator under tests/golden and by the
cpu_baseline leg of bench.py, never by the product package.  The library is built by oracle/Makefile.ref
from /root/reference and is absent wherever the reference is (it is not kept in git); available() tells.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libptref.so")
_LIB = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(_PATH)
        L.ref_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


class RefError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise RefError(lib().ref_last_error().decode())


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


# ---------------------------------------------------------------- leaf functions
def ray_aabb(o, inv_dir, bmin, bmax, max_t):
    o, inv_dir, bmin, bmax, max_t = map(_f32, (o, inv_dir, bmin, bmax, max_t))
    n = len(max_t)
    hit = np.zeros(n, np.uint8)
    _check(lib().ref_ray_aabb(C.c_uint64(n), _p(o), _p(inv_dir), _p(bmin), _p(bmax), _p(max_t), _p(hit)))
    return hit


def ray_triangle(o, d, v0, v1, v2, max_t):
    o, d, v0, v1, v2, max_t = map(_f32, (o, d, v0, v1, v2, max_t))
    n = len(max_t)
    hit = np.zeros(n, np.uint8)
    t, u, v = (np.zeros(n, np.float32) for _ in range(3))
    _check(lib().ref_ray_triangle(C.c_uint64(n), _p(o), _p(d), _p(v0), _p(v1), _p(v2), _p(max_t), _p(hit), _p(t), _p(u), _p(v)))
    return hit, t, u, v


def ray_sphere(o, d, centre, radius, max_t):
    o, d, centre, radius, max_t = map(_f32, (o, d, centre, radius, max_t))
    n = len(max_t)
    hit = np.zeros(n, np.uint8)
    t = np.zeros(n, np.float32)
    _check(lib().ref_ray_sphere(C.c_uint64(n), _p(o), _p(d), _p(centre), _p(radius), _p(max_t), _p(hit), _p(t)))
    return hit, t


def sampling(u1, u2):
    u1, u2 = _f32(u1), _f32(u2)
    n = len(u1)
    cos_h, disk = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
    sphere, tri = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
    _check(lib().ref_sampling(C.c_uint64(n), _p(u1), _p(u2), _p(cos_h), _p(disk), _p(sphere), _p(tri)))
    return {"cosine_hemisphere": cos_h, "concentric_disk": disk, "uniform_sphere": sphere, "uniform_triangle": tri}


def tonemap(rgb, exposure, gamma):
    rgb = _f32(rgb).reshape(-1, 3)
    a, b = np.zeros_like(rgb), np.zeros_like(rgb)
    _check(lib().ref_tonemap(C.c_uint64(len(rgb)), _p(rgb), C.c_float(exposure), C.c_float(gamma), _p(a), _p(b)))
    return a, b


def load_texture(filename, flip=True):
    w, h = C.c_int32(0), C.c_int32(0)
    _check(lib().ref_load_texture(filename.encode(), int(flip), C.byref(w), C.byref(h), None, C.c_uint64(0)))
    px = np.zeros((h.value, w.value, 4), np.uint8)
    _check(lib().ref_load_texture(filename.encode(), int(flip), C.byref(w), C.byref(h), _p(px), C.c_uint64(px.nbytes)))
    return px


def load_skybox_face(filename, flip=True):
    """One cube map face as Skybox::Load decodes it (stbi_loadf, 4 channels, hdr scale and gamma 1): (h, w, 4) floats."""
    w, h = C.c_int32(0), C.c_int32(0)
    _check(lib().ref_load_skybox_face(filename.encode(), int(flip), C.byref(w), C.byref(h), None, C.c_uint64(0)))
    px = np.zeros((h.value, w.value, 4), np.float32)
    _check(lib().ref_load_skybox_face(filename.encode(), int(flip), C.byref(w), C.byref(h), _p(px), C.c_uint64(px.size)))
    return px


def save_image(filename, rgb):
    """The reference's Image::Save (quantiser + stb_image_write) on an (h, w, 3) float picture."""
    rgb = _f32(rgb)
    h, w = rgb.shape[:2]
    _check(lib().ref_save_image(filename.encode(), _p(rgb), w, h))


def bvh_build(desc, split_method=0):
    from pathtracer_amd import abi
    n = desc.num_prims
    nodes = np.zeros(2 * n - 1, abi.NODE_DTYPE)
    order = np.zeros(n, np.uint32)
    count = C.c_uint64(0)
    _check(lib().ref_bvh_build(C.byref(desc), split_method, _p(nodes), _p(order), C.byref(count)))
    return nodes[:count.value], order


def bvh_build_with(desc, build_fn, device=0):
    """A PT::BVH filled through the reference-side binding of the GPU builder (oracle/pt_bvh_binding.hpp), with
    `build_fn` (the ctypes function pt_build_bvh) in place of BVH::Build; same outputs as bvh_build."""
    from pathtracer_amd import abi
    n = desc.num_prims
    nodes = np.zeros(2 * n - 1, abi.NODE_DTYPE)
    order = np.zeros(n, np.uint32)
    count = C.c_uint64(0)
    lib().ref_bvh_build_with.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    _check(lib().ref_bvh_build_with(C.byref(desc), C.cast(build_fn, C.c_void_p), device, _p(nodes), _p(order), C.byref(count)))
    return nodes[:count.value], order


# ---------------------------------------------------------------- scenes
class RefScene:
    def __init__(self, desc):
        self._desc = desc
        self._h = C.c_void_p()
        _check(lib().ref_scene_create(C.byref(desc), C.byref(self._h)))
        self.width, self.height = desc.image_width, desc.image_height

    def close(self):
        if self._h:
            lib().ref_scene_destroy(self._h)
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
        _check(lib().ref_intersect(self._h, C.c_uint64(n), _p(o), _p(d), _p(out["prim"]), _p(out["t"]), _p(out["u"]),
                                   _p(out["v"]), _p(out["position"]), _p(out["normal"]), _p(out["tangent"]),
                                   _p(out["bitangent"]), _p(out["uv"]), _p(out["material"]), _p(out["nodes"]),
                                   _p(out["prim_tests"])))
        return out

    def occluded(self, origins, dirs, t_max):
        o, d, tm = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3), _f32(t_max)
        occ = np.zeros(len(o), np.uint8)
        _check(lib().ref_occluded(self._h, C.c_uint64(len(o)), _p(o), _p(d), _p(tm), _p(occ)))
        return occ

    def texture_fetch(self, tex, u, v):
        u, v = _f32(u), _f32(v)
        out = np.zeros((len(u), 4), np.float32)
        _check(lib().ref_texture_fetch(self._h, int(tex), C.c_uint64(len(u)), _p(u), _p(v), _p(out)))
        return out

    def li_image(self, spp_total, s0=0, s1=None, seed=0, table=None, draws_per_sample=0, emulate_stale_origin=False):
        """Sum over samples [s0, s1) of Li per pixel, (H, W, 3) float32, plus the ray counts."""
        s1 = spp_total if s1 is None else s1
        out = np.zeros((self.height, self.width, 3), np.float32)
        rc, rs = C.c_uint64(0), C.c_uint64(0)
        tab = _f32(table) if table is not None else None
        _check(lib().ref_li_image(self._h, int(spp_total), int(s0), int(s1), 1 if tab is not None else 0,
                                  C.c_uint64(seed), _p(tab), C.c_uint64(0 if tab is None else tab.size),
                                  int(draws_per_sample), _p(out), C.byref(rc), C.byref(rs), int(emulate_stale_origin)))
        return out, {"rays_closest": rc.value, "rays_shadow": rs.value}

    def render(self, camera_rotation_rad, spp, seed=0):
        """The reference's own PathTracer::Render: tonemapped image (H, W, 3)."""
        rot = _f32(camera_rotation_rad)
        out = np.zeros((self.height, self.width, 3), np.float32)
        _check(lib().ref_render(self._h, _p(rot), int(spp), C.c_uint64(seed), _p(out)))
        return out

    def flatten(self, camera_rotation_rad):
        """The reference-side binding (oracle/pt_flatten.hpp) on this scene's PT::Scene: a SceneData copied from
        the PtSceneDesc that a maintainer's `Flatten( scene )` hands to pt_upload_scene."""
        from pathtracer_amd import abi
        from pathtracer_amd.scene import SceneData
        rot = _f32(camera_rotation_rad)
        desc = abi.PtSceneDesc()
        _check(lib().ref_flatten(self._h, _p(rot), C.byref(desc)))
        return SceneData.from_desc(desc)

    def render_gpu(self, camera_rotation_rad, spp, seed=0, save_as=None, tonemap_on_gpu=False):
        """The reference's PT::Scene / PathTracer / SaveImage with Render's body replaced by the binding of
        oracle/pt_render_binding.hpp, on the GPU library: tonemapped image (H, W, 3).  Needs a GPU."""
        from pathtracer_amd import api
        gpu = api.lib()
        entry = (C.c_void_p * 5)(*[C.cast(getattr(gpu, n), C.c_void_p) for n in
                                    ("pt_upload_scene", "pt_render", "pt_readback", "pt_destroy_scene", "pt_last_error")])
        rot = _f32(camera_rotation_rad)
        out = np.zeros((self.height, self.width, 3), np.float32)
        _check(lib().ref_render_gpu(self._h, _p(rot), int(spp), C.c_uint64(seed), entry, int(tonemap_on_gpu), _p(out),
                                    save_as.encode() if save_as else None))
        return out
