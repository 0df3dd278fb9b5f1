"""Binding of libpthost.so (include/pt_host.h): scene files, BVH build, procedural scenes, image output."""
import ctypes as C
import os

import numpy as np

from . import abi
from .scene import SceneData

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class HostError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpthost.so")
        if not os.path.exists(path):
            raise HostError(f"{path} is missing: run `make host` (or __graft_entry__.build()) first")
        L = C.CDLL(path)
        L.pth_scene_new.restype = C.c_void_p
        L.pth_scene_free.argtypes = [C.c_void_p]
        L.pth_scene_free.restype = None
        L.pth_load_scene_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.pth_generate_atrium.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]
        L.pth_generate_knot.argtypes = [C.c_void_p, C.c_uint32, C.c_int32]
        L.pth_generate_soup.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int32]
        L.pth_set_image.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float]
        L.pth_set_max_depth.argtypes = [C.c_void_p, C.c_int32]
        L.pth_scene_desc.argtypes = [C.c_void_p, C.c_int32, C.POINTER(abi.PtSceneDesc)]
        L.pth_scene_info.argtypes = [C.c_void_p, C.POINTER(abi.PthSceneInfo)]
        L.pth_build_bvh.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.pth_set_bvh_builder.argtypes = [C.c_void_p, C.c_int32]
        L.pth_camera_vectors.argtypes = [C.c_void_p] * 4
        L.pth_sphere_world_to_local.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.pth_write_image.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.pth_read_image.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_uint64]
        L.pth_tonemap.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_int32]
        L.pth_read_image_float.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_uint64]
        L.pth_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise HostError(lib().pth_last_error().decode())


class HostScene:
    """Owner of a C++ host scene (the role of the reference's Scene object above the boundary)."""

    def __init__(self):
        self._h = lib().pth_scene_new()
        if not self._h:
            raise HostError("pth_scene_new failed")

    def close(self):
        if self._h:
            lib().pth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @classmethod
    def from_file(cls, filename, resource_dir):
        s = cls()
        if not resource_dir.endswith("/"):
            resource_dir += "/"
        _check(lib().pth_load_scene_file(s._h, filename.encode(), resource_dir.encode()))
        return s

    @classmethod
    def atrium(cls, seed=1, triangles=262267, texture_size=1024):
        s = cls()
        _check(lib().pth_generate_atrium(s._h, seed, triangles, texture_size))
        return s

    @classmethod
    def knot(cls, seed=2, triangles=871414):
        s = cls()
        _check(lib().pth_generate_knot(s._h, seed, triangles))
        return s

    @classmethod
    def soup(cls, seed=3, triangles=20_000_000, edge=0.01, skybox_size=1024):
        s = cls()
        _check(lib().pth_generate_soup(s._h, seed, triangles, edge, skybox_size))
        return s

    def set_image(self, width, height, aspect_ratio=-1.0):
        _check(lib().pth_set_image(self._h, width, height, aspect_ratio))

    def set_max_depth(self, depth):
        _check(lib().pth_set_max_depth(self._h, depth))

    def desc(self, device=0):
        """PtSceneDesc pointing into this host scene (valid while the scene is alive and unchanged)."""
        d = abi.PtSceneDesc()
        _check(lib().pth_scene_desc(self._h, device, C.byref(d)))
        d._keep = self
        return d

    def info(self):
        i = abi.PthSceneInfo()
        _check(lib().pth_scene_info(self._h, C.byref(i)))
        return {"spp": [i.spp[k] for k in range(i.num_spp)], "samples_per_area_light": i.samples_per_area_light,
                "split_method": i.split_method, "output_filename": i.output_filename.decode()}

    def to_data(self):
        return SceneData.from_desc(self.desc())


def use_gpu_bvh_builder(device=0):
    """Scenes loaded or generated from now on build their SAH trees with pt_build_bvh on GPU `device`
    (None: back to the host builder).  The trees are the same node for node."""
    if device is None:
        _check(lib().pth_set_bvh_builder(None, 0))
        return
    from . import api
    _check(lib().pth_set_bvh_builder(C.cast(api.lib().pt_build_bvh, C.c_void_p), device))


def build_bvh(boxes, split_method=abi.PTH_SPLIT_SAH):
    """boxes: (n, 6) float32 [min xyz, max xyz] -> (nodes record array, order uint32 array)."""
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    n = len(boxes)
    nodes = np.zeros(2 * n - 1, abi.NODE_DTYPE)
    order = np.zeros(n, np.uint32)
    count = C.c_uint64(0)
    _check(lib().pth_build_bvh(boxes.ctypes.data, n, split_method, nodes.ctypes.data, order.ctypes.data, C.byref(count)))
    return nodes[:count.value], order


def camera_vectors(rotation_rad):
    r = np.ascontiguousarray(rotation_rad, np.float32)
    view, up, right = (np.zeros(3, np.float32) for _ in range(3))
    _check(lib().pth_camera_vectors(r.ctypes.data, view.ctypes.data, up.ctypes.data, right.ctypes.data))
    return view, up, right


def sphere_world_to_local(position, rotation_rad, radius):
    p = np.ascontiguousarray(position, np.float32)
    r = np.ascontiguousarray(rotation_rad, np.float32)
    out = np.zeros(16, np.float32)
    _check(lib().pth_sphere_world_to_local(p.ctypes.data, r.ctypes.data, float(radius), out.ctypes.data))
    return out


def read_image(filename, flip=True):
    """RGBA8 pixels (h, w, 4) as Texture::Load decodes them (src/resource/texture.cpp:19-33)."""
    w, h = C.c_int32(0), C.c_int32(0)
    _check(lib().pth_read_image(filename.encode(), int(flip), C.byref(w), C.byref(h), None, 0))
    px = np.zeros((h.value, w.value, 4), np.uint8)
    _check(lib().pth_read_image(filename.encode(), int(flip), C.byref(w), C.byref(h), px.ctypes.data, px.nbytes))
    return px


def tonemap(rgb, exposure, gamma, op=0):
    """The reference's tonemap pass (src/path_tracer.cpp:211-219) on the host: Uncharted2 (op 0) or Reinhard (1),
    gamma, clamp - with the C library's powf, hence the reference's picture bit for bit."""
    out = np.array(rgb, np.float32, order="C")
    _check(lib().pth_tonemap(out.ctypes.data, C.c_uint64(out.size // 3), C.c_float(exposure), C.c_float(gamma), int(op)))
    return out


def read_image_float(filename, flip=True):
    """RGBA float pixels (h, w, 4) as Skybox::Load decodes a face (src/resource/skybox.cpp:33-37)."""
    w, h = C.c_int32(0), C.c_int32(0)
    _check(lib().pth_read_image_float(filename.encode(), int(flip), C.byref(w), C.byref(h), None, 0))
    px = np.zeros((h.value, w.value, 4), np.float32)
    _check(lib().pth_read_image_float(filename.encode(), int(flip), C.byref(w), C.byref(h), px.ctypes.data, px.size))
    return px


def write_image(filename, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    _check(lib().pth_write_image(filename.encode(), rgb.ctypes.data, w, h))
