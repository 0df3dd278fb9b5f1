"""Binding of libptamd.so, the HIP path-tracing core, through its C-ABI (include/pt_api.h).

There is no CPU fallback: if the library is missing or no GPU is visible, every entry point raises.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTAMD_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("PTAMD_LIB") or os.path.join(_HERE, "libptamd.so")
_LIB = None


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[pt error {code}] {message}")
        self.code = code


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch's wheel carries its own libamdhip64.so.7 and libptamd.so asks for the
    same soname, so whichever is loaded first serves both.  Device pointers are handed to torch
    (path_tracer.film_tensor) and torch cannot start on the system copy once that is loaded, so when PyTorch is
    installed but not imported yet its copy is loaded first.  Without PyTorch the system runtime is used."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PtError(abi.PT_ERR_DEVICE, f"{LIB_PATH} is missing: build it with `make device` "
                                             f"(or __graft_entry__.build()); there is no CPU fallback")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        L.pt_upload_scene.argtypes = [C.POINTER(abi.PtSceneDesc), C.POINTER(C.c_void_p)]
        L.pt_render.argtypes = [C.c_void_p, C.POINTER(abi.PtRenderParams)]
        L.pt_readback.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(abi.PtStats)]
        L.pt_trace_closest.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 8
        L.pt_trace_any.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 6
        L.pt_trace_closest_mode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 8
        L.pt_trace_closest_static.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 8
        L.pt_trace_any_mode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 6
        L.pt_film_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.pt_clear_film.argtypes = [C.c_void_p]
        L.pt_read_layout.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.pt_download_film.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_upload_film.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_merge_films.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32]
        L.pt_merge_films_ex.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32]
        L.pt_destroy_scene.argtypes = [C.c_void_p]
        L.pt_build_bvh.argtypes = [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_float)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise PtError(rc, lib().pt_last_error().decode())


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def build_bvh(boxes, device=0, return_ms=False):
    """BVH::Build with the SAH split (src/bvh.cpp:204-228) on the GPU.

    boxes: (n, 6) float32 [min xyz, max xyz] -> (nodes record array of 2n - 1, order uint32 array of n); the same
    result as pathtracer_amd.host.build_bvh, which runs on the CPU.  With return_ms also the build time on the GPU."""
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    n = len(boxes)
    nodes = np.zeros(max(2 * n - 1, 0), abi.NODE_DTYPE)
    order = np.zeros(n, np.uint32)
    count, ms = C.c_uint64(0), C.c_float(0)
    _check(lib().pt_build_bvh(device, boxes.ctypes.data, n, nodes.ctypes.data, order.ctypes.data, C.byref(count), C.byref(ms)))
    nodes = nodes[:count.value]
    return (nodes, order, ms.value) if return_ms else (nodes, order)


class DeviceScene:
    """A scene resident on one GPU (PtScene*)."""

    def __init__(self, desc):
        self._h = C.c_void_p()
        _check(lib().pt_upload_scene(C.byref(desc), C.byref(self._h)))
        self.width, self.height = desc.image_width, desc.image_height
        self.max_depth = desc.max_depth

    def close(self):
        if getattr(self, "_h", None):
            lib().pt_destroy_scene(self._h)
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

    def render(self, spp_total, sample_begin=0, sample_end=None, seed=0, table=None, table_draws_per_sample=0,
               accumulate=False, profile=False, samples_per_pass=0, stream=None, ordered=False, wide=False, wide_shadows=False):
        """ordered: front-to-back traversal (PT_RENDER_ORDERED) - same film, fewer node visits; wide: the same with
        the closest hits found over the eight-wide node layout (PT_RENDER_WIDE); wide_shadows: the shadow rays over that
        layout too (PT_RENDER_WIDE_SHADOWS)."""
        p = abi.PtRenderParams()
        p.struct_size = C.sizeof(abi.PtRenderParams)
        p.flags = ((abi.PT_RENDER_ACCUMULATE if accumulate else 0) | (abi.PT_RENDER_PROFILE if profile else 0)
                   | (abi.PT_RENDER_ORDERED if ordered else 0) | (abi.PT_RENDER_WIDE if wide else 0) | (abi.PT_RENDER_WIDE_SHADOWS if wide_shadows else 0))
        p.spp_total = int(spp_total)
        p.sample_begin = int(sample_begin)
        p.sample_end = int(spp_total if sample_end is None else sample_end)
        p.seed = int(seed)
        p.samples_per_pass = int(samples_per_pass)
        p.stream = stream
        keep = None
        if table is not None:
            keep = _f32(table)
            p.rng_mode = abi.PT_RNG_TABLE
            p.sample_table = keep.ctypes.data
            p.sample_table_len = keep.size
            p.table_draws_per_sample = int(table_draws_per_sample)
        else:
            p.rng_mode = abi.PT_RNG_PHILOX
        _check(lib().pt_render(self._h, C.byref(p)))

    def readback(self, tonemap=abi.PT_TONEMAP_UNCHARTED2, linear=True, tonemapped=True):
        """Returns (linear (H,W,3) or None, tonemapped (H,W,3) or None, PtStats)."""
        lin = np.zeros((self.height, self.width, 3), np.float32) if linear else None
        tm = np.zeros((self.height, self.width, 3), np.float32) if tonemapped else None
        stats = abi.PtStats()
        _check(lib().pt_readback(self._h, tonemap, lin.ctypes.data if linear else None,
                                 tm.ctypes.data if tonemapped else None, C.byref(stats)))
        return lin, tm, stats

    def stats(self):
        stats = abi.PtStats()
        _check(lib().pt_readback(self._h, 0, None, None, C.byref(stats)))
        return stats

    def clear_film(self):
        _check(lib().pt_clear_film(self._h))

    def download_film(self):
        """(H, W, 4) float32: sum r, sum g, sum b and sample count per pixel."""
        film = np.zeros((self.height, self.width, 4), np.float32)
        _check(lib().pt_download_film(self._h, film.ctypes.data))
        return film

    def upload_film(self, film):
        film = _f32(film)
        if film.shape != (self.height, self.width, 4):
            raise ValueError(f"film must be ({self.height}, {self.width}, 4)")
        _check(lib().pt_upload_film(self._h, film.ctypes.data))

    def merge_films(self, others, stage_always=False):
        """Adds the films of the other DeviceScenes (same image size, any GPU) to this one's, on this one's GPU.
        stage_always: through a copy on this GPU (PT_MERGE_STAGE_ALWAYS), the path of GPUs without peer access."""
        handles = (C.c_void_p * len(others))(*[o._h for o in others])
        _check(lib().pt_merge_films_ex(self._h, handles, len(others), abi.PT_MERGE_STAGE_ALWAYS if stage_always else 0))

    def read_layout(self, which):
        """One of the scene's record arrays (abi.PT_LAYOUT_*) as bytes, copied from the GPU."""
        nbytes = C.c_uint64()
        _check(lib().pt_read_layout(self._h, which, None, 0, C.byref(nbytes)))
        out = np.zeros(nbytes.value, np.uint8)
        if nbytes.value:
            _check(lib().pt_read_layout(self._h, which, out.ctypes.data, nbytes.value, C.byref(nbytes)))
        return out

    def film_device_ptr(self):
        ptr, nbytes = C.c_void_p(), C.c_uint64()
        _check(lib().pt_film_device_ptr(self._h, C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def trace_closest(self, origins, dirs, traversal=abi.PT_TRAVERSAL_REFERENCE):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        out = {"prim": np.zeros(n, np.int32), "t": np.zeros(n, np.float32), "u": np.zeros(n, np.float32),
               "v": np.zeros(n, np.float32), "nodes": np.zeros(n, np.uint32), "prim_tests": np.zeros(n, np.uint32)}
        _check(lib().pt_trace_closest_mode(self._h, traversal, n, o.ctypes.data, d.ctypes.data, out["prim"].ctypes.data,
                                           out["t"].ctypes.data, out["u"].ctypes.data, out["v"].ctypes.data,
                                           out["nodes"].ctypes.data, out["prim_tests"].ctypes.data))
        return out

    def trace_closest_static(self, origins, dirs):
        """pt_trace_closest_static: the same query without any device allocation (at most abi.PT_TRACE_STATIC_MAX_RAYS rays)."""
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        out = {"prim": np.zeros(n, np.int32), "t": np.zeros(n, np.float32), "u": np.zeros(n, np.float32),
               "v": np.zeros(n, np.float32), "nodes": np.zeros(n, np.uint32), "prim_tests": np.zeros(n, np.uint32)}
        _check(lib().pt_trace_closest_static(self._h, n, o.ctypes.data, d.ctypes.data, out["prim"].ctypes.data,
                                             out["t"].ctypes.data, out["u"].ctypes.data, out["v"].ctypes.data,
                                             out["nodes"].ctypes.data, out["prim_tests"].ctypes.data))
        return out

    def trace_any(self, origins, dirs, t_max, traversal=abi.PT_TRAVERSAL_REFERENCE):
        o, d, tm = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3), _f32(t_max)
        n = len(o)
        out = {"occluded": np.zeros(n, np.uint8), "nodes": np.zeros(n, np.uint32), "prim_tests": np.zeros(n, np.uint32)}
        _check(lib().pt_trace_any_mode(self._h, traversal, n, o.ctypes.data, d.ctypes.data, tm.ctypes.data,
                                       out["occluded"].ctypes.data, out["nodes"].ctypes.data, out["prim_tests"].ctypes.data))
        return out
