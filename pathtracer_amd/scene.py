"""SceneData: a flattened scene held in numpy arrays, convertible to and from the C-ABI's PtSceneDesc.

This is the Python-side carrier for the arrays that cross the drop-in boundary (include/pt_api.h): fixtures
under tests/golden are SceneData saved as .npz, and every library in this repository (the HIP core, the CPU
checker, the reference harness) is driven with the PtSceneDesc it produces.
"""
import ctypes as C

import numpy as np

from . import abi

_ARRAYS = {
    "nodes": abi.NODE_DTYPE, "prims": abi.PRIM_DTYPE, "triangles": abi.TRIANGLE_DTYPE,
    "spheres": abi.SPHERE_DTYPE, "meshes": abi.MESH_DTYPE, "materials": abi.MATERIAL_DTYPE,
    "textures": abi.TEXTURE_DTYPE, "lights": abi.LIGHT_DTYPE,
}


def _from_ptr(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    nbytes = int(count) * np.dtype(dtype).itemsize
    buf = (C.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=int(count)).copy()


class SceneData:
    """All arrays of one flattened scene plus the scalar settings Render reads (src/scene.hpp:27-37)."""

    def __init__(self):
        for name, dt in _ARRAYS.items():
            setattr(self, name, np.zeros(0, dtype=dt))
        self.vertices = np.zeros((0, 3), np.float32)
        self.normals = np.zeros((0, 3), np.float32)
        self.tangents = np.zeros((0, 3), np.float32)
        self.uvs = np.zeros((0, 2), np.float32)
        self.texels = np.zeros((0, 4), np.uint8)
        self.skybox = np.zeros(0, np.float32)  # 6 * h * w * 4, empty = none
        self.skybox_width = 0
        self.skybox_height = 0
        self.background_radiance = np.zeros(3, np.float32)
        self.camera = {"position": np.zeros(3, np.float32), "view_dir": np.array([0, 0, -1], np.float32),
                       "up_dir": np.array([0, 1, 0], np.float32), "right_dir": np.array([1, 0, 0], np.float32),
                       "vfov": np.float32(np.radians(45.0)), "aspect_ratio": np.float32(16.0 / 9.0),
                       "exposure": np.float32(1), "gamma": np.float32(1)}
        self.image_width = 1280
        self.image_height = 720
        self.max_depth = 5

    # ------------------------------------------------------------------ C-ABI
    @classmethod
    def from_desc(cls, desc):
        """Copies everything a PtSceneDesc points to."""
        s = cls()
        for name, dt in _ARRAYS.items():
            setattr(s, name, _from_ptr(getattr(desc, name), getattr(desc, "num_" + name), dt))
        nv = desc.num_vertices
        s.vertices = _from_ptr(desc.vertices, nv * 3, np.float32).reshape(-1, 3)
        s.normals = _from_ptr(desc.normals, nv * 3, np.float32).reshape(-1, 3)
        s.tangents = _from_ptr(desc.tangents, nv * 3, np.float32).reshape(-1, 3)
        s.uvs = _from_ptr(desc.uvs, nv * 2, np.float32).reshape(-1, 2)
        s.texels = _from_ptr(desc.texels, desc.num_texels * 4, np.uint8).reshape(-1, 4)
        s.skybox_width, s.skybox_height = desc.skybox_width, desc.skybox_height
        s.skybox = _from_ptr(desc.skybox, 6 * desc.skybox_width * desc.skybox_height * 4 if desc.skybox else 0,
                             np.float32)
        s.background_radiance = np.array(list(desc.background_radiance), np.float32)
        cam = desc.camera
        s.camera = {"position": np.array(list(cam.position), np.float32),
                    "view_dir": np.array(list(cam.view_dir), np.float32),
                    "up_dir": np.array(list(cam.up_dir), np.float32),
                    "right_dir": np.array(list(cam.right_dir), np.float32),
                    "vfov": np.float32(cam.vfov), "aspect_ratio": np.float32(cam.aspect_ratio),
                    "exposure": np.float32(cam.exposure), "gamma": np.float32(cam.gamma)}
        s.image_width, s.image_height, s.max_depth = desc.image_width, desc.image_height, desc.max_depth
        return s

    def to_desc(self, device=0, flags=0, top_nodes=0):
        """Returns a PtSceneDesc pointing into this object's arrays.  The descriptor keeps the arrays alive
        through an attribute, but the SceneData must not be modified while the descriptor is in use."""
        d = abi.PtSceneDesc()
        d.struct_size = C.sizeof(abi.PtSceneDesc)
        d.version = abi.PT_API_VERSION
        keep = []

        def ptr(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            keep.append(a)
            return a.ctypes.data if a.size else None

        for name, dt in _ARRAYS.items():
            arr = getattr(self, name)
            setattr(d, name, ptr(arr, dt))
            setattr(d, "num_" + name, len(arr))
        d.vertices = ptr(self.vertices, np.float32)
        d.normals = ptr(self.normals, np.float32)
        d.tangents = ptr(self.tangents, np.float32)
        d.uvs = ptr(self.uvs, np.float32)
        d.num_vertices = len(self.vertices)
        d.texels = ptr(self.texels, np.uint8)
        d.num_texels = len(self.texels)
        d.skybox = ptr(self.skybox, np.float32)
        d.skybox_width, d.skybox_height = int(self.skybox_width), int(self.skybox_height)
        d.background_radiance = (C.c_float * 3)(*[float(x) for x in self.background_radiance])
        for key in ("position", "view_dir", "up_dir", "right_dir"):
            setattr(d.camera, key, (C.c_float * 3)(*[float(x) for x in self.camera[key]]))
        for key in ("vfov", "aspect_ratio", "exposure", "gamma"):
            setattr(d.camera, key, float(self.camera[key]))
        d.image_width, d.image_height = int(self.image_width), int(self.image_height)
        d.max_depth = int(self.max_depth)
        d.device = int(device)
        d.flags, d.top_nodes = int(flags), int(top_nodes)
        d._keep = keep
        return d

    # ------------------------------------------------------------------ files
    def save(self, path):
        cam = np.concatenate([self.camera[k] for k in ("position", "view_dir", "up_dir", "right_dir")]
                             + [np.array([self.camera[k] for k in ("vfov", "aspect_ratio", "exposure", "gamma")],
                                         np.float32)]).astype(np.float32)
        np.savez_compressed(
            path, **{n: getattr(self, n) for n in _ARRAYS},
            vertices=self.vertices, normals=self.normals, tangents=self.tangents, uvs=self.uvs,
            texels=self.texels, skybox=self.skybox,
            skybox_size=np.array([self.skybox_width, self.skybox_height], np.int32),
            background_radiance=self.background_radiance, camera=cam,
            settings=np.array([self.image_width, self.image_height, self.max_depth], np.int32))

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        s = cls()
        for n, dt in _ARRAYS.items():
            setattr(s, n, z[n].astype(dt))
        s.vertices, s.normals, s.tangents, s.uvs = z["vertices"], z["normals"], z["tangents"], z["uvs"]
        s.texels, s.skybox = z["texels"], z["skybox"]
        s.skybox_width, s.skybox_height = [int(x) for x in z["skybox_size"]]
        s.background_radiance = z["background_radiance"]
        cam = z["camera"]
        s.camera = {"position": cam[0:3], "view_dir": cam[3:6], "up_dir": cam[6:9], "right_dir": cam[9:12],
                    "vfov": cam[12], "aspect_ratio": cam[13], "exposure": cam[14], "gamma": cam[15]}
        s.image_width, s.image_height, s.max_depth = [int(x) for x in z["settings"]]
        return s

    # ------------------------------------------------------------------ helpers
    def copy(self):
        import copy
        return copy.deepcopy(self)

    def with_image(self, width, height, aspect_ratio=None):
        s = self.copy()
        s.image_width, s.image_height = int(width), int(height)
        if aspect_ratio is not None:
            s.camera["aspect_ratio"] = np.float32(aspect_ratio)
        return s

    def draws_per_bounce(self):
        """Random draws one bounce consumes: two per area-light sample plus two for the BRDF sample
        (src/shapes.cpp:45,107 and src/resource/material.cpp:22)."""
        area = self.lights[self.lights["type"] == abi.PT_LIGHT_AREA]
        return 2 * int(area["n_samples"].sum()) + 2

    def draws_per_sample(self):
        return 2 + self.max_depth * self.draws_per_bounce()
