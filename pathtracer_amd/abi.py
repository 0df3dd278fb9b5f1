"""ctypes / numpy mirror of include/pt_api.h and include/pt_host.h.

Every structure here must stay byte-compatible with the C headers; tests/test_abi.py checks the sizes
against the values the C compiler reports through the libraries.
"""
import ctypes as C

import numpy as np

PT_API_VERSION = 2

PT_OK, PT_ERR_INVALID_ARG, PT_ERR_INVALID_SCENE, PT_ERR_DEVICE, PT_ERR_STATE = range(5)
PT_PRIM_TRIANGLE, PT_PRIM_SPHERE = 0, 1
PT_LIGHT_POINT, PT_LIGHT_DIRECTIONAL, PT_LIGHT_AREA = 0, 1, 2
PT_RNG_PHILOX, PT_RNG_TABLE = 0, 1
PT_TONEMAP_UNCHARTED2, PT_TONEMAP_REINHARD, PT_TONEMAP_NONE = 0, 1, 2
PT_RENDER_ACCUMULATE, PT_RENDER_PROFILE, PT_RENDER_ORDERED, PT_RENDER_WIDE, PT_RENDER_WIDE_SHADOWS = 1, 2, 4, 8, 16
PT_TRAVERSAL_REFERENCE, PT_TRAVERSAL_ORDERED, PT_TRAVERSAL_WIDE = 0, 1, 2
PT_TRACE_STATIC_MAX_RAYS = 262144
PT_SCENE_NO_DIRECT_SUM, PT_SCENE_TOP_BREADTH_FIRST, PT_SCENE_NO_ORDERED_LAYOUT, PT_SCENE_NO_PARALLEL_SHADOWS = 1, 2, 4, 8
PT_SCENE_NO_WIDE_LAYOUT = 16
PT_SCENE_HOST_LAYOUTS = 32
(PT_LAYOUT_REFERENCE_NODES, PT_LAYOUT_ORDERED_NODES, PT_LAYOUT_PRIMS, PT_LAYOUT_ATTRIBUTES, PT_LAYOUT_WIDE_NODES, PT_LAYOUT_WIDE_PRIMS,
 PT_LAYOUT_SHADOW_NODES, PT_LAYOUT_SHADOW_TOP, PT_LAYOUT_LIGHT_WIDE_NODES) = range(9)
PT_MERGE_STAGE_ALWAYS = 1
PT_K_RAYGEN, PT_K_EXTEND, PT_K_SHADE, PT_K_CONNECT, PT_K_FILM, PT_K_TONEMAP, PT_K_RESOLVE, PT_K_COUNT = range(8)
KERNEL_NAMES = ("raygen", "extend", "shade", "connect", "film", "tonemap", "resolve")
PTH_SPLIT_SAH, PTH_SPLIT_MIDDLE, PTH_SPLIT_EQUAL_COUNTS = 0, 1, 2

# numpy record types of the array elements
NODE_DTYPE = np.dtype([("aabb_min", "<f4", 3), ("aabb_max", "<f4", 3), ("offset", "<i4"),
                       ("num_shapes", "<u2"), ("axis", "u1"), ("pad", "u1")])
PRIM_DTYPE = np.dtype([("type", "<u4"), ("index", "<u4")])
TRIANGLE_DTYPE = np.dtype([("i0", "<u4"), ("i1", "<u4"), ("i2", "<u4"), ("mesh", "<u4")])
MESH_DTYPE = np.dtype([("vertex_offset", "<u4"), ("num_vertices", "<u4"), ("material", "<i4"), ("reserved", "<u4")])
SPHERE_DTYPE = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("world_to_local", "<f4", 16), ("material", "<i4")])
MATERIAL_DTYPE = np.dtype([("albedo", "<f4", 3), ("Ke", "<f4", 3), ("albedo_texture", "<i4")])
TEXTURE_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("texel_offset", "<u8")])
LIGHT_DTYPE = np.dtype([("type", "<u4"), ("n_samples", "<i4"), ("Lemit", "<f4", 3), ("position", "<f4", 3),
                        ("direction", "<f4", 3), ("prim", "<i4")])

assert NODE_DTYPE.itemsize == 32 and PRIM_DTYPE.itemsize == 8 and TRIANGLE_DTYPE.itemsize == 16
assert MESH_DTYPE.itemsize == 16 and SPHERE_DTYPE.itemsize == 84 and MATERIAL_DTYPE.itemsize == 28
assert TEXTURE_DTYPE.itemsize == 16 and LIGHT_DTYPE.itemsize == 48


class PtCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("view_dir", C.c_float * 3), ("up_dir", C.c_float * 3),
                ("right_dir", C.c_float * 3), ("vfov", C.c_float), ("aspect_ratio", C.c_float),
                ("exposure", C.c_float), ("gamma", C.c_float)]


class PtSceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("version", C.c_uint32),
        ("nodes", C.c_void_p), ("num_nodes", C.c_uint64),
        ("prims", C.c_void_p), ("num_prims", C.c_uint64),
        ("triangles", C.c_void_p), ("num_triangles", C.c_uint64),
        ("spheres", C.c_void_p), ("num_spheres", C.c_uint64),
        ("meshes", C.c_void_p), ("num_meshes", C.c_uint64),
        ("vertices", C.c_void_p), ("normals", C.c_void_p), ("tangents", C.c_void_p), ("uvs", C.c_void_p),
        ("num_vertices", C.c_uint64),
        ("materials", C.c_void_p), ("num_materials", C.c_uint64),
        ("textures", C.c_void_p), ("num_textures", C.c_uint64),
        ("texels", C.c_void_p), ("num_texels", C.c_uint64),
        ("lights", C.c_void_p), ("num_lights", C.c_uint64),
        ("skybox", C.c_void_p), ("skybox_width", C.c_int32), ("skybox_height", C.c_int32),
        ("background_radiance", C.c_float * 3),
        ("camera", PtCamera),
        ("image_width", C.c_int32), ("image_height", C.c_int32),
        ("max_depth", C.c_int32), ("device", C.c_int32),
        ("flags", C.c_uint32), ("top_nodes", C.c_uint32),
    ]


class PtRenderParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("flags", C.c_uint32),
        ("spp_total", C.c_int32), ("sample_begin", C.c_int32), ("sample_end", C.c_int32),
        ("rng_mode", C.c_uint32), ("seed", C.c_uint64),
        ("sample_table", C.c_void_p), ("sample_table_len", C.c_uint64),
        ("table_draws_per_sample", C.c_int32), ("samples_per_pass", C.c_int32),
        ("stream", C.c_void_p),
    ]


class PtStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("nodes_closest", C.c_uint64), ("nodes_shadow", C.c_uint64),
        ("tri_tests_closest", C.c_uint64), ("tri_tests_shadow", C.c_uint64),
        ("sphere_tests_closest", C.c_uint64), ("sphere_tests_shadow", C.c_uint64),
        ("shaded_hits", C.c_uint64), ("node_steps_closest", C.c_uint64), ("node_steps_shadow", C.c_uint64),
        ("rays_redone", C.c_uint64),
        ("launches", C.c_uint64 * PT_K_COUNT), ("kernel_ms", C.c_double * PT_K_COUNT),
        ("render_ms", C.c_double),
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_ if name not in ("launches", "kernel_ms")}
        d["launches"] = {KERNEL_NAMES[i]: self.launches[i] for i in range(PT_K_COUNT)}
        d["kernel_ms"] = {KERNEL_NAMES[i]: self.kernel_ms[i] for i in range(PT_K_COUNT)}
        return d

    def bytes_closest(self):
        """Algorithmic bytes of all closest-hit rays: 32 B per node, 48 B per triangle test, 84 B per sphere
        test, 32 B ray in, 16 B hit out."""
        return (32 * self.nodes_closest + 48 * self.tri_tests_closest + 84 * self.sphere_tests_closest
                + 48 * self.rays_closest)

    def bytes_shadow(self):
        """Algorithmic bytes of all any-hit rays: 32 B per node, 48 B per triangle test, 16 B per sphere test,
        32 B ray in, 4 B result out."""
        return (32 * self.nodes_shadow + 48 * self.tri_tests_shadow + 16 * self.sphere_tests_shadow
                + 36 * self.rays_shadow)


class PthSceneInfo(C.Structure):
    _fields_ = [("num_spp", C.c_int32), ("spp", C.c_int32 * 32), ("samples_per_area_light", C.c_int32),
                ("split_method", C.c_int32), ("output_filename", C.c_char * 512)]
