"""The C-ABI libraries load and export every symbol their headers declare; the ctypes mirror has the C layout.
CPU only: no compute entry point is called."""
import ctypes as C
import os
import re
import subprocess

import pytest

from pathtracer_amd import abi, api, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pth?_[a-z_0-9]+)\s*\(", text)))


def test_libptamd_exports_every_declared_entry_point():
    names = declared_functions("pt_api.h")
    assert {"pt_upload_scene", "pt_render", "pt_readback", "pt_trace_closest", "pt_trace_any", "pt_destroy_scene",
            "pt_last_error", "pt_film_device_ptr", "pt_clear_film"} <= set(names)
    lib = api.lib()
    for n in names:
        assert hasattr(lib, n), f"libptamd.so does not export {n}"


def test_libpthost_exports_every_declared_entry_point():
    lib = host.lib()
    names = declared_functions("pt_host.h")
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"libpthost.so does not export {n}"


def test_libptamd_contains_gfx950_code():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", api.LIB_PATH], capture_output=True, text=True)
    assert "gfx950" in out.stdout + out.stderr


def test_ctypes_mirror_matches_the_c_layout(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pt_host.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(PtBvhNode),sizeof(PtPrim),sizeof(PtTriangle),sizeof(PtMesh),sizeof(PtSphere),sizeof(PtMaterial),sizeof(PtTexture),"
                   "sizeof(PtLight),sizeof(PtCamera),sizeof(PtSceneDesc),sizeof(PtRenderParams),sizeof(PtStats),sizeof(PthSceneInfo),"
                   "offsetof(PtSceneDesc,camera),offsetof(PtRenderParams,sample_table));return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    mine = [abi.NODE_DTYPE.itemsize, abi.PRIM_DTYPE.itemsize, abi.TRIANGLE_DTYPE.itemsize, abi.MESH_DTYPE.itemsize,
            abi.SPHERE_DTYPE.itemsize, abi.MATERIAL_DTYPE.itemsize, abi.TEXTURE_DTYPE.itemsize, abi.LIGHT_DTYPE.itemsize,
            C.sizeof(abi.PtCamera), C.sizeof(abi.PtSceneDesc), C.sizeof(abi.PtRenderParams), C.sizeof(abi.PtStats),
            C.sizeof(abi.PthSceneInfo), abi.PtSceneDesc.camera.offset, abi.PtRenderParams.sample_table.offset]
    assert mine == sizes
    assert sizes[0] == 32   # the reference's LinearBVHNode, src/bvh.hpp:10-21


def test_product_fails_loudly_without_its_library(monkeypatch):
    """There is no CPU fallback: a missing libptamd.so is an error, not a slow path."""
    monkeypatch.setattr(api, "_LIB", None)
    monkeypatch.setattr(api, "LIB_PATH", "/nonexistent/libptamd.so")
    with pytest.raises(api.PtError, match="no CPU fallback"):
        api.lib()


def test_product_does_not_import_the_checker():
    """Nothing under pathtracer_amd/ may use oracle/: the checker is test infrastructure."""
    pkg = os.path.join(ROOT, "pathtracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text and "pt_oracle" not in text and "libptref" not in text, os.path.join(dirpath, f)
