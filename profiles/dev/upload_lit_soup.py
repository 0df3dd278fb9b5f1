"""pt_upload_scene by phase (PTAMD_UPLOAD_TIMING=1) for the 20 M triangle soup lit by one directional light, records
written by kernels and by the host loops.  usage: PTAMD_UPLOAD_TIMING=1 python3 profiles/dev/upload_lit_soup.py [triangles]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pathtracer_amd import abi, api, host

triangles = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
host.use_gpu_bvh_builder(0)
with host.HostScene.soup(seed=3, triangles=triangles, edge=0.01, skybox_size=64) as hs:
    for flags, what in ((0, "kernels"), (abi.PT_SCENE_HOST_LAYOUTS, "host loops")):
        desc = hs.desc(0)
        desc.flags |= flags
        light = np.zeros(1, abi.LIGHT_DTYPE)
        light["type"], light["n_samples"], light["prim"] = abi.PT_LIGHT_DIRECTIONAL, 1, -1
        light["Lemit"], light["direction"] = (5, 5, 5), np.asarray((0.3, -0.8, 0.5), np.float32)
        desc.lights, desc.num_lights = light.ctypes.data, 1
        print(f"== records written by {what}", flush=True)
        t = time.perf_counter()
        with api.DeviceScene(desc) as dev:
            print(f"pt_upload_scene {1e3 * (time.perf_counter() - t):.0f} ms, light-space records {len(dev.read_layout(abi.PT_LAYOUT_SHADOW_NODES))} bytes", flush=True)
