"""Time of pt_upload_scene (host-side layout work + copies) for the 20 M triangle soup, and of the scene's generation."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, host

p = argparse.ArgumentParser()
p.add_argument("--triangles", type=int, default=20_000_000)
a = p.parse_args()
t = time.time()
with host.HostScene.soup(seed=3, triangles=a.triangles, edge=0.01, skybox_size=1024) as hs:
    print(f"scene generation and BVH build {time.time() - t:.2f} s", flush=True)
    for k in range(2):
        t = time.time()
        with api.DeviceScene(hs.desc()) as dev:
            print(f"pt_upload_scene {time.time() - t:.2f} s", flush=True)
            dev.render(1, seed=1, ordered=True)
            lin, _, st = dev.readback()
            print("checksum", float(lin.sum()), st.rays_closest, flush=True)
