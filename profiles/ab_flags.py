"""A/B of scene flags on the bench scene: film of every variant must equal the reference order's bit for bit;
prints kernel milliseconds for each.  usage: python3 profiles/ab_flags.py [--spp N]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import abi, api, host

def main():
    p = argparse.ArgumentParser()
    p.add_argument("--spp", type=int, default=32)
    a = p.parse_args()
    hs = host.HostScene.atrium(seed=1, triangles=262267, texture_size=1024)
    hs.set_image(1920, 1080, 1920 / 1080)
    films = {}
    variants = (("ordered", 0, True), ("ordered, shadow rays through the BVH", abi.PT_SCENE_NO_PARALLEL_SHADOWS, True), ("reference", 0, False))
    for name, flags, ordered in variants:
        desc = hs.desc(0)
        desc.flags |= flags
        with api.DeviceScene(desc) as dev:
            dev.render(8, seed=3, ordered=ordered)
            t0 = time.time()
            dev.render(a.spp, seed=3, ordered=ordered, profile=True)
            dt = time.time() - t0
            films[name] = dev.download_film()
            st = dev.stats().as_dict()
            keys = ("rays_closest", "nodes_closest", "rays_shadow", "nodes_shadow", "tri_tests_shadow", "rays_redone")
            print(name, "ms %.1f" % (dt * 1e3), {k: st[k] for k in keys},
                  {k: round(v, 2) for k, v in st["kernel_ms"].items() if v}, flush=True)
    for k in films:
        same = np.array_equal(films[k].view(np.uint32), films["reference"].view(np.uint32))
        print(k, "film equals reference order:", same, flush=True)
        assert same
main()
