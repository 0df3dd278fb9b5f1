"""Do two renders of the same frame on two streams of one GPU finish sooner than one render of twice the samples?
One scene at S spp per step against two uploads of it at S / 2 each, rendered from two host threads (pt_render drops the
GIL).
usage: python3 profiles/dev/two_streams.py [spp per step] [steps]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pathtracer_amd import api, host

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kw = dict(seed=7, accumulate=True, ordered=True, wide=True, wide_shadows=True)

hs = host.HostScene.atrium(seed=1, triangles=262267, texture_size=1024)
hs.set_image(1920, 1080, 1920 / 1080)
devs = [api.DeviceScene(hs.desc(0)) for _ in range(2)]


def run(dev, per_step, first_sample):
    for k in range(steps + 1):
        if k == 1:
            barrier.wait()
            times.append(time.perf_counter())
        dev.render(1 << 20, first_sample + k * per_step, first_sample + (k + 1) * per_step, **kw)
    times.append(time.perf_counter())


for what, jobs in (("one stream", [(devs[0], spp, 0)]), ("two streams", [(devs[0], spp // 2, 0), (devs[1], spp // 2, 1 << 19)])):
    for d in devs:
        d.clear_film()
    times = []
    barrier = threading.Barrier(len(jobs))
    threads = [threading.Thread(target=run, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    elapsed = max(times) - min(times)
    print(f"{what}: {steps} steps of {spp} spp in {elapsed * 1e3:.1f} ms", flush=True)
