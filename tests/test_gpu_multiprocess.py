"""The product's several-GPU path on the one GPU a test box has.

* PathTracer.Render under torch.distributed with 2 and 3 processes, all on GPU 0 (gloo: RCCL refuses two ranks on one
  device): every rank renders its sample_range of the frame through the C-ABI, reduce_film sums the float4 films onto
  rank 0, rank 0 reads the picture back.  Against one process: the film's sample count is spp in every pixel and the
  picture is equal up to the re-association of the per-rank partial sums.
* bench.py as the driver launches it for N > 1 (python -m torch.distributed.run), rehearsed on GPU 0, in both
  scaling modes: one JSON line from rank 0, every sample of the frame rendered once.
* reduce_film over RCCL with the one rank a one-GPU box allows: the in-place reduce on the library's device buffer.
* pt_merge_films, what a single host process with several GPUs uses instead (pathTracer --devices): films of several
  scenes added on the device.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE, SPP, SEED = 48, 9, 13


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pathtracer_amd.path_tracer import PathTracer, Scene, sample_range
    from pathtracer_amd.scene import SceneData

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = SceneData.load(os.path.join(ROOT, "tests", "golden", "mixed_scene.npz")).with_image(SIZE, SIZE)
    scene = Scene(data, spp_list=[SPP])
    tracer = PathTracer(device=0, seed=SEED)
    tracer.Render(scene, 0)
    begin, end = sample_range(SPP, rank, world)
    assert tracer.stats.samples == SIZE * SIZE * (end - begin)          # this rank's share, nothing else
    if rank == 0:
        film = scene.device_scene(0).download_film()
        np.savez(out_path, linear=tracer.linearImage, tonemapped=tracer.renderedImage, film=film)
    else:
        assert tracer.renderedImage is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_render_split_over_processes_on_one_gpu(world, tmp_path, scenes):
    import torch.multiprocessing as mp
    from pathtracer_amd import api

    out = str(tmp_path / "rank0.npz")
    mp.spawn(_rank, args=(world, free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    assert (got["film"][..., 3] == SPP).all()                            # every sample exactly once
    with api.DeviceScene(scenes["mixed"].with_image(SIZE, SIZE).to_desc()) as dev:
        dev.render(SPP, seed=SEED, ordered=True)
        linear, tonemapped, _ = dev.readback()
    assert np.allclose(got["linear"], linear, rtol=2e-6, atol=1e-7)      # partial sums per rank, then added
    assert np.abs(got["tonemapped"] - tonemapped).max() <= 1e-5
    assert linear.max() > 0.05


def _nccl_single_rank(rank, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from pathtracer_amd import api
    from pathtracer_amd.path_tracer import film_tensor, reduce_film
    from pathtracer_amd.scene import SceneData

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    data = SceneData.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz")).with_image(SIZE, SIZE)
    with api.DeviceScene(data.to_desc()) as dev:
        dev.render(SPP, seed=SEED)
        before = dev.download_film()
        assert film_tensor(dev, 0).is_cuda
        reduce_film(dev, 0, dist)                       # RCCL reduce, in place on the library's own device buffer
        after = dev.download_film()
        linear, _, _ = dev.readback()
    np.savez(out_path, before=before, after=after, linear=linear)
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_film_over_rccl_in_place(tmp_path):
    """reduce_film's nccl branch - one RCCL reduce on the film's device memory, wrapped as a tensor without a copy - with
    the one rank a one-GPU box allows: the film comes back as it was, and the picture can still be read."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "nccl.npz")
    mp.spawn(_nccl_single_rank, args=(free_port(), out), nprocs=1, join=True)
    got = np.load(out)
    assert_bits_equal(got["after"], got["before"], "film after a one-rank reduce")
    assert (got["after"][..., 3] == SPP).all() and got["linear"].max() > 0.05


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launched_with_two_ranks(scaling):
    width, height, spp, steps, warmup = 64, 36, 4, 2, 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup),
           "--rehearse-on-one-gpu", "--width", str(width), "--height", str(height), "--triangles", "3000", "--texture-size", "16",
           "--spp-per-step", str(spp), "--scaling", scaling, "--cpu-baseline-spp", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                    # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["scaling"] == scaling and line["value"] > 0
    frame_spp = spp * steps * 2        # either way a rank renders `spp` samples per step; strong: as shares of one frame
    assert line["config"]["spp_rendered"] == frame_spp
    assert round(line["msamples_per_s"] * line["ms_per_step"] * steps * 1e3) == pytest.approx(width * height * frame_spp, rel=1e-3)


def test_merge_films_on_the_device(scenes):
    """Three scenes on GPU 0 render samples 0-3, 4-6 and 7-8 of nine; pt_merge_films adds the second and third film
    to the first.  Equal to adding the downloaded films in that order on the host, bit for bit, and to one
    nine-sample render up to re-association."""
    from pathtracer_amd import abi, api
    desc = scenes["cornell"].with_image(40, 24).to_desc()
    with api.DeviceScene(desc) as a, api.DeviceScene(desc) as b, api.DeviceScene(desc) as c:
        for dev, (begin, end) in zip((a, b, c), ((0, 4), (4, 7), (7, 9))):
            dev.render(9, begin, end, seed=5, ordered=True)
        want = (a.download_film() + b.download_film()) + c.download_film()
        first = a.download_film()
        a.merge_films([b, c])
        assert_bits_equal(a.download_film(), want, "films added on the device")
        assert (want[..., 3] == 9).all()
        # the path of GPUs that cannot map each other's memory: every source through a copy on the destination's GPU
        a.upload_film(first)
        a.merge_films([b, c], stage_always=True)
        assert_bits_equal(a.download_film(), want, "films added through the staging copy")
        assert_bits_equal((first + b.download_film()) + c.download_film(), want, "sources untouched")
        merged, _, _ = a.readback()
        a.render(9, seed=5, ordered=True)
        whole, _, _ = a.readback()
        assert np.allclose(merged, whole, rtol=2e-6, atol=1e-7)
        with api.DeviceScene(scenes["cornell"].with_image(16, 16).to_desc()) as small:
            with pytest.raises(api.PtError) as e:
                a.merge_films([small])
            assert e.value.code == abi.PT_ERR_INVALID_ARG
