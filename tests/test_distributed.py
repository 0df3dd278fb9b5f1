"""Several GPUs: the sample range is split over one process per GPU and the float4 films (sum r, g, b, sample
count) are summed onto rank 0 with one reduce.  Here the split and the reduce run under gloo with two and three
CPU processes; each rank's film comes from the CPU checker (there is no GPU in this suite), which renders
exactly the samples the rank's GPU would."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sample_ranges_partition_the_samples():
    from pathtracer_amd.path_tracer import sample_range
    for spp in (1, 2, 7, 16, 1024):
        for world in (1, 2, 3, 4, 8):
            ranges = [sample_range(spp, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == spp
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _worker(rank, world, port, spp, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import checker
    from pathtracer_amd.path_tracer import sample_range
    from pathtracer_amd.scene import SceneData

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = SceneData.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz")).with_image(24, 24)
    orc = checker.OracleScene(scene.to_desc())
    begin, end = sample_range(spp, rank, world)
    sums, _ = orc.li_image(spp, begin, end, seed=21, threads=2)
    film = np.concatenate([sums, np.full((24, 24, 1), end - begin, np.float32)], axis=2)   # the float4 film
    t = torch.from_numpy(film.copy())
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,spp", [(2, 8), (3, 7)])
def test_film_reduce_over_gloo_equals_single_process(world, spp, tmp_path):
    import torch.multiprocessing as mp
    from oracle import checker
    from pathtracer_amd.scene import SceneData

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(world, port, spp, out), nprocs=world, join=True)
    film = np.load(out)
    assert (film[..., 3] == spp).all()                       # every sample was rendered exactly once
    scene = SceneData.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz")).with_image(24, 24)
    whole, _ = checker.OracleScene(scene.to_desc()).li_image(spp, seed=21)
    # per-rank partial sums added in rank order instead of one running sum: equal up to float re-association
    assert np.allclose(film[..., :3], whole, rtol=1e-5, atol=1e-6)
    mean = film[..., :3] / film[..., 3:]
    assert np.abs(mean - whole / np.float32(spp)).max() <= 1e-4
