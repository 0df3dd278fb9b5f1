#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and Msamples/s of the path-tracing hot path on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Workload (the configuration the headline metric is quoted on): the Sponza scene of
resources/scenes/sponza.json at 1920x1080, depth 5.  sponza.obj and eight of its textures are not part of the
reference checkout, so the scene is the procedural atrium stand-in of pathtracer_amd/host/generators.cpp with
the same triangle count (262,267), camera, light and 24 textured 1024^2 materials; `data` says so.

One step = one pass of the hot path over one batch: --spp-per-step samples of every pixel (default 128, all in
flight at once: 265 M paths, 48 GB of path state), i.e. 1920 * 1080 * 128 = 265.4 M pixel samples per GPU and
step.  --scaling weak (default): every rank renders its own --spp-per-step samples per step.  --scaling strong: step k
is samples [k * S * N, (k + 1) * S * N) of ONE frame, S = --spp-per-step, N = ranks, cut into contiguous shares
(pathtracer_amd.path_tracer.sample_range), so the ranks together render one picture of S * N * steps samples per pixel;
a rank still renders S samples per pass, because a small pass pays the drain tail of every traversal launch.  The JSON
line says which of the two it measured.  After the last step the per-GPU films are summed onto rank 0 with one RCCL
reduce, inside the timed region.  The default --steps 8 on one GPU is the full 1024 spp frame.

The JSON line carries `roofline` for the dominant kernel (closest-hit traversal or shadow-ray traversal, whichever
took longer): the bytes that left the L2 per launch - from the committed counter summary of this same command under
profiles/, used only while it was collected with the same sources, traversal mode and workload - over the launch's time,
against the 8 TB/s HBM peak.  The scene's records (a few tens of MB) live in the caches, so that fraction is small: memory
does not bound the kernel, and `ceilings` lists the units that do (vector issue of the SIMDs, texture addressers, scalar
units; `binding_unit` names the busiest).  Beside it, measured live: the kernel's average launch time from HIP events on
the launch stream, its algorithmic bytes (the reference's per-ray byte model times the kernel's own counters) and the
compulsory bytes (scene records once, rays in, hits out).  `cpu_baseline`: the reference's own compiled hot path
(oracle/_ref/libptref.so) or, without it, the CPU restatement (oracle/pt_oracle.c), both OpenMP, timed on this host over
a bounded sample of the same frame, before the GPU's warm-up.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
# everything that decides what the render kernels execute and what they read: the kernels, and the code that lays the scene out for them
KERNEL_SOURCES = ("pathtracer_amd/csrc/kernels.hip", "pathtracer_amd/csrc/api.hip", "pathtracer_amd/csrc/wide_bvh.hip", "pathtracer_amd/csrc/wide_bvh.h",
                  "pathtracer_amd/csrc/layout_build.hip", "pathtracer_amd/csrc/layout_build.h",
                  "pathtracer_amd/csrc/pt_device.h", "pathtracer_amd/csrc/pt_sincosf.h", "pathtracer_amd/csrc/pt_powf.h", "pathtracer_amd/csrc/pt_acosf.h",
                  "include/pt_api.h")


def kernel_source_hash():
    """Identifies the kernels a counter summary under profiles/ was collected with."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--spp-per-step", type=int, default=128)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--triangles", type=int, default=262267)
    p.add_argument("--texture-size", type=int, default=1024)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--traversal", choices=("wide", "ordered", "reference"), default="wide",
                   help="wide (default): the tree collapsed to eight children per node with quantised boxes, for closest hits and "
                        "shadow rays; ordered: front to back over the binary tree; reference: the reference's first-child-first "
                        "order (counters equal to the CPU's).  The same hits and film from each")
    p.add_argument("--no-wide-shadows", action="store_true", help="with --traversal wide: shadow rays as under --traversal ordered")
    p.add_argument("--scene-flags", type=int, default=0, help="PT_SCENE_* flags for pt_upload_scene (measurement switches)")
    p.add_argument("--cpu-baseline-spp", type=int, default=64, help="most samples per pixel the CPU baseline renders (0 = skip)")
    p.add_argument("--cpu-baseline-seconds", type=float, default=20.0, help="CPU time the baseline sample is sized for")
    p.add_argument("--save-image", default="", help="write the tonemapped frame of rank 0 to this file")
    p.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                   help="weak: --spp-per-step samples per step and GPU; strong: --spp-per-step samples per step shared by the GPUs")
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="run all ranks on GPU 0 with the gloo backend: a dry run of the multi-process path on a one-GPU box "
                        "(the numbers mean nothing)")
    return p.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    import torch
    from pathtracer_amd import abi, api, host, path_tracer
    from pathtracer_amd.scene import _from_ptr

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback for the hot path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))

    # tensors of the small collectives: on the GPU for RCCL, in host memory for the gloo rehearsal
    coll_device = "cpu" if args.rehearse_on_one_gpu else f"cuda:{local_rank}"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(local_rank)

    # ---- scene (host side, outside the timed region) and upload
    t0 = time.time()
    hs = host.HostScene.atrium(seed=1, triangles=args.triangles, texture_size=args.texture_size)
    hs.set_image(args.width, args.height, args.width / args.height)
    desc = hs.desc(local_rank)
    desc.flags |= args.scene_flags
    dev = api.DeviceScene(desc)
    setup_s = time.time() - t0

    spp_step = args.spp_per_step
    strong = args.scaling == "strong"
    # Either way a rank renders spp_step samples per pixel in a step, as one pass: small passes pay the drain tail of every
    # traversal launch.  weak: rank r's samples of step k are its own block of the frame; strong: step k is samples
    # [k * S * N, (k + 1) * S * N) of the frame, S = spp_step, N = ranks, cut into N contiguous shares.
    spp_total = spp_step * world * (args.steps + args.warmup)
    wide = args.traversal == "wide"
    wide_shadows = wide and not args.no_wide_shadows
    ordered = args.traversal == "ordered" or wide

    def run_step(step, profile):
        if strong:   # this rank's share of the step's samples of the frame
            begin, end = path_tracer.sample_range(spp_step * world, rank, world)
            first, last = step * spp_step * world + begin, step * spp_step * world + end
        else:
            first = (step * world + rank) * spp_step
            last = first + spp_step
        if last > first:
            dev.render(spp_total, first, last, seed=args.seed, accumulate=True, profile=profile, ordered=ordered, wide=wide, wide_shadows=wide_shadows)

    # CPU baseline beside the GPU number, rank 0 at N = 1 only, before the GPU's warm-up: the reference's own compiled
    # Li / BVH / shape code (oracle/_ref/libptref.so, built from /root/reference in the build container) when that
    # library is here, else this repository's C restatement; on a bounded sample of the same frame sized for ~20 s
    cpu = None
    if args.cpu_baseline_spp > 0 and world == 1:
        from oracle import checker, ref
        threads = checker.max_threads()
        if ref.available():
            cpu_scene, kind, what = ref.RefScene(desc), "reference", "the reference's sources (oracle/Makefile.ref) under oracle/ref_harness.cpp, OpenMP"
        else:
            cpu_scene, kind, what = checker.OracleScene(desc), "port", "oracle/pt_oracle.c with OpenMP"
        t1 = time.perf_counter()
        _, probe = cpu_scene.li_image(spp_total, 0, 1, seed=args.seed)
        probe_s = time.perf_counter() - t1
        n_spp = max(1, min(args.cpu_baseline_spp, int(args.cpu_baseline_seconds / max(probe_s, 1e-3))))
        t1 = time.perf_counter()
        _, cstats = cpu_scene.li_image(spp_total, 1, 1 + n_spp, seed=args.seed)
        cpu_s = time.perf_counter() - t1
        cpu_rays = cstats["rays_closest"] + cstats["rays_shadow"]
        cpu_samples = args.width * args.height * n_spp
        cpu = {"value": round(cpu_rays / cpu_s / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": kind,
               "msamples_per_s": round(cpu_samples / cpu_s / 1e6, 4), "seconds": round(cpu_s, 2),
               "sample": f"samples 1..{n_spp} of every pixel of the same {args.width}x{args.height} frame "
                         f"({cpu_samples} pixel samples, {cpu_rays} rays incl. zero-contribution shadow rays), {what}"}
        del cpu_scene

    dev.clear_film()
    for w in range(args.warmup):
        run_step(w, False)
    warm = dev.stats()
    if dist is not None:
        # the collective's first use sets up RCCL's channels and buffers: do that on a scratch tensor of the
        # film's shape, outside the timed region like the warm-up steps
        scratch = torch.zeros_like(path_tracer.film_tensor(dev, local_rank), device=coll_device)
        dist.reduce(scratch, dst=0, op=dist.ReduceOp.SUM)
        del scratch

    barrier()
    t_start = time.perf_counter()
    for k in range(args.steps):
        run_step(args.warmup + k, True)
    if dist is not None:
        path_tracer.reduce_film(dev, local_rank, dist)
    barrier()
    elapsed = time.perf_counter() - t_start

    if dist is not None:
        t = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = dev.stats()
    counters = ["samples", "rays_closest", "rays_shadow", "nodes_closest", "nodes_shadow", "tri_tests_closest",
                "tri_tests_shadow", "sphere_tests_closest", "sphere_tests_shadow", "shaded_hits", "node_steps_closest",
                "node_steps_shadow"]
    mine = {c: getattr(stats, c) - getattr(warm, c) for c in counters}   # timed steps only
    if dist is not None:
        t = torch.tensor([mine[c] for c in counters], device=coll_device, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = dict(zip(counters, [int(x) for x in t.tolist()]))
    else:
        total = mine

    if rank == 0:
        rays = total["rays_closest"] + total["rays_shadow"]
        # dominant traversal kernel on this rank, timed with HIP events around every launch (PT_RENDER_PROFILE)
        ms = {abi.KERNEL_NAMES[i]: stats.kernel_ms[i] for i in range(abi.PT_K_COUNT)}
        launches = {abi.KERNEL_NAMES[i]: int(stats.launches[i] - warm.launches[i]) for i in range(abi.PT_K_COUNT)}
        # Algorithmic bytes, per kernel: what the traversal asks of memory by the reference's per-ray byte model (a node
        # visit reads the node, a primitive test the primitive's vertices, a ray comes in and a hit goes out), with the
        # record sizes of the layout the mode reads.  Binary tree: 32 B per node (reference order), 36 B front to back
        # (the ray's link), 48 B per triangle test.  Eight-wide layout: 80 B per node, 64 B per primitive record (its
        # vertices and the box of its leaf).  Light-space records of a directional light's shadow rays: 16 B.
        one_directional = desc.num_lights == 1 and int(_from_ptr(desc.lights, 1, abi.LIGHT_DTYPE)["type"][0]) == abi.PT_LIGHT_DIRECTIONAL
        light_space = ordered and not wide_shadows and one_directional and not (desc.flags & abi.PT_SCENE_NO_PARALLEL_SHADOWS)
        node_bytes, prim_bytes = (80, 64) if wide else ((36, 48) if ordered else (32, 48))
        bytes_closest = node_bytes * mine["nodes_closest"] + prim_bytes * mine["tri_tests_closest"] + 84 * mine["sphere_tests_closest"] + 48 * mine["rays_closest"]
        shadow_node_bytes, shadow_prim_bytes = (80, 64) if wide_shadows else ((16, 48) if light_space else (32, 48))
        bytes_shadow = shadow_node_bytes * mine["nodes_shadow"] + shadow_prim_bytes * mine["tri_tests_shadow"] + 16 * mine["sphere_tests_shadow"] + 36 * mine["rays_shadow"]
        # records of the scene a kernel reads at least once (device sizes), beside the rays in and the results out
        binary_scene = int(desc.num_nodes) * (64 if ordered else 32) + int(desc.num_prims) * 64
        wide_scene = int(desc.num_nodes) * 25 + int(desc.num_prims) * 64    # about 0.3 wide nodes of 80 B per node of the tree
        if ms["extend"] >= ms["connect"]:
            kernel, kbytes, kms, klaunch, krays = ("ExtendWideKernel" if wide else "ExtendKernel"), bytes_closest, ms["extend"], launches["extend"], mine["rays_closest"]
            knodes, ksteps = mine["nodes_closest"], mine["node_steps_closest"]
            if wide:
                instance = lambda name: name.startswith("ExtendWideKernel<")
            else:
                instance = lambda name: name.startswith("ExtendKernel<") and name.rstrip(">").split(",")[-1].strip() == ("true" if ordered else "false")
            compulsory = (wide_scene if wide else binary_scene) * klaunch + 48 * krays
        else:
            kernel, kbytes, kms, klaunch, krays = ("ConnectWideKernel" if wide_shadows else "ConnectKernel"), bytes_shadow, ms["connect"], launches["connect"], mine["rays_shadow"]
            knodes, ksteps = mine["nodes_shadow"], mine["node_steps_shadow"]
            prefix = "ConnectWideKernel<" if wide_shadows else ("ConnectParallelKernel<" if light_space else "ConnectKernel<")
            instance = lambda name: name.startswith(prefix)
            compulsory = (wide_scene if wide_shadows else int(desc.num_nodes) * 32 + int(desc.num_prims) * 64) * klaunch + 36 * krays
        launch_s = kms * 1e-3 / max(1, klaunch)
        algorithmic_gbs = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        compulsory_gbs = compulsory / (kms * 1e-3) / 1e9 if kms > 0 else 0.0

        # Hardware counters of that kernel from the committed rocprofv3 summary of this same command
        # (profiles/collect_pmc.sh): used only if it was collected with the same sources (kernels and scene layout code),
        # traversal mode and workload, so that a stale summary never sits beside fresh timings.
        mode = args.traversal + ("" if wide_shadows or not wide else "-no-wide-shadows")
        profile, profile_name, why_not = None, None, "no counter summary under profiles/"
        default_workload = (args.spp_per_step, args.width, args.height, args.triangles, args.texture_size) == (128, 1920, 1080, 262267, 1024)
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
            try:
                summary = json.load(open(path))
            except (OSError, ValueError):
                continue
            meta = summary.get("_meta", {})
            if not default_workload:
                why_not = "the committed counter summaries are of the default workload"
                break
            if meta.get("traversal") != mode:   # the summary of another mode: look on
                if profile is None and why_not.startswith("no counter"):
                    why_not = f"no counter summary of traversal mode {mode} under profiles/"
                continue
            if meta.get("kernel_sources") != kernel_source_hash():
                why_not = f"{os.path.relpath(path, ROOT)} was collected with other kernel sources"
                continue
            name = next((k for k in sorted(summary) if instance(k)), None)
            if name and "per_dispatch" in summary[name]:
                profile, profile_name, kernel = summary[name], os.path.relpath(path, ROOT), name
                break
        hbm = {"peak": HBM_PEAK_GBS, "unit": "GB/s",
               "algorithmic": round(algorithmic_gbs, 1), "algorithmic_frac": round(algorithmic_gbs / HBM_PEAK_GBS, 4),
               "compulsory": round(compulsory_gbs, 1), "compulsory_frac": round(compulsory_gbs / HBM_PEAK_GBS, 4),
               "note": "fabric = bytes that left the L2 per launch, Infinity Cache hits included (FETCH_SIZE as it is for these "
                       "kernels' gathers of 64-byte lines - calibrated in profiles/r03_fetch_size_calibration.txt - plus WRITE_SIZE); "
                       "algorithmic = the reference's per-ray byte model times the kernel's own counters (cache hits included, "
                       "can exceed the HBM peak); compulsory = scene records once per launch + rays in + results out"}
        traffic = None
        if profile:
            per = profile["per_dispatch"]
            traffic = round(profile.get("fabric_read_bytes", 0.0) + profile.get("fabric_write_bytes", 0.0))
            cycles = per.get("GRBM_GUI_ACTIVE", 0.0) / 8                      # the counter sums the 8 XCDs
            prof_s = profile.get("trace", {}).get("avg_ns", 0.0) * 1e-9 or launch_s
            busy = per.get("SQ_ACTIVE_INST_VALU", 0.0) * 4                    # SIMD cycles spent issuing vector instructions
            hbm.update({"fabric": round(traffic / prof_s / 1e9, 1), "fabric_frac": round(traffic / prof_s / 1e9 / HBM_PEAK_GBS, 4)})
            # The roofline this tier asks for is the memory one: bytes that left the L2 per launch over the launch's time,
            # against the HBM peak.  The scene's records sit in the caches, so that fraction is small and says that memory
            # does not bound the kernel; `ceilings` says what does - each unit as busy cycles over the cycles it had: the
            # SIMDs' vector issue (1,024 SIMDs; SQ_ACTIVE_INST_VALU at four cycles an instruction, which is what one wave's
            # vector instruction occupies a SIMD for - with several waves resident a SIMD gets plain fp32 instructions
            # through in about 2.7, profiles/microbench/valu_rate.hip, so that the figure can pass 1: read that as
            # saturated), the texture addressers through which every vector memory instruction passes (one per CU;
            # TA_TA_BUSY), the scalar units (one per CU).
            ta_busy = per.get("TA_TA_BUSY_sum", 0.0)
            ceilings = {"valu_issue": (busy, 1024 * cycles), "texture_addresser": (ta_busy, 256 * cycles)}
            if per.get("SQ_INSTS_SALU"):   # one scalar unit per CU, at most one instruction per cycle
                ceilings["scalar_issue"] = (per["SQ_INSTS_SALU"], 256 * cycles)
            fracs = {k: round(a / b, 4) if b else 0.0 for k, (a, b) in ceilings.items()}
            fracs["hbm"] = hbm["fabric_frac"]
            roofline = {"bound": "hbm", "kernel": kernel, "achieved": hbm["fabric"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fracs["hbm"],
                        "binding_unit": max(fracs, key=fracs.get), "ceilings": fracs}
            roofline.update({"counters_from": profile_name, "counters_kernel_sources": kernel_source_hash(),
                             "counters_avg_launch_ms": round(prof_s * 1e3, 4),
                             "counters": {k: round(profile[k], 4) for k in ("l1_hit_rate", "l2_hit_rate", "wave_time_waiting_on_memory",
                                                                             "wave_time_issue_stalled") if k in profile}})
        else:
            # without matching counters only the compulsory bytes are known to have moved
            roofline = {"bound": "hbm", "kernel": kernel, "achieved": round(compulsory_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(compulsory_gbs / HBM_PEAK_GBS, 4), "counters_from": None, "counters_missing_because": why_not}
        roofline.update({"traffic": traffic, "hbm": hbm, "traversal": mode,
                         "bytes_per_launch": round(kbytes / max(1, klaunch)), "avg_launch_ms": round(kms / max(1, klaunch), 4),
                         "launches": klaunch, "bytes_per_ray": round(kbytes / max(1, krays), 1),
                         "grays_per_s_in_kernel": round(krays / (kms * 1e-3) / 1e9, 3) if kms > 0 else 0.0,
                         "nodes_per_ray": round(knodes / max(1, krays), 2),
                         "lanes_per_node_step": round(knodes / max(1, ksteps), 2)})

        # what a plain device-to-device copy reaches on this GPU (read + write bytes), beside the nominal peak
        if world == 1:
            n = 1 << 30
            src = torch.empty(n, dtype=torch.uint8, device=f"cuda:{local_rank}").fill_(1)
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize(local_rank)
            t1 = time.perf_counter()
            for _ in range(10):
                dst.copy_(src)
            torch.cuda.synchronize(local_rank)
            hbm["copy_measured"] = round(10 * 2 * n / (time.perf_counter() - t1) / 1e9, 1)
            del src, dst

        if args.save_image:
            _, tm, _ = dev.readback(abi.PT_TONEMAP_UNCHARTED2, linear=False)
            host.write_image(args.save_image, tm)

        out = {
            "metric": "Mrays/s (closest-hit + shadow rays traced, whole job), sponza-class scene 1080p depth 5",
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "msamples_per_s": round(total["samples"] / elapsed / 1e6, 3),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"sponza.json stand-in (procedural atrium, {args.triangles} triangles, 24 textured materials "
                                   f"{args.texture_size}^2, camera/light of resources/scenes/sponza.json) {args.width}x{args.height}, "
                                   f"depth 5, {spp_step} spp per step and GPU",
                       "traversal": mode,
                       "spp_rendered": spp_step * world * args.steps,
                       "sample_split": (f"strong: every step is {spp_step * world} spp of one frame, shared by {world} GPU(s)" if strong
                                        else f"weak: {world} x {spp_step} spp per step, each GPU its own samples"),
                       "rays_per_sample": round(rays / max(1, total["samples"]), 3),
                       "nodes_per_ray": round((total["nodes_closest"] + total["nodes_shadow"]) / max(1, rays), 2),
                       "tri_tests_per_ray": round((total["tri_tests_closest"] + total["tri_tests_shadow"]) / max(1, rays), 3)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernel_ms": {k: round(v, 3) for k, v in ms.items()},
            "setup_s": round(setup_s, 2),
        }
        print(json.dumps(out), flush=True)

    dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
