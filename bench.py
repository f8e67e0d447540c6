#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary) and ms/frame on Sponza 1920x1080, BVH8,
diffuse + plastic materials, NEE + MIS + Russian roulette, 10 bounces, samples 0..3 (4 spp).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the whole frame = one sample per pixel:
generate -> (trace, sort, shade, shadow-trace) x bounces -> accumulate (Pathtracer::render of the
reference, Pathtracer.cpp:738-855).  The steps are submitted the way the library is meant to be
driven: the 4 samples of a frame as ONE wavefront (rt_render_samples: every launch carries 4 paths
per pixel; bit-identical to 4 single-sample calls) and up to --samples-in-flight such submissions
concurrently; --steps K renders exactly K samples.  With N > 1 the frame is split into row tiles
dealt round-robin to the ranks (each rank holds a full scene replica) and one RCCL all-gather per
completed 4-spp frame rebuilds the float4 frame on every rank: the total work is fixed, so the
scaling is STRONG.  Scene data is resident in HBM before the timed region; nothing crosses
PCIe inside it.

Rank 0 prints ONE JSON line.  `value` counts closest-hit rays (bounce 0 = primary, bounces >= 1 =
secondary) of all ranks per second of max-over-ranks wall time; shadow rays are reported
separately in `config`.  `roofline` prices the dominant kernel (kernel_trace_bvh8) in ALGORITHMIC
bytes (40 B per ray + 80 B per BVH8 node fetched + 48 B per triangle tested + 52 B
per transformed instance entry + 4 B per identity entry, the node / triangle counts measured by the
counting variant of the same kernel on the same rays) over its HIP-event time, against the 8 TB/s
HBM3E peak of MI355X.  `cpu_baseline` times the CPU oracle (a port of the reference's device
traversal, OpenMP over rays) on a bounded sample of the same rays on this box's host cores, and
the reference's own BVH builder (oracle/_ref, compiled verbatim) where it has been built.
"""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto 4 hardware queues by default; the tracer keeps (samples in flight) x 2 streams busy
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
NUM_BOUNCES = 10
SPP = 4
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3-6.6 TB/s is achievable


def build_scene(grt):
    """BASELINE config #2: Sponza, every odd diffuse material -> roughplastic alpha 0.3."""
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("sponza"))
    for i in range(1, scene.material_count, 2):
        if scene.material_type(i) == grt.MATERIAL_DIFFUSE:
            scene.set_material(i, grt.MATERIAL_PLASTIC, None, 0.3)
    grt.config_set(num_bounces=NUM_BOUNCES)
    return scene


def cpu_baseline(grt, pt, scene):
    """CPU leg (rank 0, N = 1 only): bounded sample, ~10-30 s of host work."""
    from oracle import binding as oracle  # checker only: never on the product path
    view = oracle.SceneView(pt)
    threads = os.cpu_count() or 1
    n = grt.RT_BATCH_SIZE
    o, d, _ = view.generate(0, 0, n)
    t0 = time.perf_counter()
    hits, stats_primary = view.trace(o, d, threads)
    t_primary = time.perf_counter() - t0
    # diffuse-bounce stand-in rays leaving the primary hit points (seeded)
    rng = np.random.default_rng(1234)
    t = hits[:, 2].view(np.float32)
    ok = hits[:, 1] != 0xffffffff
    org = (o + d * np.where(ok, t, 1.0).astype(np.float32) * np.float32(0.999))[:, ok]
    dirs = rng.normal(size=org.shape).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=0)
    t0 = time.perf_counter()
    _, stats_secondary = view.trace(org, dirs, threads)
    t_secondary = time.perf_counter() - t0
    rays = n + org.shape[1]
    out = {
        "value": round(rays / (t_primary + t_secondary) / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "oracle bvh8_trace (OpenMP, %d threads) over the %d primary rays of batch 0 / sample 0 plus %d seeded diffuse-bounce rays from their hit points" % (threads, n, org.shape[1]),
        "primary_mrays_s": round(n / t_primary / 1e6, 3), "secondary_mrays_s": round(org.shape[1] / t_secondary / 1e6, 3),
    }
    # same rays on the GPU through the C ABI: like-for-like ratio + algorithmic GB/s per ray class
    _, ms_p = grt.trace_rays(pt.ctx, o, d, repeat=5)
    _, ms_s = grt.trace_rays(pt.ctx, org, dirs, repeat=5)
    out["gpu_same_rays"] = {
        "primary_mrays_s": round(n / ms_p / 1e3, 1), "secondary_mrays_s": round(org.shape[1] / ms_s / 1e3, 1),
        "primary_alg_gbps": round(stats_primary.algorithmic_bytes() / (ms_p * 1e-3) / 1e9, 1),
        "secondary_alg_gbps": round(stats_secondary.algorithmic_bytes() / (ms_s * 1e-3) / 1e9, 1),
        "nodes_per_ray": [round(stats_primary.nodes / stats_primary.rays, 2), round(stats_secondary.nodes / stats_secondary.rays, 2)],
        "triangles_per_ray": [round(stats_primary.triangles / stats_primary.rays, 2), round(stats_secondary.triangles / stats_secondary.rays, 2)],
    }
    if oracle.ref_lib() is not None:  # the reference's own CPU path: BVH2 + BVH8 build of all 383 Sponza meshes
        scene.wait_until_loaded()
        t0 = time.perf_counter()
        ms2 = ms8 = 0.0
        for m in range(scene.mesh_data_count):
            ref = oracle.ref_build(scene.mesh_data_array(m, "triangles", np.float32))
            ms2 += ref["ms_bvh2"]; ms8 += ref["ms_bvh8"]
        out["reference_bvh_build"] = {"kind": "reference", "cores": 1, "ms_sah_bvh2": round(ms2, 1), "ms_bvh8_convert": round(ms8, 1),
                                      "ms_wall": round((time.perf_counter() - t0) * 1e3, 1), "meshes": scene.mesh_data_count,
                                      "product_builder_ms_parallel": round(scene.bvh_build_ms, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)   # PerfTest BUFFER_SIZE = 32 frames (Util/PerfTest.h:9)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0, help="debug: render only rank 0's tiles of an N-GPU split on one GPU (no collective), to exercise the N > 1 code path")
    ap.add_argument("--batch", type=int, default=SPP, help="samples per pixel per submission (rt_render_samples), 1..%d" % SPP)
    ap.add_argument("--samples-in-flight", type=int, default=0, help="samples per pixel rendered concurrently (rt_set_samples_in_flight)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import torch
    import torch.distributed as dist
    import gpu_raytracer_amd as grt
    import importlib
    parallel = importlib.import_module("gpu_raytracer_amd.parallel")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # debug only: BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 runs the N-rank flow (rendezvous, tile split,
    # stream hand-over, reductions) with every rank on GPU 0 and the gather staged through the host
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    scene = build_scene(grt)
    pt = grt.Pathtracer(scene, WIDTH, HEIGHT, device=local_rank)
    pt.update()
    lib = grt.device_lib()
    ctx = pt.ctx
    if args.samples_in_flight <= 0:   # measured optima (profiles/r01_sample_batching.txt): the smaller a rank's share, the more submissions
        split_n = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
        args.samples_in_flight = 3 if split_n == 1 else (4 if split_n == 2 else 8)
    grt.set_samples_in_flight(ctx, args.samples_in_flight)
    split_world = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
    split = parallel.TileSplit(rank, split_world, WIDTH, HEIGHT)
    pitch = pt.pitch
    device = torch.device("cuda", local_rank)

    # this rank's tiles, rendered as scan-order pixel ranges; the gather buffers live in torch
    packed = torch.zeros((split.local_pixels, 4), dtype=torch.float32, device=device)
    gathered = torch.zeros((split_world * split.local_pixels, 4), dtype=torch.float32, device=device)
    import ctypes
    lib.rt_pack_pixels.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_unpack_pixels.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_synchronize.argtypes = [ctypes.c_void_p]
    lib.rt_stream_wait_for_context.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rt_context_wait_for_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]

    def check(status):
        if status != 0:
            raise RuntimeError(lib.rt_last_error(ctx).decode())

    def render_step(sample_index, frame_complete=False, count=1):
        """`count` consecutive samples for this rank's share of the frame, as one submission (the
        library keeps `samples_in_flight` submissions running concurrently); with N > 1 the
        accumulated frame is gathered once it is complete, i.e. after its last sample -- every rank
        accumulates its own tiles, so nothing has to be exchanged between the samples of a frame."""
        if split_world == 1:
            check(lib.rt_set_pixel_range(ctx, 0, WIDTH * HEIGHT))
            check(lib.rt_render_samples(ctx, sample_index, count))
        else:
            check(lib.rt_set_pixel_tiles(ctx, split.tile_pixels, rank, split_world))
            check(lib.rt_render_samples(ctx, sample_index, count))
            if frame_complete and not os.environ.get("BENCH_NO_GATHER"):
                # stream-ordered, the host does not block: pack after the previous all_gather has read
                # `packed`, all_gather after the pack; the next frames are already being traced meanwhile
                torch_stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                check(lib.rt_context_wait_for_stream(ctx, torch_stream))
                check(lib.rt_pack_pixels(ctx, packed.data_ptr(), split.tile_pixels, rank, split_world, split.tiles_per_rank))
                check(lib.rt_stream_wait_for_context(ctx, torch_stream))
                if world > 1 and backend == "nccl":
                    dist.all_gather_into_tensor(gathered, packed)
                elif world > 1:
                    host = torch.empty(gathered.shape, dtype=gathered.dtype)
                    dist.all_gather_into_tensor(host, packed.cpu())
                    gathered.copy_(host)
                else:
                    gathered[:split.local_pixels].copy_(packed)   # --emulate-world: stand-in for the collective
                # ... and the gathered tiles are scattered into the final framebuffer of this rank (every rank ends
                # up with the whole frame), stream-ordered after the collective
                if world > 1:
                    check(lib.rt_context_wait_for_stream(ctx, torch_stream))
                    check(lib.rt_unpack_pixels(ctx, gathered.data_ptr(), split.tile_pixels, split_world, split.tiles_per_rank))

    def counters():
        c = pt.counters()
        return c, sum(c.trace[:NUM_BOUNCES]), sum(c.shadow[:NUM_BOUNCES])

    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]

    # ---- warm-up (untimed) ---------------------------------------------------------------------
    def submissions(steps):
        """[(first sample index, count)] covering exactly `steps` samples, frame by frame."""
        out, k = [], 0
        while k < steps:
            first = k % SPP
            count = min(SPP - first, steps - k, args.batch)
            out.append((first, count)); k += count
        return out

    # untimed priming: every submission slot allocates its queues / streams on first use (GBs of hipMalloc)
    for _ in range(args.samples_in_flight):
        render_step(0, frame_complete=True, count=min(args.batch, SPP))
    check(lib.rt_synchronize(ctx))
    for first, count in submissions(args.warmup):
        render_step(first, frame_complete=True, count=count)
    check(lib.rt_synchronize(ctx))

    # ---- untimed statistics pass: rays per sample and the work counters of the trace kernels ---
    rays_per_sample, shadow_per_sample, alg_bytes_per_sample, trace_rays_stat = [], [], [], []
    grt.set_trace_statistics(ctx, True)
    for s in range(SPP):
        render_step(s)
        _, closest, shadow = counters()
        stats = grt.get_trace_statistics(ctx)
        rays_per_sample.append(closest); shadow_per_sample.append(shadow)
        alg_bytes_per_sample.append(stats["closest"]["algorithmic_bytes"])
        trace_rays_stat.append(stats)
    grt.set_trace_statistics(ctx, False)

    # ---- profiled pass (HIP events per stage, on the tracer's stream): kernel time of the trace launches
    grt.set_profiling(ctx, True)
    trace_ms, stage_ms = [], {}
    alone_subs = submissions(SPP)
    for rep in range(2):
        for first, count in alone_subs:
            render_step(first, count=count)
            c, _, _ = counters()
            if rep == 1:
                trace_ms.append(c.ms_trace)
                for k in ("ms_generate", "ms_trace", "ms_sort", "ms_shade", "ms_shadow", "ms_post"):
                    stage_ms[k] = stage_ms.get(k, 0.0) + getattr(c, k) / SPP
    grt.set_profiling(ctx, False)

    # ---- timed region: exactly K steps -------------------------------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    check(lib.rt_synchronize(ctx))
    # HIP events around every trace launch of the timed region, each on the stream the launch runs on
    # (mode 2: no serialisation -- the side stream and the samples in flight stay as in production)
    grt.set_profiling(ctx, 0 if os.environ.get("BENCH_NO_LAUNCH_TIMING") else 2)
    t0 = time.perf_counter()
    plan = submissions(args.steps)
    for i, (first, count) in enumerate(plan):
        render_step(first, frame_complete=(first + count == SPP or i == len(plan) - 1), count=count)
    check(lib.rt_synchronize(ctx))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timed_counters = pt.counters()
    timed_subs = [plan[i] for i in range(0, len(plan), 3)]       # mode 2 times every 3rd submission (events are not free)
    timed_trace_ms = timed_counters.ms_trace                     # sum over their closest-hit launches
    timed_alg_bytes = sum(alg_bytes_per_sample[first + j] for first, count in timed_subs for j in range(count))
    grt.set_profiling(ctx, False)

    local = torch.tensor([elapsed, float(sum(rays_per_sample)), float(sum(shadow_per_sample)), float(sum(alg_bytes_per_sample)), float(sum(trace_ms))], dtype=torch.float64, device=device)
    if world > 1:
        mx = local.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = local.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0]); rays_4spp = float(sm[1]); shadow_4spp = float(sm[2])
    else:
        rays_4spp, shadow_4spp = float(local[1]), float(local[2])

    if rank == 0:
        rays_per_step = rays_4spp / SPP
        total_rays = rays_per_step * args.steps
        value = total_rays / elapsed / 1e6
        launches_per_sample = NUM_BOUNCES  # closest-hit trace launches per submission (one per bounce)
        achieved = timed_alg_bytes / (max(timed_trace_ms, 1e-9) * 1e-3) / 1e9  # rank 0's launches of the timed region
        achieved_alone = sum(alg_bytes_per_sample) / (sum(trace_ms) * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "kernel": "kernel_trace_bvh8", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": round(timed_alg_bytes / len(timed_subs) / launches_per_sample),
            "samples_per_launch": args.batch,
            "avg_launch_ms": round(timed_trace_ms / len(timed_subs) / launches_per_sample, 4), "launches_per_step": round(launches_per_sample * len(plan) / args.steps, 2),
            "avg_launch_ms_running_alone": round(sum(trace_ms) / len(alone_subs) / launches_per_sample, 4), "frac_running_alone": round(achieved_alone / HBM_PEAK_GBPS, 4),
            "bytes_per_ray": round(sum(alg_bytes_per_sample) / max(sum(rays_per_sample), 1), 1),
            "nodes_per_ray": round(sum(s["closest"]["nodes"] for s in trace_rays_stat) / max(sum(rays_per_sample), 1), 2),
            "triangles_per_ray": round(sum(s["closest"]["triangles"] for s in trace_rays_stat) / max(sum(rays_per_sample), 1), 2),
            "measured_stream_read_gbps": round(grt.measure_stream_bandwidth(ctx, 1 << 30, 5), 1),
            "note": "achieved = algorithmic bytes of the timed region's trace launches / sum of their HIP-event durations while they share the GPU with the shadow launch and the other sample in flight (what rocprofv3 --kernel-trace sees); *_running_alone = the same launches in a serialised pass. Working set (2.6 MB nodes + 12.6 MB triangle positions) is L2/Infinity-Cache resident; algorithmic bytes >> DRAM traffic",
        }
        traffic_file = os.path.join(ROOT, "profiles", "pmc_trace_traffic.json")
        if os.path.exists(traffic_file):
            roofline["traffic"] = json.load(open(traffic_file)).get("hbm_bytes_per_launch")
        result = {
            "metric": "Mrays/s (primary+secondary) + ms/frame, Sponza 1920x1080 4spp BVH8", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "Sponza (Crytek, 262 687 triangles, 384 instances) 1920x1080, samples 0..3 (4 spp), BVH8/CWBVH, diffuse + roughplastic(odd materials, alpha 0.3), NEE+MIS+RR, 10 bounces, constant white sky, mipmapping on, 19 diffuse textures at the reference's dimensions (1024x1024 RGBA8 + mips, ~106 MB; texels replicated 4x4 from the quarter-size maps that travel with the repo), the 5 maps missing upstream are the reference's 1x1 fallback texel",
                "step": "one sample per pixel for the whole frame; the 4 samples of a frame are submitted as one wavefront (rt_render_samples), the reference's 777 600-pixel batches (a VRAM bound) are not needed",
                "rays_per_step": round(rays_per_step), "shadow_rays_per_step": round(shadow_4spp / SPP),
                "mrays_s_including_shadow": round((rays_4spp + shadow_4spp) / SPP * args.steps / elapsed / 1e6, 1),
                "ms_per_4spp_frame": round(elapsed / args.steps * SPP * 1e3, 3),
                "emulated_world": args.emulate_world, "ranks": (dist.get_world_size() if world > 1 else 1),
                "parallelism": "tile-split x%d + one RCCL all-gather of the accumulated float4 frame per %d-spp frame" % (world, SPP) if world > 1 else "single GPU",
                "samples_per_submission": args.batch, "submissions_in_flight": args.samples_in_flight,
                "stage_ms_per_step": {k: round(v, 3) for k, v in stage_ms.items()},
            },
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(grt, pt, scene)
        print(json.dumps(result))

    pt.close()
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
