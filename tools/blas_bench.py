#!/usr/bin/env python3
"""Device BLAS build (config device_blas = 1, rt_build_geometry) against the host build on the benchmark scene:
build time, node count, and what the trees cost to traverse (the benchmark's frame loop, ms per step).
usage (GPU box): python tools/blas_bench.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402


def main():
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    presplits = [float(v) for v in os.environ.get("DEVICE_PRESPLIT", "0.05").split(",")]   # early split clipping in front of the device build (config device_presplit; 0: off)
    # (device trees twice: as the device's collapse leaves them, and seated by the slot learner in the integrator's first update -- round 6)
    for device_blas, presplit, seating in [(0, 0.0, 1)] + [(1, v, s) for v in presplits for s in (0, 1)]:
        scene = bench.build_scene(grt)
        grt.config_set(device_blas=device_blas, device_presplit=presplit, merge_static=int(os.environ.get("MERGE_STATIC", "1")))   # (MERGE_STATIC=0: one tree per mesh under the TLAS, the reference's layout)
        if not seating:
            grt.config_set(static_slot_learning_rays=0)
        t0 = time.perf_counter()
        pt = grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0); pt.update()
        setup_s = time.perf_counter() - t0
        nodes = pt.array("bvh8_nodes").view(np.uint8).reshape(-1, 80).shape[0]
        ctx = pt.ctx
        for _ in range(3):
            lib.rt_render_samples(ctx, 0, 4)
        lib.rt_synchronize(ctx)
        steps = 32
        t0 = time.perf_counter()
        for _ in range(steps // 4):
            lib.rt_render_samples(ctx, 0, 4)
        lib.rt_synchronize(ctx)
        ms = (time.perf_counter() - t0) / steps * 1e3
        grt.set_trace_statistics(ctx, True)
        lib.rt_render_samples(ctx, 0, 1); stats = grt.get_trace_statistics(ctx); c = pt.counters()
        rays = sum(c.trace[:bench.NUM_BOUNCES])
        print("device_blas=%d presplit=%.3f seated=%d (%.2f s) (%d triangle copies): host BVH build %.1f ms (all meshes, worker threads), device build %.3f ms, %d nodes, integrator set-up %.2f s | %.3f ms per step, %.2f nodes and %.2f triangles per closest-hit ray"
              % (device_blas, presplit, seating, pt.last_reseat_seconds, int((pt.array("alias_mesh_ids") >= 0).sum()), scene.bvh_build_ms, pt.device_blas_build_ms, nodes, setup_s, ms, stats["closest"]["nodes"] / max(rays, 1), stats["closest"]["triangles"] / max(rays, 1)), flush=True)
        pt.close(); scene.close()


if __name__ == "__main__":
    main()
