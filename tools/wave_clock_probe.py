#!/usr/bin/env python3
"""The drain of the persistent traversal launch, wave by wave. Needs the probe build of the device library
(tools/build_trace_variant.sh waveclock "-DRT_WAVE_CLOCK=1"; GRT_DEVICE_LIB=gpu-raytracer_amd/csrc/_variants/waveclock/libgrt_device.so):
every wave leaves the 100 MHz clock at which it started, left the closest-hit engine and left the launch. Submits the driver's plan (20 steps = five
4-sample frames as one burst), on the whole frame or on rank 0's tiles of an N-way split (--world N), and prints per traversal launch of the LAST burst:
rays, duration (first start to last end), when 50 / 90 / 99 % of the waves had left, and the wave-time lost between a wave's exit and the launch's end.
usage (GPU box): GRT_DEVICE_LIB=... python tools/wave_clock_probe.py [--world 8] [--steps 20]"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--world", type=int, default=1); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    import bench
    import gpu_raytracer_amd as grt
    import importlib
    parallel = importlib.import_module("gpu_raytracer_amd.parallel")
    scene = bench.build_scene(grt)
    pt = grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0); pt.update()
    lib = grt.device_lib(); ctx = pt.ctx
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.rt_synchronize.argtypes = [ctypes.c_void_p]
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_debug_read_wave_clock.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    pixels = bench.WIDTH * bench.HEIGHT
    if a.world > 1:
        split = parallel.TileSplit(0, a.world, bench.WIDTH, bench.HEIGHT)
        assert lib.rt_set_pixel_tiles(ctx, split.tile_pixels, 0, a.world) == 0
        pixels = split.local_pixels
    plan, k = [], 0
    while k < a.steps:
        first = k % bench.SPP; count = min(bench.SPP - first, a.steps - k); k += count; plan.append((first, count))
    grt.set_frame_pipelining(ctx, True)
    grt.set_stream_batch(ctx, sum(c for _, c in plan[:8]) * pixels)
    for _ in range(a.repeat):
        base = grt.submissions_completed(ctx)
        for first, count in plan:
            assert lib.rt_render_samples(ctx, first, count) == 0, lib.rt_last_error(ctx)
        while grt.submissions_completed(ctx) - base < len(plan):
            grt.advance(ctx)
        assert lib.rt_synchronize(ctx) == 0
    SLOTS, WAVES = 32, 16384
    clocks = np.zeros((SLOTS, WAVES, 3), np.uint64); meta = np.zeros((SLOTS, 4), np.int32)
    assert lib.rt_debug_read_wave_clock(clocks.ctypes.data, meta.ctypes.data) == 0
    order = np.argsort(meta[:, 0])
    rows = [s for s in order if meta[s, 3] > 0]
    print("world %d, %d steps; per traversal launch (iteration order, the last %d launches): times in us" % (a.world, a.steps, len(rows)))
    print("%5s %10s %10s %6s | %8s | %7s %7s %7s %7s | %9s %9s | %s" % ("iter", "closest", "shadow", "waves", "launch", "50%", "90%", "99%", "first", "lost", "lost/all", "closest phase: last wave out, 50 % out"))
    for s in rows:
        n = int(meta[s, 3]); c = clocks[s, :n].astype(np.float64) / 100.0   # us
        took_part = c[:, 2] > 0
        ran = took_part & ((c[:, 2] - c[:, 0]) > 2.0)    # waves that claimed rays (the rest of the machine-sized grid leaves at once)
        if not ran.any():
            continue
        t0 = c[took_part, 0].min(); end = c[ran, 2].max(); ends = np.sort(c[ran, 2] - t0); mids = np.sort(c[ran, 1] - t0)
        lost = (end - t0 - ends).sum()
        q = lambda arr, f: arr[min(len(arr) - 1, int(f * len(arr)))]
        print("%5d %10d %10d %6d | %8.1f | %7.1f %7.1f %7.1f %7.1f | %9.0f %9.3f | %8.1f %8.1f" % (meta[s, 0], meta[s, 1], meta[s, 2], int(ran.sum()), end - t0, q(ends, 0.5), q(ends, 0.9), q(ends, 0.99), ends[0],
                                                                               lost, lost / (len(ends) * (end - t0)), mids[-1], q(mids, 0.5)))
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
