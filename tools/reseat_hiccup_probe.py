#!/usr/bin/env python3
"""Frame times around a seating beside the frame loop (Integrator::start_reseat_worker): the camera jumps to the reference's ninth Sponza point of view, then
4-sample frames are rendered one by one (update + render + synchronize) and timed; prints the frames above 1.5 x the median with the worker's state.
usage (GPU box): python tools/reseat_hiccup_probe.py [frames = 300]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    import bench
    import gpu_raytracer_amd as grt
    scene = bench.build_scene(grt)
    pt = grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0); pt.update()
    lib = grt.device_lib(); ctx = pt.ctx
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.rt_synchronize.argtypes = [ctypes.c_void_p]
    def frame():
        assert lib.rt_render_samples(ctx, 0, bench.SPP) == 0
        assert lib.rt_synchronize(ctx) == 0
    for _ in range(5):
        frame()
    position, rotation = bench.SPONZA_POVS[8]
    scene.set_camera(position, rotation)
    rows = []
    for f in range(frames):
        t0 = time.perf_counter(); pt.update(); t1 = time.perf_counter(); frame(); t2 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, pt.reseat_pending, pt.reseats_completed))
    upd = np.array([r[0] for r in rows]) * 1e3; ren = np.array([r[1] for r in rows]) * 1e3
    print("frames %d: render median %.3f ms, max %.3f; update median %.3f ms, max %.3f; seatings completed %d (%.2f s on the worker)" % (frames, np.median(ren), ren.max(), np.median(upd), upd.max(), rows[-1][3], pt.last_reseat_seconds))
    last_state = None
    for f, (u, r, pending, done) in enumerate(rows):
        state = (pending, done)
        if r * 1e3 > 1.5 * np.median(ren) or u * 1e3 > 1.0 or state != last_state:
            print("  frame %3d: update %8.3f ms  render %8.3f ms  worker pending %d  completed %d" % (f, u * 1e3, r * 1e3, pending, done))
        last_state = state
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
