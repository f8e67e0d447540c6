#!/usr/bin/env python3
"""Experiment: is there anything to gain from running one wavefront's sort / shade launches BESIDE another wavefront's traversal launch?
Two contexts on ONE GPU render the benchmark's frames (4-spp submissions, merged wavefront each) from two host threads; the aggregate
rate is compared with one context rendering all of them. GRT_TRACE_BLOCKS_PER_CU caps the persistent traversal grid so that the other
context's kernels find free wave slots.   python tools/two_context_overlap.py [frames]"""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import gpu_raytracer_amd as grt

def run(contexts, frames_each):
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; lib.rt_synchronize.argtypes = [ctypes.c_void_p]
    scene = bench.build_scene(grt)
    pts = [grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0) for _ in range(contexts)]
    for pt in pts:
        pt.update()
        for _ in range(3): lib.rt_render_samples(pt.ctx, 0, bench.SPP)
        lib.rt_synchronize(pt.ctx)
    lib.rt_render_samples(pts[0].ctx, 0, 1); rays = sum(pts[0].counters().trace[:bench.NUM_BOUNCES]); lib.rt_synchronize(pts[0].ctx)
    def loop(pt):
        for _ in range(frames_each): lib.rt_render_samples(pt.ctx, 0, bench.SPP)
        lib.rt_synchronize(pt.ctx)
    threads = [threading.Thread(target=loop, args=(pt,)) for pt in pts]
    t0 = time.perf_counter()
    for t in threads: t.start()
    for t in threads: t.join()
    elapsed = time.perf_counter() - t0
    steps = contexts * frames_each * bench.SPP
    for pt in pts: pt.close()
    scene.close()
    return elapsed / steps * 1e3, rays * steps / elapsed / 1e6

if __name__ == "__main__":
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    contexts = int(os.environ.get("CONTEXTS", "1"))
    ms, mrays = run(contexts, frames // contexts)
    print("contexts %d  GRT_TRACE_BLOCKS_PER_CU %s: %.4f ms per step, %.1f Mrays/s aggregate" % (contexts, os.environ.get("GRT_TRACE_BLOCKS_PER_CU", "-"), ms, mrays))
