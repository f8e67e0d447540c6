"""What one rank of an N-GPU job does, on one GPU: render rank 0's tiles of world W (no collective)
with different numbers of samples in flight; prints ms per sample and the implied speed-up bound.
usage (GPU box): python tools/rank_emulation.py [worlds ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402

parallel = __import__("importlib").import_module("gpu_raytracer_amd.parallel")


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    scene = bench.build_scene(grt)
    pt = grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0)
    pt.update()
    lib, ctx = grt.device_lib(), pt.ctx
    import ctypes
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    base = None
    for world in worlds:
        split = parallel.TileSplit(0, world, bench.WIDTH, bench.HEIGHT)
        for in_flight in [int(v) for v in os.environ.get("IN_FLIGHT", "1,2,3,4,6,8").split(",")]:
            grt.set_samples_in_flight(ctx, in_flight)
            if world == 1:
                lib.rt_set_pixel_range(ctx, 0, bench.WIDTH * bench.HEIGHT)
            else:
                lib.rt_set_pixel_tiles(ctx, split.tile_pixels, 0, world)
            batch = int(os.environ.get("BATCH", "1"))
            for k in range(0, 8, batch):
                lib.rt_render_samples(ctx, k % 4, batch)
            lib.rt_synchronize(ctx)
            t0 = time.perf_counter()
            steps = 48
            for k in range(0, steps, batch):
                lib.rt_render_samples(ctx, k % 4, batch)
            submit_ms = (time.perf_counter() - t0) / steps * 1e3
            lib.rt_synchronize(ctx)
            ms = (time.perf_counter() - t0) / steps * 1e3
            if base is None or (world == 1 and ms < base):
                base = ms if world == 1 else base
            print("world %d  in flight %d  %.3f ms/sample, host submission %.3f ms/sample  (speed-up over best 1-GPU %.2fx)" % (world, in_flight, ms, submit_ms, (base or ms) / ms), flush=True)
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
