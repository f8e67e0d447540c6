#!/usr/bin/env python3
"""Does a SECOND wavefront on the same GPU hide the tails of the first one's launches?

A rank of an 8-way tile split runs launches an eighth the size of the whole frame's: every traversal launch ends in a
tail (a persistent launch cannot end before its longest ray) and the deep bounces are a chain of small launches
(profiles/r04_tile_split_rank_timeline.txt). This probe renders the same share of the frame with 1, 2, 3 ... contexts
that sit on ONE GPU, each with its own stream and its own merged wavefront over a subset of the rank's tiles (context c
of k: tiles first + world * c, stride world * k), submitted from one host thread: while one context's traversal launch
drains, the other's kernels take the wave slots it frees.

Usage: tools/dual_wavefront_probe.py [--steps 20] [--configs N1x1,N1x2,N8x1,N8x2,N8x3,N8x4] [--repeat 5]
  NWxK: rank 0's share of a W-way split rendered by K contexts. Prints ms per step (min / median over the repeats).
No exchange in any row (bench.py --emulate-world adds a pack + unpack per frame)."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the benchmark's scene and constants)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--configs", default="N1x1,N1x2,N8x1,N8x2,N8x3,N8x4")
    ap.add_argument("--repeat", type=int, default=5)
    args = ap.parse_args()
    import gpu_raytracer_amd as grt
    lib = grt.device_lib()
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.rt_synchronize.argtypes = [ctypes.c_void_p]
    W, H, SPP = bench.WIDTH, bench.HEIGHT, bench.SPP
    tile_pixels = W * 8
    tiles_total = (H + 7) // 8
    scene = bench.build_scene(grt)

    def check(ctx, status):
        if status != 0:
            raise RuntimeError(lib.rt_last_error(ctx).decode())

    plan = []
    k = 0
    while k < args.steps:
        count = min(SPP, args.steps - k); plan.append((k % SPP, count)); k += count

    for config in args.configs.split(","):
        world, contexts = (int(v) for v in config[1:].split("x"))
        stride = world * contexts
        pts = [grt.Pathtracer(scene, W, H, device=0) for _ in range(contexts)]
        ctxs = []
        for c, pt in enumerate(pts):
            pt.update()
            ctx = pt.ctx
            grt.set_scheduler(ctx, "merged")
            first = world * c
            if stride > 1:
                check(ctx, lib.rt_set_pixel_tiles(ctx, tile_pixels, first, stride))
            pixels = W * H if stride == 1 else len(range(first, tiles_total, stride)) * tile_pixels
            grt.set_frame_pipelining(ctx, True)
            grt.set_stream_batch(ctx, sum(count for _, count in plan[:8]) * pixels)
            ctxs.append(ctx)

        def run():
            base = [grt.submissions_completed(ctx) for ctx in ctxs]
            for first, count in plan:
                for ctx in ctxs:
                    check(ctx, lib.rt_render_samples(ctx, first, count))
            pending = list(range(len(ctxs)))
            while pending:
                for i in list(pending):
                    if grt.submissions_completed(ctxs[i]) - base[i] >= len(plan):
                        pending.remove(i)
                    else:
                        grt.advance(ctxs[i])
            for ctx in ctxs:
                check(ctx, lib.rt_synchronize(ctx))

        run(); run()
        times = []
        for _ in range(args.repeat):
            t0 = time.perf_counter(); run(); times.append((time.perf_counter() - t0) / args.steps * 1e3)
        times = np.sort(times)
        print("%-6s rank 0 of %d by %d context(s): %.4f ms per step (median %.4f, max %.4f), %d steps" % (config, world, contexts, times[0], times[len(times) // 2], times[-1], args.steps), flush=True)
        for pt in pts:
            pt.close()
    scene.close()


if __name__ == "__main__":
    main()
