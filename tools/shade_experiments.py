#!/usr/bin/env python3
"""Where does the shade stage's time go? The benchmark's frame loop (Sponza 1080p, 4-sample submissions in the merged
wavefront) with one feature of the shade kernels switched off at a time, stage times from rt_set_profiling(ctx, 3)
(HIP events around every launch). usage (GPU box): python tools/shade_experiments.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402


def run(label, steps, **config):
    scene = bench.build_scene(grt)
    grt.config_set(**config)
    pt = grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0); pt.update()
    lib, ctx = grt.device_lib(), pt.ctx
    def frames(n):
        for k in range(n):
            assert lib.rt_render_samples(ctx, 0, bench.SPP) == 0
        lib.rt_synchronize(ctx)
    import ctypes
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    frames(2)
    grt.set_profiling(ctx, 3)
    frames(steps // bench.SPP)
    out = []
    for kind in ("trace", "sort", "material_diffuse", "material_plastic", "generate", "accumulate"):
        t = grt.launch_timings(ctx, kind)
        out.append("%s %.3f" % (kind.replace("material_", ""), float(t.sum()) / steps))
    grt.set_profiling(ctx, False)
    print("%-34s ms per step: %s" % (label, "  ".join(out)), flush=True)
    pt.close(); scene.close()


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    run("default", steps)
    run("mipmapping off", steps, enable_mipmapping=0)
    run("block compression off (RGBA8)", steps, enable_block_compression=0)
    run("next event estimation off", steps, enable_next_event_estimation=0)
    run("MIS off", steps, enable_multiple_importance_sampling=0)
    run("2 bounces", steps, num_bounces=2)


if __name__ == "__main__":
    main()
