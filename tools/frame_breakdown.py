"""Per-launch breakdown of one bench sample: GRT_STAGE_TRACE prints every stage interval (HIP
events, serialised profiling pass), this script adds the queue sizes per bounce.
usage (GPU box): GRT_STAGE_TRACE=1 python tools/frame_breakdown.py [sample_index]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GRT_STAGE_TRACE", "1")
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402


def main():
    sample = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    scene = bench.build_scene(grt)
    pt = grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0)
    pt.update()
    lib = grt.device_lib()
    for _ in range(3):
        lib.rt_render_sample(pt.ctx, sample)
    lib.rt_synchronize(pt.ctx)
    grt.set_profiling(pt.ctx, True)
    lib.rt_render_sample(pt.ctx, sample)
    lib.rt_render_sample(pt.ctx, sample)
    sys.stderr.flush()
    print("---- profiled sample (stage lines above are printed by the library for this call)", flush=True)
    c = pt.counters()
    nb = bench.NUM_BOUNCES
    for b in range(nb):
        print("bounce %2d: trace %8d shadow %8d diffuse %8d plastic %8d" % (b, c.trace[b], c.shadow[b], c.diffuse[b], c.plastic[b]))
    print("ms: generate %.3f trace %.3f sort %.3f shade %.3f shadow %.3f post %.3f total %.3f" % (c.ms_generate, c.ms_trace, c.ms_sort, c.ms_shade, c.ms_shadow, c.ms_post, c.ms_total))
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
