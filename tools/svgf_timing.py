"""Stage times of an SVGF + TAA frame at the BASELINE size (config #3): Sponza 1920x1080, 6 a-trous
iterations, TAA on. Profiling mode 1 (serialised), median of the frames after warm-up."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402


def main():
    scene = bench.build_scene(grt)
    grt.config_set(enable_svgf=1, enable_taa=1, num_atrous_iterations=6)
    pt = grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0)
    pt.update()
    for _ in range(4):
        pt.render(); pt.update()
    grt.set_profiling(pt.ctx, True)
    rows = []
    for _ in range(12):
        pt.render(); c = pt.counters(); pt.update()
        rows.append([c.ms_generate, c.ms_trace, c.ms_sort, c.ms_shade, c.ms_shadow, c.ms_post, c.ms_total])
    med = np.median(np.array(rows), axis=0)
    px = bench.WIDTH * bench.HEIGHT
    print("SVGF+TAA frame, Sponza 1920x1080, 1 spp: generate %.3f trace %.3f sort %.3f shade %.3f shadow %.3f | svgf+taa %.3f ms | total %.3f ms" % tuple(med))
    print("svgf+taa: %.1f ns per pixel; 6 a-trous iterations" % (med[5] * 1e6 / px))
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
