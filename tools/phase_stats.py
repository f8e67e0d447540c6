"""Debug: SIMD lane occupancy of the node / triangle phases of kernel_trace_bvh8 per bounce.
Needs a library built with -DRT_PHASE_STATS (GRT_DEVICE_LIB=...)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402


def main():
    scene = bench.build_scene(grt)
    for bounces in (1, 2, 3, 10):
        grt.config_set(num_bounces=bounces)
        pt = grt.Pathtracer(scene, bench.WIDTH, bench.HEIGHT, device=0)
        pt.update()
        grt.set_trace_statistics(pt.ctx, True)
        grt.device_lib().rt_render_sample(pt.ctx, 1)
        raw = np.zeros(10, np.uint64)
        grt.device_lib().rt_get_trace_statistics.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        grt.device_lib().rt_get_trace_statistics(pt.ctx, raw.ctypes.data)
        nodes, tris, _, _, rays, iters, node_execs, node_lanes, tri_rounds, tri_lanes = [int(v) for v in raw]
        print("bounces<=%2d rays %9d nodes/ray %.2f tris/ray %.2f | wave iterations %d, node executions %d (%.1f lanes = %.0f%%), triangle rounds %d (%.1f lanes = %.0f%%), tri rounds per iteration %.2f" % (
            bounces, rays, nodes / rays, tris / rays, iters, node_execs, node_lanes / max(node_execs, 1), 100 * node_lanes / max(node_execs, 1) / 64,
            tri_rounds, tri_lanes / max(tri_rounds, 1), 100 * tri_lanes / max(tri_rounds, 1) / 64, tri_rounds / max(iters, 1)))
        pt.close()
    scene.close()


if __name__ == "__main__":
    main()
