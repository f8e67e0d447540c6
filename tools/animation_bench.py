"""Animated scene: 441 instances, every frame 40 of them move, the host rebuilds the TLAS
(Integrator::build_tlas) and uploads it with the instance tables; one sample per frame at 1920x1080.
Frames in flight 1 vs 3, and the old behaviour (drain the GPU around every upload) for comparison.
usage (GPU box): python tools/animation_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402
import config_suite  # noqa: E402


def main():
    grt.config_reset(); grt.config_set(num_bounces=10)
    scene = grt.Scene(config_suite.instancing_scene(os.path.join(ROOT, "assets", "_cache", "configs", "instancing")))
    pt = grt.Pathtracer(scene, 1920, 1080, device=0); pt.update()
    lib, ctx = grt.device_lib(), pt.ctx
    base = [scene.mesh_transform(m) for m in range(scene.mesh_count)]

    def animate(frame):
        for m in range(2 + frame % 11, scene.mesh_count, 11):
            pos, _, scale = base[m]
            a = 0.1 * frame + m
            scene.set_mesh_transform(m, [pos[0], pos[1] + 0.3 * np.sin(a), pos[2]], [0.0, float(np.sin(a / 2)), 0.0, float(np.cos(a / 2))], scale)

    for label, in_flight, drain in (("1 frame in flight, drained around uploads (old behaviour)", 1, True), ("1 frame in flight", 1, False), ("3 frames in flight", 3, False)):
        grt.set_samples_in_flight(ctx, in_flight)
        for f in range(6):
            animate(f); pt.invalidate("scene"); pt.update(); lib.rt_render_sample(ctx, 1)
        lib.rt_synchronize(ctx)
        frames, t_host = 48, 0.0
        t0 = time.perf_counter()
        for f in range(frames):
            h0 = time.perf_counter()
            animate(f)
            pt.invalidate("scene")
            if drain:
                lib.rt_synchronize(ctx)
            pt.update()            # Mesh::update, SAH TLAS build, CWBVH conversion, versioned uploads
            t_host += time.perf_counter() - h0
            lib.rt_render_sample(ctx, 1)
        lib.rt_synchronize(ctx)
        ms = (time.perf_counter() - t0) / frames * 1e3
        print("%-62s %.3f ms per frame (host animate + TLAS rebuild + upload %.3f ms of it, not overlapped when drained)" % (label, ms, t_host / frames * 1e3), flush=True)
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
