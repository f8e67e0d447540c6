"""Animated scene: 441 instances, every frame 40 of them move and the TLAS is rebuilt (Integrator::build_tlas) -- on the
host (SAH build + CWBVH conversion + uploads of the TLAS and the instance tables) or on the device (rt_build_tlas: one
launch); one sample per frame at 1920x1080. Frames in flight 1 vs 3, and the reference's protocol (drain the GPU around
every upload) for comparison.
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


def big_scene(directory, count):
    """`count` instances of a small mesh (the TLAS build is what is being timed, not the BLAS)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_tlas import instanced_scene_file
    return instanced_scene_file(directory, count=count, seed=11)


def main():
    count = int(os.environ.get("ANIM_INSTANCES", "0"))
    path = big_scene(os.path.join(ROOT, "assets", "_cache", "configs", "instances_%d" % count), count) if count else config_suite.instancing_scene(os.path.join(ROOT, "assets", "_cache", "configs", "instancing"))
    print("scene: %s" % ("%d instances of a 256-triangle mesh" % count if count else "441 instances x 102 400 triangles"), flush=True)
    for label, device_tlas, in_flight, drain in (("host TLAS, 1 frame in flight, drained around uploads (reference protocol)", 0, 1, True), ("host TLAS, 1 frame in flight", 0, 1, False),
                                                 ("host TLAS, 3 frames in flight", 0, 3, False), ("DEVICE TLAS (rt_build_tlas), 1 frame in flight", 1, 1, False), ("DEVICE TLAS (rt_build_tlas), 3 frames in flight", 1, 3, False)):
        grt.config_reset(); grt.config_set(num_bounces=10, device_tlas=device_tlas)
        scene = grt.Scene(path)
        pt = grt.Pathtracer(scene, 1920, 1080, device=0); pt.update()
        lib, ctx = grt.device_lib(), pt.ctx
        grt.set_scheduler(ctx, "slots")       # frames in flight across scene versions: the slot scheduler (what enable_scene_update selects)
        base = [scene.mesh_transform(m) for m in range(scene.mesh_count)]

        def animate(frame):
            for m in range(3 + frame % 11, scene.mesh_count, 11):
                pos, _, scale = base[m]
                a = 0.1 * frame + m
                scene.set_mesh_transform(m, [pos[0], pos[1] + 0.3 * np.sin(a), pos[2]], [0.0, float(np.sin(a / 2)), 0.0, float(np.cos(a / 2))], scale)

        grt.set_samples_in_flight(ctx, in_flight)
        for f in range(6):
            animate(f); pt.invalidate("scene"); pt.update(); lib.rt_render_sample(ctx, 1)
        lib.rt_synchronize(ctx)
        # host work per frame: over the first 8 frames, while the host still runs ahead of the device (12 scene versions: from
        # then on Integrator::update waits for the frame that last read the version it is about to overwrite -- backpressure)
        frames, t_update, head = 48, 0.0, 8
        t0 = time.perf_counter()
        for f in range(frames):
            animate(f)
            pt.invalidate("scene")
            if drain:
                lib.rt_synchronize(ctx)
            h1 = time.perf_counter()
            pt.update()            # Mesh::update + (host: SAH TLAS build, CWBVH conversion, versioned uploads | device: scene-order tables, one launch)
            if f < head:
                t_update += time.perf_counter() - h1
            lib.rt_render_sample(ctx, 1)
        lib.rt_synchronize(ctx)
        ms = (time.perf_counter() - t0) / frames * 1e3
        print("%-78s %.3f ms per frame; Integrator::update on the host %.3f ms per frame" % (label, ms, t_update / head * 1e3), flush=True)
        pt.close(); scene.close()


if __name__ == "__main__":
    main()
