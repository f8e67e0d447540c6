"""Can small persistent trace launches from different streams overlap on this GPU? N contexts (N streams),
each traces the same 20 000 incoherent rays `repeat` times from its own host thread (ctypes releases the
GIL); wall time for N in parallel vs one alone."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402


def main():
    n_rays, repeat = int(os.environ.get("RAYS", "20000")), 40
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("sponza"))
    pts = [grt.Pathtracer(scene, 640, 360, device=0) for _ in range(8)]
    for pt in pts:
        pt.update()
    o, d, _ = grt.generate_rays(pts[0].ctx, 0, 0, 640 * 360)
    hits, _ = grt.trace_rays(pts[0].ctx, o, d)
    t = hits[:, 2].view(np.float32); ok = hits[:, 1] != 0xffffffff
    rng = np.random.default_rng(1)
    so = (o + d * np.where(ok, t, 1).astype(np.float32) * np.float32(0.999))[:, ok][:, :n_rays]
    sd = rng.normal(size=so.shape).astype(np.float32); sd /= np.linalg.norm(sd, axis=0)
    for pt in pts:
        grt.trace_rays(pt.ctx, so, sd, repeat=3)
    for n in (1, 2, 4, 8):
        kernel_ms = [0.0] * n
        def work(i):
            _, ms = grt.trace_rays(pts[i].ctx, so, sd, repeat=repeat)
            kernel_ms[i] = ms
        threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        t0 = time.perf_counter()
        for th in threads: th.start()
        for th in threads: th.join()
        wall = (time.perf_counter() - t0) * 1e3
        print("%d streams x %d launches of %d rays: wall %.2f ms, %.3f ms per launch per stream (HIP events: %.3f ms); ideal if fully concurrent = the 1-stream figure" % (n, repeat, so.shape[1], wall, wall / repeat, float(np.mean(kernel_ms))), flush=True)
    for pt in pts:
        pt.close()
    scene.close()


if __name__ == "__main__":
    main()
