"""BASELINE.json configs 3, 4 and 5 at their full sizes on one MI355X (config 2 is bench.py).

  3  Sponza 1920x1080, SVGF (6 a-trous iterations) + TAA, one sample per frame
  4  instancing stand-in (the LEGO scene is not in the mount): 21 x 21 = 441 rotated /
     scaled instances of one seeded ~100 k triangle mesh over a floor, area light, diffuse + plastic
  5  glass-of-water stand-in: box with a rough-dielectric sphere enclosing a homogeneous medium,
     a smooth dielectric, a rough conductor, area light; 16 spp

Prints per config: rays and ms per sample per pixel, Mrays/s (closest-hit rays), and for 4 / 5 how
the frame was submitted. usage (GPU box): python tools/config_suite.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np  # noqa: E402
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402

W, H = 1920, 1080


def blob_obj(n, seed):
    """Closed lumpy sphere, n x 2n quads (2 triangles each), value-noise-like displacement."""
    rng = np.random.default_rng(seed)
    k = rng.uniform(1.0, 6.0, (6, 2)); a = rng.uniform(0.02, 0.08, 6); ph0 = rng.uniform(0, 6.28, 6)
    th = np.pi * np.arange(n + 1)[:, None] / n
    ph = np.pi * np.arange(2 * n)[None, :] / n
    r = 1.0 + sum(a[i] * np.sin(k[i, 0] * th + ph0[i]) * np.cos(k[i, 1] * ph) for i in range(6))
    x, y, z = r * np.sin(th) * np.cos(ph), r * np.cos(th) * np.ones_like(ph), r * np.sin(th) * np.sin(ph)
    lines = ["v %.6f %.6f %.6f" % t for t in zip(x.ravel(), y.ravel(), z.ravel())]
    for i in range(n):
        for j in range(2 * n):
            p, q = i * 2 * n + j + 1, i * 2 * n + (j + 1) % (2 * n) + 1
            s, t = (i + 1) * 2 * n + (j + 1) % (2 * n) + 1, (i + 1) * 2 * n + j + 1
            lines.append("f %d %d %d %d" % (p, q, s, t))
    return "\n".join(lines) + "\n"


def instancing_scene(directory):
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "blob.obj"), "w") as f:
        f.write(blob_obj(160, 1234))
    rng = np.random.default_rng(7)
    shapes = []
    for i in range(21):
        for j in range(21):
            bsdf = '<bsdf type="roughplastic"><rgb name="diffuseReflectance" value="%.2f, %.2f, %.2f"/><float name="alpha" value="0.3"/></bsdf>' % tuple(rng.uniform(0.2, 0.8, 3)) \
                if (i + j) % 2 else '<bsdf type="diffuse"><rgb name="reflectance" value="%.2f, %.2f, %.2f"/></bsdf>' % tuple(rng.uniform(0.2, 0.8, 3))
            shapes.append('<shape type="obj"><string name="filename" value="blob.obj"/><transform name="toWorld"><scale value="%.3f"/>'
                          '<rotate y="1" angle="%.1f"/><rotate x="1" angle="%.1f"/><translate x="%.2f" y="%.2f" z="%.2f"/></transform>%s</shape>'
                          % (rng.uniform(0.7, 1.2), rng.uniform(0, 360), rng.uniform(-30, 30), 3.0 * (i - 10), 1.3, 3.0 * (j - 10), bsdf))
    xml = ('<scene version="0.5.0"><integrator type="path"><integer name="maxDepth" value="10"/></integrator>'
           '<sensor type="perspective"><float name="fov" value="55"/><transform name="toWorld"><lookat origin="0, 14, 42" target="0, 0, 0" up="0, 1, 0"/></transform></sensor>'
           '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="-90"/><scale value="60"/></transform><bsdf type="diffuse"><rgb name="reflectance" value="0.5, 0.5, 0.5"/></bsdf></shape>'
           '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><scale value="12"/><translate y="30"/></transform><emitter type="area"><rgb name="radiance" value="30, 28, 25"/></emitter></shape>'
           '%s</scene>' % "".join(shapes))
    path = os.path.join(directory, "scene.xml")
    with open(path, "w") as f:
        f.write(xml)
    return path


def glass_scene(directory):
    os.makedirs(directory, exist_ok=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_materials_svgf import GLASS_SCENE  # the parity-tested scene, here at 1080p / 16 spp
    path = os.path.join(directory, "scene.xml")
    with open(path, "w") as f:
        f.write(GLASS_SCENE)
    return path


def measure(pt, samples, batch, label, svgf=False):
    lib, ctx = grt.device_lib(), pt.ctx
    def frame():
        if svgf:
            for s in range(samples):
                lib.rt_render_sample(ctx, s)
        else:
            for first in range(0, samples, batch):
                lib.rt_render_samples(ctx, first, min(batch, samples - first))
    for _ in range(3):
        frame()
    lib.rt_synchronize(ctx)
    grt.set_trace_statistics(ctx, True)
    rays = shadow = 0
    for s in range(min(samples, 4)):
        lib.rt_render_sample(ctx, s); c = pt.counters()
        rays += sum(c.trace); shadow += sum(c.shadow)
    rays /= min(samples, 4); shadow /= min(samples, 4)
    grt.set_trace_statistics(ctx, False)
    frame(); lib.rt_synchronize(ctx)
    reps = 16 if svgf else 6   # SVGF: 64 filtered frames, so that the 9 iterations that drain the wavefront do not dominate
    t0 = time.perf_counter()
    for _ in range(reps):
        frame()
    lib.rt_synchronize(ctx)
    ms = (time.perf_counter() - t0) / (reps * samples) * 1e3
    img = pt.read_framebuffer()[:, :W, :3]
    print("%s: %.2f M rays + %.2f M shadow rays per sample, %.3f ms per sample (%.2f ms per %d-spp frame), %.0f Mrays/s; image mean %.4f finite %s"
          % (label, rays / 1e6, shadow / 1e6, ms, ms * samples, samples, rays / ms / 1e3, float(img.mean()), bool(np.isfinite(img).all())), flush=True)


def main():
    cache = os.path.join(ROOT, "assets", "_cache", "configs")
    only = os.environ.get("CONFIG_ONLY", "")
    # config 3 (under both schedulers: the merged wavefront is the default, the per-frame launch chains are round 1's)
    for scheduler in ("merged", "slots"):
        scene = bench.build_scene(grt)
        grt.config_set(enable_svgf=1, enable_taa=1, num_atrous_iterations=6)
        pt = grt.Pathtracer(scene, W, H, device=0); pt.update()
        grt.set_scheduler(pt.ctx, scheduler)
        measure(pt, 4, 1, "config 3  Sponza + SVGF/TAA (1 sample per filtered frame), %s scheduler" % scheduler, svgf=True)
        pt.close(); scene.close()
    if only == "3":
        return
    # config 4
    grt.config_reset()
    t0 = time.perf_counter()
    scene = grt.Scene(instancing_scene(os.path.join(cache, "instancing")))
    grt.config_set(num_bounces=10)
    pt = grt.Pathtracer(scene, W, H, device=0); pt.update()
    print("config 4  scene load + BLAS/TLAS build %.2f s" % (time.perf_counter() - t0))
    measure(pt, 4, 4, "config 4  441 instances x 102 400 triangles, diffuse + plastic, 4 spp (one submission)")
    pt.close(); scene.close()
    # config 5
    grt.config_reset()
    scene = grt.Scene(glass_scene(os.path.join(cache, "glass")))
    pt = grt.Pathtracer(scene, W, H, device=0); pt.update()
    measure(pt, 16, 8, "config 5  rough dielectric + medium + conductor, 16 spp (two submissions of 8)")
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
