"""Render a bundled / generated scene on the GPU and write a tone-mapped PNG (no dependencies).
usage: python tools/render_png.py <cornellbox|sponza|instancing|glass|sponza_ao> <out.png> [width height spp]"""
import os
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def main():
    name, out = sys.argv[1], sys.argv[2]
    w, h, spp = (int(v) for v in (sys.argv[3:6] + ["480", "270", "64"][len(sys.argv) - 3:]))
    cache = os.path.join(ROOT, "assets", "_cache", "configs")
    grt.config_reset()
    ao = False
    if name == "sponza" or name == "sponza_ao":
        scene = bench.build_scene(grt); ao = name.endswith("_ao")
    elif name == "instancing":
        import config_suite
        scene = grt.Scene(config_suite.instancing_scene(os.path.join(cache, "instancing"))); grt.config_set(num_bounces=10)
    elif name == "glass":
        import config_suite
        scene = grt.Scene(config_suite.glass_scene(os.path.join(cache, "glass")))
    else:
        scene = grt.Scene(grt.scene_path(name)); grt.config_set(num_bounces=8)
    if ao:
        pt = grt.AO(scene, w, h, device=0, radius=2.0)
        for s in range(spp):
            pt.update(); pt.render()
    else:
        pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
        pt.render_samples(1)          # sample 0 (the reference's accumulate overwrites it with sample 1)
        left = spp
        while left > 0:
            pt.update(); n = min(left, 16); pt.render_samples(n); left -= n
    img = pt.read_framebuffer()[:, :w, :3]
    print("%s %dx%d %d spp: mean %.4f max %.3f finite %s" % (name, w, h, spp, img.mean(), img.max(), np.isfinite(img).all()))
    ldr = np.clip(img / (1.0 + img) if not ao else img, 0.0, 1.0) ** (1.0 / 2.2)   # Reinhard + gamma
    write_png(out, np.ascontiguousarray((ldr * 255.0 + 0.5).astype(np.uint8)[::-1]))   # row 0 is the bottom row (GL convention of the reference)
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
