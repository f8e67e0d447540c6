#!/usr/bin/env python3
"""Fuzzing of what sits between the loaders and the device for the flattened static geometry: the parallel spatial-split builder, the 8-wide collapse and the
seating learner (host/StaticBVHBuilder.cpp, BVH.cpp, SlotOrder.cpp), on triangle soups with degenerate, extreme and non-finite vertices and on whole scenes
(OBJ meshes through the loaders into a host-only integrator). Meant for the sanitizer build of the host library, like tools/fuzz_loaders.py:

    make -C gpu-raytracer_amd host/libgrt_host_asan.so
    cp gpu-raytracer_amd/host/libgrt_host.so /tmp/keep.so && cp gpu-raytracer_amd/host/libgrt_host_asan.so gpu-raytracer_amd/host/libgrt_host.so
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" \
        python tools/fuzz_flatten.py <seed> <iterations> [soups|scenes]
    cp /tmp/keep.so gpu-raytracer_amd/host/libgrt_host.so

Every input must either build or be rejected with an error; a sanitizer report is a bug. Round 5 ran 300 + 300 iterations clean AFTER the one it found:
BVH8Converter::gather_children walked past a node's eight slots on cost tables of NaN / +inf (tests/test_static_geometry.py pins it)."""
import ctypes
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpu_raytracer_amd as grt  # noqa: E402


def soups(rng, iterations):
    lib = grt.host_lib()
    for it in range(iterations):
        n = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 33, 200, 3000, 12000]))
        tri = rng.uniform(-1, 1, (n, 1, 3)) * rng.choice([1.0, 1e-6, 1e6]) + rng.normal(size=(n, 3, 3)) * rng.choice([0.0, 1e-3, 0.05, 1.0])
        for _ in range(int(rng.integers(0, 4))):
            tri[rng.integers(n), rng.integers(3), rng.integers(3)] = rng.choice([np.inf, -np.inf, np.nan, 3e38, -3e38, 0.0, 1e-40])
        if rng.random() < 0.2:
            tri[:] = tri[0]
        t24 = np.zeros((n, 24), np.float32); t24[:, :9] = tri.astype(np.float32).reshape(-1, 9)
        handle = lib.grt_build_static_bvh(t24.ctypes.data, n, int(rng.integers(0, 5)))
        if not handle:
            print(it, n, "rejected:", lib.grt_last_error()[:80], flush=True); continue
        rc = lib.grt_built_learn_slot_order(handle, int(rng.choice([0, 100, 5000, 40000])), int(rng.integers(0, 9)))
        size = ctypes.c_size_t(0); lib.grt_built_array(handle, b"bvh8_nodes", ctypes.byref(size))
        lib.grt_built_free(handle)
        print(it, n, "ok" if rc == 0 else "learner rejected", size.value // 80, "nodes", flush=True)


def scenes(rng, iterations):
    d = tempfile.mkdtemp()

    def obj(path, n, poison):
        v = rng.uniform(-1, 1, (n * 3, 3)) * rng.choice([1.0, 1e-4, 1e4])
        lines = []
        for p in v:
            p = list(p)
            if poison and rng.random() < 0.02:
                p[rng.integers(3)] = rng.choice([0.0, 1e-30, 1e15, -1e15, 1e18, 1e-38, 3e38, float("inf"), float("nan")])
            lines.append("v %s %s %s" % tuple(repr(float(x)) for x in p))
        for t in range(n):
            a = 3 * t + 1
            lines.append("f %d %d %d" % ((a, a, a) if rng.random() < 0.05 else (a, a + 1, a + 2)))
        open(path, "w").write("\n".join(lines) + "\n")
    for it in range(iterations):
        xml = ('<scene version="0.5.0"><integrator type="path"><integer name="maxDepth" value="4"/></integrator><sensor type="perspective"><float name="fov" value="50"/>'
               '<transform name="toWorld"><lookat origin="0, 2, 6" target="0, 0.7, 0" up="0, 1, 0"/></transform></sensor>')
        for m in range(int(rng.integers(2, 5))):
            obj(os.path.join(d, "m%d.obj" % m), int(rng.integers(1, 400)), poison=rng.random() < 0.7)
            xml += '<shape type="obj"><string name="filename" value="m%d.obj"/><bsdf type="diffuse"><rgb name="reflectance" value="0.7, 0.6, 0.5"/></bsdf></shape>' % m
        xml += ('<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><scale value="0.4"/><translate x="-1.5" y="3"/></transform>'
                '<emitter type="area"><rgb name="radiance" value="20, 5, 5"/></emitter></shape></scene>')
        open(os.path.join(d, "s.xml"), "w").write(xml)
        grt.config_reset(); grt.config_set(static_slot_learning_rays=20000)
        t0 = time.time()
        try:
            scene = grt.Scene(os.path.join(d, "s.xml")); pt = grt.Pathtracer(scene, 64, 48, device=-1); pt.update()
            members = pt.static_geometry_members
            pt.close(); scene.close()
            print(it, "ok, %d members flattened, %.2f s" % (members, time.time() - t0), flush=True)
        except RuntimeError as e:
            print(it, "rejected:", str(e)[:100], flush=True)


if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    iterations = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    what = sys.argv[3] if len(sys.argv) > 3 else "soups"
    (soups if what == "soups" else scenes)(rng, iterations)
