#!/usr/bin/env python3
"""Mutation fuzzing of the host-side file loaders (PLY, serialized, hair, OBJ, TGA, PPM here; the image decoders
have a stand-alone harness in the same spirit). Meant to run against a sanitizer build of the host library:

    make -C gpu-raytracer_amd host/libgrt_host_asan.so
    cp gpu-raytracer_amd/host/libgrt_host.so /tmp/keep.so && cp gpu-raytracer_amd/host/libgrt_host_asan.so gpu-raytracer_amd/host/libgrt_host.so
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" \
        python tools/fuzz_loaders.py <seed> <iterations>
    cp /tmp/keep.so gpu-raytracer_amd/host/libgrt_host.so

Every mutated file must either load or raise RuntimeError; a sanitizer report is a bug. Round 1 ran ~60 000
mutations across all formats; what they found is pinned by tests/test_loaders.py::test_hostile_inputs_are_rejected_not_trusted.
"""
import sys, os, glob, numpy as np, tempfile, struct, ctypes
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import gpu_raytracer_amd as grt
from test_loaders import _ply_bytes, _serialized_archive
rng = np.random.default_rng(int(sys.argv[1]))
N = int(sys.argv[2])
d = tempfile.mkdtemp()
positions = np.round(rng.random((7, 3)) * 4 - 2, 3).astype(np.float32)
uvs = np.round(rng.random((7, 2)), 3)
faces = [[0, 1, 2], [2, 3, 4, 5], [1, 6, 5, 4, 3]]
seeds = {}
for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
    seeds["m_%s.ply" % fmt] = _ply_bytes(fmt, positions, None, uvs, faces, "int", True)
quad = dict(name="quad", double=True, positions=[[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], normals=[[0, 0, 1]] * 4, uvs=[[0, 0], [1, 0], [1, 1], [0, 1]], faces=[[0, 1, 2], [0, 2, 3]])
seeds["a.serialized"] = _serialized_archive([quad, quad], 4)
seeds["b.serialized"] = _serialized_archive([quad], 3)
seeds["h.hair"] = b"0 0 0\n0 1 0\n0.2 2 0\n\n1 0 0\n1 1 1\n\n"
seeds["hb.hair"] = b"BINARY_HAIR" + struct.pack("<I", 5) + np.array([[0,0,0],[0,1,0],[0,2,0]], np.float32).tobytes() + struct.pack("<f", np.inf) + np.array([[1,0,0],[1,1,1]], np.float32).tobytes() + struct.pack("<f", np.inf)
seeds["t.obj"] = b"v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3/1/1\nf -1 -2 -3\nf 1//1 2//1 4//1 3//1\n"
tga = bytes([0,0,2, 0,0,0,0,0, 0,0,0,0, 4,0,3,0, 24,0]) + bytes(rng.integers(0,256,36).astype(np.uint8))
seeds["t.tga"] = tga
seeds["r.tga"] = bytes([0,0,10, 0,0,0,0,0, 0,0,0,0, 4,0,2,0, 32,8]) + bytes([0x83, 1,2,3,4, 0x03, 5,6,7,8, 9,10,11,12, 13,14,15,16, 17,18,19,20])
seeds["p.ppm"] = b"P6\n# c\n3 2\n255\n" + bytes(18)
def mutate(data):
    b = bytearray(data); kind = rng.integers(0, 4)
    if kind == 0:
        for _ in range(rng.integers(1, 6)): b[rng.integers(0, len(b))] = rng.integers(0, 256)
    elif kind == 1: b = b[:rng.integers(1, len(b))]
    elif kind == 2:
        i = rng.integers(0, len(b)); b[i:i] = bytes(rng.integers(0,256,rng.integers(1,12)).astype(np.uint8))
    else:
        i = rng.integers(0, max(1, len(b) - 4)); b[i:i+4] = bytes([255,255,255,255]) if rng.integers(0,2) else bytes(4)
    return bytes(b)
names = list(seeds); ok = bad = 0
grt.config_reset()
for it in range(N):
    name = names[it % len(names)]
    data = mutate(seeds[name])
    path = os.path.join(d, "f_" + name); open(path, "wb").write(data)
    try:
        if name.endswith((".tga", ".ppm")):
            grt.load_texture(path)
        else:
            if name.endswith(".serialized"):
                xml = os.path.join(d, "s.xml"); open(xml, "w").write('<scene version="0.5.0"><shape type="serialized"><string name="filename" value="f_%s"/><integer name="shapeIndex" value="%d"/></shape></scene>' % (name, it % 2)); target = xml
            elif name.endswith(".hair"):
                xml = os.path.join(d, "s.xml"); open(xml, "w").write('<scene version="0.5.0"><shape type="hair"><string name="filename" value="f_%s"/></shape></scene>' % name); target = xml
            else:
                target = path
            s = grt.Scene(target); s.wait_until_loaded(); s.close()
        ok += 1
    except RuntimeError:
        bad += 1
print("loaded", ok, "rejected", bad)
