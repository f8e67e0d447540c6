#!/usr/bin/env python3
"""Experiment input for wave_sim: ALL triangles of the scene in ONE bottom-level tree, no top-level tree (header field
tlas_count = 0 tells wave_sim to start inside the tree) -- how the flattened static geometry of DESIGN.md 4.6 was priced
before it was built.   export_merged.py <scene> <out.bin> [builder] [optimise]
builder: 0 = the per-mesh SAH sweep (BVH.cpp), 1 = the reference's SBVH (SBVH.cpp), 2 = StaticBVHBuilder (binned object +
spatial splits, all threads); optimise: 1 = Bittner's insertion optimiser on the binary tree; then the 8-wide collapse
(environment GRT_PRIMITIVE_COST: the collapse's cost of a triangle test relative to a node step)."""
import ctypes, os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpu_raytracer_amd as grt

name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/wave_sim_%s_merged.bin" % name
sbvh = int(sys.argv[3]) if len(sys.argv) > 3 else 0; optimize = int(sys.argv[4]) if len(sys.argv) > 4 else 0
w, h = 1920, 1080
grt.config_reset()
if os.environ.get("STATIC_PRESPLIT"): grt.config_set(static_presplit=float(os.environ["STATIC_PRESPLIT"]))
scene = grt.Scene(grt.scene_path(name)); scene.wait_until_loaded()
pt = grt.Pathtracer(scene, w, h, device=-1); pt.update()
tris = np.concatenate([scene.mesh_data_array(m, "triangles", np.float32).reshape(-1, 24) for m in range(scene.mesh_data_count)])
lib = grt.host_lib()
import time; t0 = time.time()
lib.grt_build_blas_variant.restype = ctypes.c_void_p; lib.grt_build_blas_variant.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
handle = lib.grt_build_blas_variant(np.ascontiguousarray(tris).ctypes.data, tris.shape[0], sbvh, optimize)
print("SAH + CWBVH over %d triangles: %.2f s" % (tris.shape[0], time.time() - t0))
def arr(n, dtype):
    size = ctypes.c_size_t(0); p = lib.grt_built_array(handle, n.encode(), ctypes.byref(size))
    return np.frombuffer((ctypes.c_char * size.value).from_address(p), dtype=dtype).copy()
nodes = arr("bvh8_nodes", np.uint8).reshape(-1, 80); idx = arr("bvh8_indices", np.int32)
t = tris[idx]
pos = np.concatenate([t[:, 0:3], t[:, 3:6] - t[:, 0:3], t[:, 6:9] - t[:, 0:3]], axis=1).astype(np.float32)   # p0, e1, e2 (host Triangle: positions first)
cam = np.frombuffer(bytes(pt.array("camera")), dtype=np.float32)[:15].copy()
with open(out, "wb") as f:
    f.write(struct.pack("6i", nodes.shape[0], pos.shape[0], 1, 0, w, h))
    f.write(nodes.tobytes()); f.write(pos.tobytes()); f.write(np.array([0x80000000 - (1 << 32)], np.int32).tobytes() if False else np.array([-2147483648], np.int32).tobytes()); f.write(np.zeros(12, np.float32).tobytes()); f.write(cam.tobytes())
print("wrote %s: %d nodes, %d triangles" % (out, nodes.shape[0], pos.shape[0]))
