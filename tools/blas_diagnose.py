#!/usr/bin/env python3
"""Which rays do the device-built and the host-built trees disagree on, and who is right? (float64 brute force over all
triangles of the scene, world space; Cornell box: every instance has the identity transform)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpu_raytracer_amd as grt
from oracle import binding as oracle
from conftest import make_pathtracer
from test_gpu_blas import rays_for, original_triangle_of

SCENE = sys.argv[1] if len(sys.argv) > 1 else "cornellbox"
EXTENT = 3.0 if SCENE == "cornellbox" else 60.0
res = {}
for device_blas in (1, 0):
    scene, pt = make_pathtracer(grt, SCENE, 320, 180, 0, device_blas=device_blas)
    view = oracle.SceneView(pt)
    o, d = rays_for(view, 320, 180, 3, EXTENT)
    hits, _ = grt.trace_rays(pt.ctx, o, d)
    tris = pt.array("triangles").view(np.float32).reshape(-1, 24).astype(np.float64)
    res[device_blas] = (hits.copy(), original_triangle_of(pt), tris)
    print("device_blas", device_blas, "identity roots", int((pt.array("mesh_bvh_root_indices") < 0).sum()), "of", scene.mesh_count)
    pt.close(); scene.close()
a, b = res[1][0], res[0][0]
ha, hb = a[:, 1] != 0xffffffff, b[:, 1] != 0xffffffff
only_dev, only_host = np.nonzero(ha & ~hb)[0], np.nonzero(~ha & hb)[0]
print("rays", len(a), "only device-built tree hits", len(only_dev), "only host-built tree hits", len(only_host))
both = ha & hb
ta, tb = a[:, 2].view(np.float32), b[:, 2].view(np.float32)
tri_a, tri_b = res[1][1][np.where(ha, a[:, 1], 0).astype(np.int64)], res[0][1][np.where(hb, b[:, 1], 0).astype(np.int64)]
print("both hit", int(both.sum()), "t bits differ", int((a[both, 2] != b[both, 2]).sum()), "max relative t difference", float((np.abs(ta[both] - tb[both]) / tb[both]).max()), "other original triangle", int((tri_a[both] != tri_b[both]).sum()), "uv differ where same triangle", int(((a[:, 3] != b[:, 3]) & both & (tri_a == tri_b)).sum()))
worst = np.nonzero(both)[0][np.argsort(-(np.abs(ta[both] - tb[both]) / tb[both]))[:5]]
for i in worst: print("  t differs: ray", int(i), "device-tree t", ta[i], "tri", tri_a[i], "host-tree t", tb[i], "tri", tri_b[i])
tris = res[0][2]
p0, e1, e2 = tris[:, 0:3], tris[:, 3:6], tris[:, 6:9]
def brute(i):
    O, D = o[:, i].astype(np.float64), d[:, i].astype(np.float64)
    h = np.cross(D, e2); det = (e1 * h).sum(1); f = 1.0 / det; s = O - p0; u = f * (s * h).sum(1)
    q = np.cross(s, e1); v = f * (q * D).sum(1); t = f * (e2 * q).sum(1)
    ok = (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0)
    if not ok.any(): return None
    k = np.argmin(np.where(ok, t, np.inf)); return float(t[k]), float(u[k]), float(v[k])
for label, idx, hits in (("only device", only_dev, a), ("only host", only_host, b)):
    for i in idx[:8]:
        print(label, "ray", int(i), "primary" if i < 320 * 180 else "random", "o", o[:, i], "d", d[:, i], "t", hits[i, 2:3].view(np.float32)[0], "brute force (t,u,v)", brute(i))

# dump what the device built (and the rays the two trees disagree on) for offline analysis
if SCENE == "cornellbox":
    scene, pt = make_pathtracer(grt, SCENE, 320, 180, 0, device_blas=1)
    bad = np.nonzero(both & (a[:, 2] != b[:, 2]))[0]
    np.savez(os.path.join(ROOT, "gpurun_out", "r03_blas_dump.npz"), nodes=pt.array("bvh8_nodes").view(np.uint8).reshape(-1, 80), triangles=pt.array("triangles").view(np.float32).reshape(-1, 24),
             roots=pt.array("mesh_bvh_root_indices"), reverse=pt.array("reverse_indices"), o=o[:, bad], d=d[:, bad], dev=a[bad], host=b[bad], tlas_indices=pt.array("tlas_indices"))
    pt.close(); scene.close()
