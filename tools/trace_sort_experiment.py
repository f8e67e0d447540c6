#!/usr/bin/env python3
"""How much would spatial ray reordering buy? Incoherent Sponza bounce rays traced (a) in queue
order, (b) sorted by Morton code of the origin (+ direction octant), (c) fully shuffled."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpu_raytracer_amd as grt

def part1by2(x):
    x = x.astype(np.uint64) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x

grt.config_reset()
scene = grt.Scene(grt.scene_path("sponza"))
pt = grt.Pathtracer(scene, 1920, 1080, device=0); pt.update()
os_, ds_ = [], []
for off in range(0, 1920 * 1080, grt.RT_BATCH_SIZE):
    cnt = min(grt.RT_BATCH_SIZE, 1920 * 1080 - off)
    o, d, _ = grt.generate_rays(pt.ctx, 0, off, cnt); os_.append(o); ds_.append(d)
o, d = np.concatenate(os_, 1), np.concatenate(ds_, 1)
hits, _ = grt.trace_rays(pt.ctx, o, d)
t = hits[:, 2].view(np.float32); ok = hits[:, 1] != 0xffffffff
rng = np.random.default_rng(1)
so = (o + d * np.where(ok, t, 1).astype(np.float32) * np.float32(0.999))[:, ok]
sd = rng.normal(size=so.shape).astype(np.float32); sd /= np.linalg.norm(sd, axis=0)
n = so.shape[1]
lo, hi = so.min(1, keepdims=True), so.max(1, keepdims=True)
for bits in (10, 5):
    q = np.clip(((so - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = (part1by2(q[0]) << 2) | (part1by2(q[1]) << 1) | part1by2(q[2])
    octant = ((sd[0] < 0).astype(np.uint64) << 2) | ((sd[1] < 0).astype(np.uint64) << 1) | (sd[2] < 0).astype(np.uint64)
    for name, k in (("morton%d" % bits, key), ("morton%d+octant" % bits, (key << 3) | octant), ("octant+morton%d" % bits, (octant << 40) | key)):
        order = np.argsort(k, kind="stable")
        _, ms = grt.trace_rays(pt.ctx, so[:, order], sd[:, order], repeat=5)
        print("%-18s n=%d %7.3f ms %8.1f Mrays/s" % (name, n, ms, n / ms / 1e3), flush=True)
_, ms = grt.trace_rays(pt.ctx, so, sd, repeat=5)
print("%-18s n=%d %7.3f ms %8.1f Mrays/s" % ("queue order", n, ms, n / ms / 1e3))
perm = rng.permutation(n)
_, ms = grt.trace_rays(pt.ctx, so[:, perm], sd[:, perm], repeat=5)
print("%-18s n=%d %7.3f ms %8.1f Mrays/s" % ("shuffled", n, ms, n / ms / 1e3))
