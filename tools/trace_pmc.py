#!/usr/bin/env python3
"""Workload for PMC collection: a few big rt_trace_rays launches on incoherent Sponza rays."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpu_raytracer_amd as grt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "secondary"
grt.config_reset()
scene = grt.Scene(grt.scene_path("sponza"))
pt = grt.Pathtracer(scene, 1920, 1080, device=0); pt.update()
o, d, _ = grt.generate_rays(pt.ctx, 0, 0, grt.RT_BATCH_SIZE)
hits, _ = grt.trace_rays(pt.ctx, o, d)
if mode == "secondary":
    t = hits[:, 2].view(np.float32); ok = hits[:, 1] != 0xffffffff
    rng = np.random.default_rng(1)
    o = (o + d * np.where(ok, t, 1).astype(np.float32) * np.float32(0.999))[:, ok]
    d = rng.normal(size=o.shape).astype(np.float32); d /= np.linalg.norm(d, axis=0)
idx = np.arange(n) % o.shape[1]
_, ms = grt.trace_rays(pt.ctx, o[:, idx], d[:, idx], repeat=3)
print("n", n, mode, "ms", ms, "Mrays/s", n / ms / 1e3)
