#!/usr/bin/env python3
"""Micro-benchmark of kernel_trace_bvh8 (BVH_TYPE=2: kernel_trace_bvh2) through rt_trace_rays: Sponza primary rays and seeded
incoherent bounce rays at several batch sizes (tail effects vs. steady-state throughput)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpu_raytracer_amd as grt

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [100_000, 777_600, 2_073_600, 8_000_000]
    grt.config_reset()
    bvh_type = int(os.environ.get("BVH_TYPE", "8"))   # 8: CWBVH kernels, 4: 4-wide BVH kernels, 2: binary-BVH kernels
    grt.config_set(bvh_type=bvh_type)
    scene = grt.Scene(grt.scene_path("sponza"))
    pt = grt.Pathtracer(scene, 1920, 1080, device=0); pt.update()
    prim_o, prim_d = [], []
    for off in range(0, 1920 * 1080, grt.RT_BATCH_SIZE):
        cnt = min(grt.RT_BATCH_SIZE, 1920 * 1080 - off)
        o, d, _ = grt.generate_rays(pt.ctx, 0, off, cnt)
        prim_o.append(o); prim_d.append(d)
    o, d = np.concatenate(prim_o, 1), np.concatenate(prim_d, 1)
    hits, ms = grt.trace_rays(pt.ctx, o, d, repeat=3)
    t = hits[:, 2].view(np.float32); ok = hits[:, 1] != 0xffffffff
    rng = np.random.default_rng(1)
    so = (o + d * np.where(ok, t, 1).astype(np.float32) * np.float32(0.999))[:, ok]
    sd = rng.normal(size=so.shape).astype(np.float32); sd /= np.linalg.norm(sd, axis=0)
    for n in sizes:
        idx = np.arange(n) % o.shape[1]
        _, ms_p = grt.trace_rays(pt.ctx, o[:, idx], d[:, idx], repeat=5)
        idx = np.arange(n) % so.shape[1]
        _, ms_s = grt.trace_rays(pt.ctx, so[:, idx], sd[:, idx], repeat=5)
        # shuffled secondary (no spatial coherence between neighbouring lanes at all)
        perm = rng.permutation(n)
        _, ms_r = grt.trace_rays(pt.ctx, so[:, idx][:, perm], sd[:, idx][:, perm], repeat=5)
        print("n=%9d primary %8.3f ms %8.1f Mrays/s | secondary %8.3f ms %8.1f Mrays/s | shuffled %8.3f ms %8.1f Mrays/s" % (n, ms_p, n / ms_p / 1e3, ms_s, n / ms_s / 1e3, ms_r, n / ms_r / 1e3), flush=True)

if __name__ == "__main__":
    main()
