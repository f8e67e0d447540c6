#!/usr/bin/env python3
"""What a seating of the flattened tree's children costs the benchmark's rays, counted on the CPU (no GPU): the benchmark scene is staged by a
host-only integrator, the oracle path-traces one sample and reports node steps / triangle tests per closest-hit ray and per shadow ray -- the
counters the device's counting kernels reproduce (tests/test_gpu_parity.py::test_trace_statistics_equal_the_oracle_counters).
Development tool on the CHECKER side (it uses oracle/, hence it lives under tests/): never imported by the product.
    python tests/slot_eval.py [width height] [key=value ...]      e.g.  static_slot_learning_rays=0 static_slot_assignment=0"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gpu_raytracer_amd as grt  # noqa: E402
from oracle import binding as oracle  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if "=" not in a]
    config = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
    w, h = (int(args[0]), int(args[1])) if len(args) >= 2 else (960, 540)
    scene = bench.build_scene(grt)
    if config:
        grt.config_set(**{k: (int(v) if float(v).is_integer() else v) for k, v in config.items()})
    t0 = time.time()
    pt = grt.Pathtracer(scene, w, h, device=-1); pt.update()
    setup = time.time() - t0
    view = oracle.SceneView(pt)
    frame = oracle.Frame(view)
    t0 = time.time()
    oc = frame.render_sample(0)
    c, s = oc.trace_stats, oc.shadow_stats
    print("%s | %dx%d | closest: %.3f nodes %.3f triangles per ray (%d rays) | shadow: %.3f / %.3f (%d rays) | cost %.2f | flatten %.2f s, sample %.1f s"
          % (" ".join("%s=%g" % kv for kv in config.items()) or "default", w, h, c.nodes / c.rays, c.triangles / c.rays, c.rays, s.nodes / max(s.rays, 1), s.triangles / max(s.rays, 1), s.rays,
             (2 * c.nodes + c.triangles + 2 * s.nodes + s.triangles) / (c.rays + s.rays), pt.static_geometry_build_seconds, time.time() - t0))
    pt.close(); scene.close()


if __name__ == "__main__":
    main()
