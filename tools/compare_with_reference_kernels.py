#!/usr/bin/env python3
"""Renders a scene three ways from the same staged arrays and reports how they agree:
  * the reference's own device code (Src/CUDA/Pathtracer.cu, compiled verbatim for the host, oracle/_ref) on the CPU,
  * the restated oracle (oracle/*.cpp) on the CPU,
  * the HIP kernels on a GPU (with --device >= 0).

    python tools/compare_with_reference_kernels.py cornellbox -W 96 -H 64 -N 3 -b 5
    python tools/compare_with_reference_kernels.py path/to/scene.xml --device 0 --set enable_svgf=1

Kulla-Conty tables: read back from the GPU when there is one, otherwise smooth synthetic tables (all renderers get
the same ones)."""
import argparse, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpu_raytracer_amd as grt
from oracle import binding as oracle


def synthetic_luts(seed=5):
    rng = np.random.default_rng(seed)
    def smooth(shape):
        grid = np.meshgrid(*[np.linspace(0, 1, n) for n in shape], indexing="ij")
        return (0.55 + 0.35 * np.cos(sum((i + 1.3) * a for i, a in enumerate(grid)) * 1.7 + rng.random())).astype(np.float32)
    return [smooth((16, 16, 16)), smooth((16, 16, 16)), smooth((16, 16)), smooth((16, 16)), smooth((32, 32)), smooth((32,))]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scene", help="cornellbox | sponza | a scene file")
    ap.add_argument("-W", type=int, default=96); ap.add_argument("-H", type=int, default=64)
    ap.add_argument("-N", type=int, default=3, help="samples"); ap.add_argument("-b", type=int, default=5, help="bounces")
    ap.add_argument("--device", type=int, default=-1)
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="config_set entries")
    args = ap.parse_args()

    grt.config_reset()
    path = grt.scene_path(args.scene) if args.scene in ("cornellbox", "sponza") else args.scene
    scene = grt.Scene(path)
    config = {"num_bounces": args.b}
    for item in args.set:
        key, value = item.split("="); config[key] = value if not value.replace(".", "").lstrip("-").isdigit() else float(value)
    grt.config_set(**config)
    pt = grt.Pathtracer(scene, args.W, args.H, device=args.device); pt.update()
    luts = grt.read_luts(pt.ctx) if args.device >= 0 else synthetic_luts()
    view = oracle.SceneView(pt, luts=luts)
    ours, theirs = oracle.Frame(view), oracle.ReferenceFrame(view)
    nb = pt.device_config().num_bounces
    w = args.W
    for s in range(args.N):
        if s:
            pt.update()
        if args.device >= 0:
            pt.render(); gc = pt.counters()
        oc, rc = ours.render_sample(pt.sample_index), theirs.render_sample(pt.sample_index)
        ref = theirs.final[:, :w, :3]
        line = "sample %d  trace/bounce reference %s oracle %s" % (pt.sample_index, [int(v) for v in rc["trace"][:nb]], list(oc.trace[:nb]))
        line += "  | oracle vs reference rel L1 %.2e" % (np.abs(ours.final[:, :w, :3] - ref).sum() / ref.sum())
        if args.device >= 0:
            gpu = pt.read_framebuffer()[:, :w, :3]
            line += "  | gpu %s rel L1 vs reference %.2e, vs oracle %.2e" % (list(gc.trace[:nb]), np.abs(gpu - ref).sum() / ref.sum(), np.abs(gpu - ours.final[:, :w, :3]).sum() / ref.sum())
        print(line)
    theirs.close(); pt.close(); scene.close()


if __name__ == "__main__":
    main()
