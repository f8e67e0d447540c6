#!/usr/bin/env python3
"""Regenerates the golden fixtures. Needs /root/reference (for oracle/_ref) -- run in the build
container only; the fixtures it writes are what travels to the GPU box.

bvh_golden.json   sha256 of the BVH2 nodes / BVH8 nodes / BVH8 indices (and, separately, the BVH4 nodes) that the REFERENCE'S OWN
                  builder (oracle/_ref/libref_bvh.so = /root/reference/Src/BVH compiled verbatim)
                  produces for every Cornell mesh, every Sponza mesh (one aggregate digest plus the
                  10 largest individually) and three seeded random triangle soups.
render_golden.npz Cornell 48x48, 2 samples, 4 bounces rendered by the CPU oracle (regression pin for
                  the oracle itself and a fixed target for the GPU parity test).
"""
import hashlib, json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpu_raytracer_amd as grt
from oracle import binding as oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def soup(seed, n):
    rng = np.random.default_rng(seed)
    p0 = (rng.random((n, 3)) * 50).astype(np.float32)
    t = np.zeros((n, 24), np.float32)
    t[:, 0:3] = p0
    t[:, 3:6] = p0 + (rng.random((n, 3)) * 2 - 1).astype(np.float32)
    t[:, 6:9] = p0 + (rng.random((n, 3)) * 2 - 1).astype(np.float32)
    return t


def main():
    assert oracle.ref_lib() is not None, "oracle/_ref/libref_bvh.so missing: run `make -C oracle ref` where /root/reference exists"
    golden = {"source": "reference BVH builder compiled verbatim (oracle/_ref)", "meshes": {}}
    for name in ("cornellbox", "sponza"):
        grt.config_reset()
        scene = grt.Scene(grt.scene_path(name))
        scene.wait_until_loaded()
        agg = hashlib.sha256()
        agg4 = hashlib.sha256()
        sizes = []
        per_mesh = []
        for m in range(scene.mesh_data_count):
            tris = scene.mesh_data_array(m, "triangles", np.float32)
            ref = oracle.ref_build(tris)
            d = digest(ref["bvh2_nodes"], ref["bvh2_indices"], ref["bvh8_nodes"], ref["bvh8_indices"])
            agg.update(d.encode())
            agg4.update(digest(ref["bvh4_nodes"]).encode())
            per_mesh.append(d)
            sizes.append(tris.size // 24)
        entry = {"mesh_data_count": scene.mesh_data_count, "aggregate": agg.hexdigest(), "aggregate_bvh4": agg4.hexdigest(), "triangles": int(sum(sizes))}
        order = np.argsort(sizes)[::-1][:10] if name == "sponza" else range(scene.mesh_data_count)
        entry["individual"] = {str(int(m)): {"triangles": int(sizes[m]), "sha256": per_mesh[m]} for m in order}
        golden["meshes"][name] = entry
        scene.close()
    golden["soups"] = {}
    for seed, n in ((1, 1), (2, 7), (3, 1000), (4, 20000)):
        ref = oracle.ref_build(soup(seed, n))
        golden["soups"]["%d_%d" % (seed, n)] = {
            "sha256": digest(ref["bvh2_nodes"], ref["bvh2_indices"], ref["bvh8_nodes"], ref["bvh8_indices"]),
            "sha256_bvh4": digest(ref["bvh4_nodes"]),
            "bvh2_nodes": int(ref["bvh2_nodes"].size // 32), "bvh8_nodes": int(ref["bvh8_nodes"].size // 80)}
    json.dump(golden, open(os.path.join(HERE, "bvh_golden.json"), "w"), indent=1, sort_keys=True)

    grt.config_reset()
    scene = grt.Scene(grt.scene_path("cornellbox"))
    grt.config_set(num_bounces=4)
    pt = grt.Pathtracer(scene, 48, 48, device=-1)
    pt.update()
    view = oracle.SceneView(pt)
    frame = oracle.Frame(view)
    counters = []
    for s in range(2):
        c = frame.render_sample(s)
        counters.append(list(c.trace[:4]) + list(c.shadow[:4]))
    o, d, px = view.generate(0, 0, 48 * 48)
    hits, stats = view.trace(o, d)
    np.savez_compressed(os.path.join(HERE, "render_golden.npz"), image=frame.final[:, :48, :3].copy(), counters=np.array(counters, np.int32),
                        ray_origin=o, ray_direction=d, hits=hits, nodes=np.int64(stats.nodes), triangles=np.int64(stats.triangles))
    print("wrote fixtures")


if __name__ == "__main__":
    main()
