#!/usr/bin/env python3
"""Regenerates the golden fixtures. Needs /root/reference (for oracle/_ref) -- run in the build
container only; the fixtures it writes are what travels to the GPU box.

bvh_golden.json   ("variants": the same for the binary trees the device gets with bvh_type = BVH / SBVH,
                  leaf-collapsed as the reference does for file-loaded meshes, and their BVH4 form)
                  sha256 of the BVH2 nodes / BVH8 nodes / BVH8 indices (and, separately, the BVH4 nodes) that the REFERENCE'S OWN
                  builder (oracle/_ref/libref_bvh.so = /root/reference/Src/BVH compiled verbatim)
                  produces for every Cornell mesh, every Sponza mesh (one aggregate digest plus the
                  10 largest individually) and three seeded random triangle soups.
reference_kernels_golden.npz  frames + queue sizes rendered by the reference's own Pathtracer.cu running on the CPU
                  (oracle/ref/ref_cuda_harness.cpp): Cornell with and without NEE, textured Sponza.
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


def slivers(seed, n):
    """Long thin overlapping triangles: the case spatial splits exist for."""
    rng = np.random.default_rng(seed)
    p0 = (rng.random((n, 3)) * 10).astype(np.float32)
    t = np.zeros((n, 24), np.float32)
    t[:, 0:3] = p0
    t[:, 3:6] = p0 + (rng.random((n, 3)) * 8 - 4).astype(np.float32)
    t[:, 6:9] = p0 + (rng.random((n, 3)) * 0.5).astype(np.float32)
    return t


VARIANT_SOUPS = (("soup", 12, 257), ("soup", 13, 5000), ("slivers", 3, 400), ("slivers", 4, 1500))
SPONZA_SBVH_STRIDE = 32   # every 32nd Sponza mesh goes through the (slow) spatial-split builder


def variant_digests(tris, sbvh_alpha=10e-5):
    out = {}
    for sbvh in (0, 1):
        for collapse in (0, 1):
            ref = oracle.ref_build_binary_variant(tris, sbvh, collapse, sbvh_alpha)
            out["%s_%s" % ("sbvh" if sbvh else "sah", "collapsed" if collapse else "raw")] = {
                "sha256": digest(ref["bvh2_nodes"], ref["bvh2_indices"]), "sha256_bvh4": digest(ref["bvh4_nodes"]),
                "nodes": int(ref["bvh2_nodes"].size // 32), "indices": int(ref["bvh2_indices"].size)}
    return out


def make_variants():
    variants = {"soups": {}, "sponza": {}}
    for kind, seed, n in VARIANT_SOUPS:
        tris = soup(seed, n) if kind == "soup" else slivers(seed, n)
        variants["soups"]["%s_%d_%d" % (kind, seed, n)] = variant_digests(tris)
    variants["soups"]["slivers_3_400_alpha0"] = variant_digests(slivers(3, 400), 0.0)["sbvh_collapsed"]
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("sponza"))
    scene.wait_until_loaded()
    agg = hashlib.sha256()
    for m in range(scene.mesh_data_count):
        tris = scene.mesh_data_array(m, "triangles", np.float32)
        ref = oracle.ref_build_binary_variant(tris, 0, 1)
        agg.update(digest(ref["bvh2_nodes"], ref["bvh2_indices"], ref["bvh4_nodes"]).encode())
        if m % SPONZA_SBVH_STRIDE == 0:
            ref = oracle.ref_build_binary_variant(tris, 1, 1)
            variants["sponza"][str(m)] = {"sha256": digest(ref["bvh2_nodes"], ref["bvh2_indices"], ref["bvh4_nodes"]),
                                          "triangles": int(tris.size // 24), "indices": int(ref["bvh2_indices"].size)}
    variants["sponza_sah_collapsed_aggregate"] = agg.hexdigest()
    scene.close()
    return variants


OPTIMIZED_CASES = (("soup", 41, 300, 0, 4), ("soup", 42, 4000, 0, 2), ("slivers", 43, 500, 0, 4), ("slivers", 43, 500, 1, 3))


def make_optimized():
    """Digests of trees after the reference's BVHOptimizer (kind, seed, n, sbvh, max_batches): limited to its
    deterministic measure-driven batches."""
    out = {}
    for kind, seed, n, sbvh, batches in OPTIMIZED_CASES:
        tris = soup(seed, n) if kind == "soup" else slivers(seed, n)
        ref = oracle.ref_build_optimized(tris, sbvh, batches)
        plain = oracle.ref_build_binary_variant(tris, sbvh, 0)
        assert not np.array_equal(ref["bvh2_nodes"], plain["bvh2_nodes"])
        arrays = [ref["bvh2_nodes"], ref["bvh2_indices"], ref["bvh4_nodes"]] + ([] if sbvh else [ref["bvh8_nodes"], ref["bvh8_indices"]])
        out["%s_%d_%d_%s_%d" % (kind, seed, n, "sbvh" if sbvh else "sah", batches)] = digest(*arrays)
    return out


def make_reference_kernel_golden():
    """reference_kernels_golden.npz: frames and per-bounce queue sizes produced by the REFERENCE'S OWN device code
    (Src/CUDA/Pathtracer.cu compiled verbatim for the host, oracle/ref/ref_cuda_harness.cpp) -- the fixture that
    pins the oracle where oracle/_ref is not available."""
    out = {}
    cases = (("cornell", dict(num_bounces=5), 64, 48, 3), ("cornell_no_nee", dict(num_bounces=4, enable_next_event_estimation=0), 64, 48, 2),
             ("sponza", dict(num_bounces=3), 80, 45, 2))
    for name, config, w, h, samples in cases:
        grt.config_reset()
        scene = grt.Scene(grt.scene_path("cornellbox" if name.startswith("cornell") else "sponza"))
        grt.config_set(**config)
        pt = grt.Pathtracer(scene, w, h, device=-1); pt.update()
        view = oracle.SceneView(pt)
        ref = oracle.ReferenceFrame(view)
        queues = []
        for s in range(samples):
            rc = ref.render_sample(s)
            queues.append([rc[k][:8] for k in ("trace", "shadow", "diffuse")])
        out[name + "_image"] = ref.final[:, :w, :3].copy()
        out[name + "_queues"] = np.array(queues, np.int32)
        ref.close(); pt.close(); scene.close()
    grt.config_reset()
    np.savez_compressed(os.path.join(HERE, "reference_kernels_golden.npz"), **out)
    print("wrote reference_kernels_golden.npz")


def main():
    if "--only-reference-kernels" in sys.argv:
        make_reference_kernel_golden()
        return
    if "--only-optimized" in sys.argv:
        path = os.path.join(HERE, "bvh_golden.json")
        golden = json.load(open(path))
        golden["optimized"] = make_optimized()
        json.dump(golden, open(path, "w"), indent=1, sort_keys=True)
        print("updated optimized")
        return
    if "--only-variants" in sys.argv:
        path = os.path.join(HERE, "bvh_golden.json")
        golden = json.load(open(path))
        golden["variants"] = make_variants()
        json.dump(golden, open(path, "w"), indent=1, sort_keys=True)
        print("updated variants")
        return
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
    golden["variants"] = make_variants()
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
    make_reference_kernel_golden()
    print("wrote fixtures")


if __name__ == "__main__":
    main()
