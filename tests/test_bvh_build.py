"""The product's SAH + CWBVH builder must reproduce the REFERENCE builder's bytes.

Golden digests (tests/golden/bvh_golden.json) come from the reference's own sources compiled
verbatim (oracle/_ref, see tests/golden/make_golden.py); where that library is present the
comparison is also made live, byte by byte.
"""
import hashlib
import json
import os

import numpy as np
import pytest

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bvh_golden.json")))


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


def product_build(grt, tris24):
    import ctypes
    lib = grt.host_lib()
    t = np.ascontiguousarray(tris24, np.float32)
    h = lib.grt_build_blas(t.ctypes.data, t.size // 24)
    assert h
    out = {}
    for name, dtype in (("bvh2_nodes", np.uint8), ("bvh2_indices", np.int32), ("bvh8_nodes", np.uint8), ("bvh8_indices", np.int32), ("bvh4_nodes", np.uint8)):
        n = ctypes.c_size_t()
        ptr = lib.grt_built_array(h, name.encode(), ctypes.byref(n))
        out[name] = np.frombuffer((ctypes.c_char * n.value).from_address(ptr), dtype=dtype).copy()
    lib.grt_built_free(h)
    return out


def slivers(seed, n):
    rng = np.random.default_rng(seed)
    p0 = (rng.random((n, 3)) * 10).astype(np.float32)
    t = np.zeros((n, 24), np.float32)
    t[:, 0:3] = p0
    t[:, 3:6] = p0 + (rng.random((n, 3)) * 8 - 4).astype(np.float32)
    t[:, 6:9] = p0 + (rng.random((n, 3)) * 0.5).astype(np.float32)
    return t


def product_device_bvh(grt, tris24, sbvh, collapse, sbvh_alpha=None, **config):
    """The binary tree (and its BVH4 form) the host hands to the device for bvh_type = BVH / SBVH."""
    import ctypes
    lib = grt.host_lib()
    grt.config_reset()
    grt.config_set(bvh_type=1 if sbvh else 2)
    if sbvh_alpha is not None:
        grt.config_set(sbvh_alpha=sbvh_alpha)
    grt.config_set(**config)
    t = np.ascontiguousarray(tris24, np.float32)
    h = lib.grt_build_device_bvh(t.ctypes.data, t.size // 24, int(collapse))
    grt.config_reset()
    assert h, lib.grt_last_error()
    out = {}
    for name, key, dtype in (("device_bvh2_nodes", "bvh2_nodes", np.uint8), ("device_bvh2_indices", "bvh2_indices", np.int32), ("device_bvh4_nodes", "bvh4_nodes", np.uint8)):
        n = ctypes.c_size_t()
        ptr = lib.grt_built_array(h, name.encode(), ctypes.byref(n))
        out[key] = np.frombuffer((ctypes.c_char * n.value).from_address(ptr), dtype=dtype).copy()
    lib.grt_built_free(h)
    return out


@pytest.mark.parametrize("key", sorted(k for k in GOLDEN["variants"]["soups"] if not k.endswith("alpha0")))
def test_sbvh_and_collapsed_trees_match_reference_digest(grt, key):
    """SBVHBuilder.cpp + BVHPartitions.cpp:103-282 (spatial splits, unsplitting) and BVHCollapser.cpp:
    node bytes, index lists (longer than the triangle count once a triangle is split) and the BVH4
    converted from them equal the verbatim reference build."""
    kind, seed, n = key.split("_")
    tris = soup(int(seed), int(n)) if kind == "soup" else slivers(int(seed), int(n))
    for variant, want in GOLDEN["variants"]["soups"][key].items():
        built = product_device_bvh(grt, tris, variant.startswith("sbvh"), variant.endswith("collapsed"))
        assert built["bvh2_nodes"].size // 32 == want["nodes"] and built["bvh2_indices"].size == want["indices"], variant
        assert digest(built["bvh2_nodes"], built["bvh2_indices"]) == want["sha256"], variant
        assert digest(built["bvh4_nodes"]) == want["sha256_bvh4"], variant
    if kind == "slivers":
        assert GOLDEN["variants"]["soups"][key]["sbvh_raw"]["indices"] > int(n)   # the fixture does exercise spatial splits


def test_full_sbvh_alpha_zero_matches_reference_digest(grt):
    want = GOLDEN["variants"]["soups"]["slivers_3_400_alpha0"]
    built = product_device_bvh(grt, slivers(3, 400), True, True, sbvh_alpha=0.0)
    assert digest(built["bvh2_nodes"], built["bvh2_indices"]) == want["sha256"] and digest(built["bvh4_nodes"]) == want["sha256_bvh4"]


def test_sponza_device_trees_match_reference_digest(grt):
    """File-loaded meshes: the SAH tree is leaf-collapsed before it (or its BVH4) reaches the device
    (AssetManager.cpp:85-87) -- all 383 Sponza meshes; the spatial-split tree for every 32nd of them."""
    want = GOLDEN["variants"]
    grt.config_reset()
    grt.config_set(bvh_type=4)
    scene = grt.Scene(grt.scene_path("sponza"))
    pt = grt.Pathtracer(scene, 8, 8, device=-1)      # init_geometry builds the device variants
    agg = hashlib.sha256()
    for m in range(scene.mesh_data_count):
        agg.update(digest(scene.mesh_data_array(m, "device_bvh2_nodes", np.uint8), scene.mesh_data_array(m, "device_bvh2_indices", np.int32),
                          scene.mesh_data_array(m, "device_bvh4_nodes", np.uint8)).encode())
    assert agg.hexdigest() == want["sponza_sah_collapsed_aggregate"]
    for m, entry in want["sponza"].items():
        tris = scene.mesh_data_array(int(m), "triangles", np.float32)
        built = product_device_bvh(grt, tris, True, True)
        assert built["bvh2_indices"].size == entry["indices"]
        assert digest(built["bvh2_nodes"], built["bvh2_indices"], built["bvh4_nodes"]) == entry["sha256"], m
    pt.close(); scene.close(); grt.config_reset()


@pytest.mark.parametrize("key", sorted(GOLDEN["soups"]))
def test_triangle_soups_match_reference_digest(grt, key):
    seed, n = map(int, key.split("_"))
    built = product_build(grt, soup(seed, n))
    assert built["bvh2_nodes"].size // 32 == GOLDEN["soups"][key]["bvh2_nodes"]
    assert built["bvh8_nodes"].size // 80 == GOLDEN["soups"][key]["bvh8_nodes"]
    assert digest(built["bvh2_nodes"], built["bvh2_indices"], built["bvh8_nodes"], built["bvh8_indices"]) == GOLDEN["soups"][key]["sha256"]
    assert digest(built["bvh4_nodes"]) == GOLDEN["soups"][key]["sha256_bvh4"]   # BVH4Converter.cpp, 128-B nodes


@pytest.mark.parametrize("scene_name", ["cornellbox", "sponza"])
def test_scene_blas_match_reference_digest(grt, scene_name):
    grt.config_reset()
    scene = grt.Scene(grt.scene_path(scene_name))
    scene.wait_until_loaded()
    want = GOLDEN["meshes"][scene_name]
    assert scene.mesh_data_count == want["mesh_data_count"]
    agg = hashlib.sha256()
    agg4 = hashlib.sha256()
    for m in range(scene.mesh_data_count):
        d = digest(scene.mesh_data_array(m, "bvh2_nodes", np.uint8), scene.mesh_data_array(m, "bvh2_indices", np.int32),
                   scene.mesh_data_array(m, "bvh8_nodes", np.uint8), scene.mesh_data_array(m, "bvh8_indices", np.int32))
        agg.update(d.encode())
        agg4.update(digest(scene.mesh_data_array(m, "bvh4_nodes", np.uint8)).encode())
        if str(m) in want["individual"]:
            assert d == want["individual"][str(m)]["sha256"], "mesh %d" % m
    assert agg.hexdigest() == want["aggregate"]
    assert agg4.hexdigest() == want["aggregate_bvh4"]
    scene.close()


def test_live_against_reference_builder(grt, oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    for seed, n in ((11, 3), (12, 257), (13, 5000)):
        tris = soup(seed, n)
        ref, built = oracle.ref_build(tris), product_build(grt, tris)
        for key in ("bvh2_nodes", "bvh2_indices", "bvh8_nodes", "bvh8_indices", "bvh4_nodes"):
            assert np.array_equal(ref[key], built[key]), key
    for tris in (soup(31, 700), slivers(32, 300)):
        for sbvh in (0, 1):
            for collapse in (0, 1):
                ref, built = oracle.ref_build_binary_variant(tris, sbvh, collapse), product_device_bvh(grt, tris, sbvh, collapse)
                for key in ("bvh2_nodes", "bvh2_indices", "bvh4_nodes"):
                    assert np.array_equal(ref[key], built[key]), (sbvh, collapse, key)


def optimized_build(grt, tris, sbvh, batches):
    if sbvh:
        return product_device_bvh(grt, tris, 1, 0, enable_bvh_optimization=1, bvh_optimizer_max_num_batches=batches)
    grt.config_reset()
    grt.config_set(enable_bvh_optimization=1, bvh_optimizer_max_num_batches=batches)
    built = product_build(grt, tris)
    grt.config_reset()
    return built


@pytest.mark.parametrize("key", sorted(GOLDEN["optimized"]))
def test_optimized_trees_match_reference_digest(grt, key):
    """BVHOptimizer::optimize (reference BVHOptimizer.cpp:225-417) while it selects nodes by measure -- the part of it
    that does not depend on the wall clock -- and the device trees converted from its result."""
    kind, seed, n, flavour, batches = key.split("_")
    tris = soup(int(seed), int(n)) if kind == "soup" else slivers(int(seed), int(n))
    built = optimized_build(grt, tris, flavour == "sbvh", int(batches))
    arrays = [built["bvh2_nodes"], built["bvh2_indices"], built["bvh4_nodes"]] + ([] if flavour == "sbvh" else [built["bvh8_nodes"], built["bvh8_indices"]])
    assert digest(*arrays) == GOLDEN["optimized"][key]


def test_optimizer_live_against_reference(grt, oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    for tris, sbvh, batches in ((soup(51, 40), 0, 4), (soup(52, 1500), 0, 0), (soup(52, 1500), 0, 3), (slivers(53, 350), 1, 4), (soup(54, 9), 0, 4)):
        ref, built = oracle.ref_build_optimized(tris, sbvh, batches), optimized_build(grt, tris, sbvh, batches)
        for key in ("bvh2_nodes", "bvh2_indices", "bvh4_nodes") + (() if sbvh else ("bvh8_nodes", "bvh8_indices")):
            assert np.array_equal(ref[key], built[key]), (sbvh, batches, key)


def test_optimizer_survives_areas_that_overflow(grt):
    """Coordinates beyond ~1e19 make every surface area infinite: no insertion point wins the search, and the node is
    paired with the root instead of indexing with INVALID."""
    for scale in (1e25, 3e37):
        tris = soup(71, 300)
        with np.errstate(over="ignore"):
            tris[:, :9] = (tris[:, :9].astype(np.float64) * scale).astype(np.float32)
        built = optimized_build(grt, tris, False, 3)
        assert sorted(built["bvh8_indices"].tolist()) == list(range(300)) and built["bvh2_nodes"].size // 32 == 600


def bvh2_sah_cost(nodes_bytes):
    """bvh_sah_cost of BVHOptimizer.cpp:15-35 with the default costs (node 4, leaf 1)."""
    raw = np.frombuffer(nodes_bytes.tobytes(), np.uint8).reshape(-1, 32)
    box = raw[:, :24].copy().view(np.float32).astype(np.float64)
    meta = raw[:, 24:32].copy().view(np.uint32)
    count = meta[:, 1] & 0x3fffffff
    d = box[:, 3:6] - box[:, 0:3]
    area = 2 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0])
    keep = np.ones(len(raw), bool); keep[1] = False
    leaf = keep & (count > 0)
    return (4.0 * area[keep & ~leaf].sum() + (area[leaf] * count[leaf]).sum()) / area[0]


def test_full_optimization_lowers_sah_cost_and_keeps_the_tree_valid(grt):
    """Unlimited run (measure-driven and random batches until ten stall): every triangle still in exactly one leaf,
    every inner box the union of its children's, children in even/odd pairs, and a cheaper tree."""
    tris = soup(61, 2500)
    plain = product_build(grt, tris)
    built = optimized_build(grt, tris, False, 1000)
    again = optimized_build(grt, tris, False, 1000)
    assert np.array_equal(built["bvh2_nodes"], again["bvh2_nodes"])            # fixed seed: reproducible, unlike the reference
    assert bvh2_sah_cost(built["bvh2_nodes"]) < 0.99 * bvh2_sah_cost(plain["bvh2_nodes"])   # (a uniform soup leaves a full sweep SAH build little to gain)
    assert sorted(built["bvh2_indices"].tolist()) == list(range(2500)) and sorted(built["bvh8_indices"].tolist()) == list(range(2500))

    raw = built["bvh2_nodes"].reshape(-1, 32)
    box = raw[:, :24].copy().view(np.float32)
    meta = raw[:, 24:32].copy().view(np.uint32)
    left, count, axis = meta[:, 0], meta[:, 1] & 0x3fffffff, meta[:, 1] >> 30
    seen_leaves, stack, visited = 0, [0], 0
    while stack:
        i = stack.pop(); visited += 1
        if count[i]:
            first = left[i]
            corners = tris[built["bvh2_indices"][first:first + count[i]]][:, :9].reshape(-1, 3)
            assert (corners >= box[i, :3] - 1e-4).all() and (corners <= box[i, 3:] + 1e-4).all()
            seen_leaves += count[i]
            continue
        l, r = int(left[i]), int(left[i]) + 1
        assert l % 2 == 0 and l >= 2
        assert np.array_equal(box[i, :3], np.minimum(box[l, :3], box[r, :3])) and np.array_equal(box[i, 3:], np.maximum(box[l, 3:], box[r, 3:]))
        assert int(axis[i]) < 3
        stack += [l, r]
    assert seen_leaves == 2500 and visited == len(raw) - 1


def test_cwbvh_structural_invariants(grt):
    """Asserts of the reference converter (BVH8Converter.cpp:21,252-254,293,303,322-323)."""
    built = product_build(grt, soup(5, 3000))
    nodes = built["bvh8_nodes"].reshape(-1, 80)
    assert built["bvh8_indices"].size == 3000 and sorted(built["bvh8_indices"].tolist()) == list(range(3000))
    meta = nodes[:, 24:32]
    imask = nodes[:, 15]
    for n in range(nodes.shape[0]):
        tri_total = 0
        for slot in range(8):
            m = int(meta[n, slot])
            if m == 0:
                continue
            if (m & 0x1f) >= 24:                      # inner child
                assert m == (0x20 | (24 + slot)) and (imask[n] >> slot) & 1
            else:                                     # leaf: unary count in the top 3 bits
                count = bin(m >> 5).count("1")
                assert 1 <= count <= 3 and (m >> 5) in (1, 3, 7) and (m & 0x1f) == tri_total
                tri_total += count
        assert tri_total <= 24
    child_base = nodes[:, 16:20].copy().view(np.uint32).reshape(-1)
    inner_counts = np.array([bin(int(x)).count("1") for x in imask])
    assert child_base[0] == 1 and (child_base + inner_counts <= nodes.shape[0]).all()
