"""The device TLAS build (gpu-raytracer_amd/csrc/kernels_build.hip + rt_tlas_build.h).

CPU part (here, `-m "not gpu"`): the builder's node arithmetic is plain C++ shared with a one-thread restatement of the
whole build (oracle/oracle_tlas.cpp). That restatement is checked for what a TLAS must be -- every instance in exactly
one leaf, every child box containing the world boxes of its instances, inner children in consecutive node slots -- and
for what it is for: tracing it (oracle traversal) gives the closest hits of the host-built TLAS, whose builder is
byte-identical to the reference's (Integrator.cpp:399-430), ray for ray and bit for bit.
GPU part (tests/test_gpu_tlas.py): the kernel reproduces the restatement's bytes, and frames rendered with it match the
oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def instanced_scene_file(directory, count=60, seed=4):
    """`count` rotated / scaled / translated instances of one small mesh over a floor, two of them emitters."""
    from test_gpu_parity import blob_obj
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "blob.obj"), "w") as f:
        f.write(blob_obj(8))
    rng = np.random.default_rng(seed)
    shapes = ['<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="-90"/><scale value="30"/><translate y="-4"/></transform><bsdf type="diffuse"/></shape>',
              '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><scale value="6"/><translate y="16"/></transform><emitter type="area"><rgb name="radiance" value="18, 18, 18"/></emitter></shape>',
              '<shape type="rectangle"><transform name="toWorld"><rotate y="1" angle="90"/><scale value="3"/><translate x="-14" y="4"/></transform><emitter type="area"><rgb name="radiance" value="9, 12, 20"/></emitter></shape>']
    for i in range(count):
        x, y, z = rng.uniform(-12, 12), rng.uniform(-3, 8), rng.uniform(-12, 12)
        shapes.append('<shape type="obj"><string name="filename" value="blob.obj"/><transform name="toWorld"><scale value="%f"/><rotate y="1" angle="%f"/><rotate x="1" angle="%f"/>'
                      '<translate x="%f" y="%f" z="%f"/></transform><bsdf type="diffuse"/></shape>' % (rng.uniform(0.5, 1.8), rng.uniform(0, 360), rng.uniform(-40, 40), x, y, z))
    path = os.path.join(directory, "scene.xml")
    with open(path, "w") as f:
        f.write('<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="60"/><transform name="toWorld">'
                '<lookat origin="0, 6, 34" target="0, 1, 0" up="0, 1, 0"/></transform></sensor>%s</scene>' % "".join(shapes))
    return path


def decode_nodes(nodes):
    """(count, 80) uint8 CWBVH nodes -> per node: origin, scales, imask, bases, meta, quantised boxes"""
    words = nodes.view(np.uint32).reshape(-1, 20)
    p = words[:, 0:3].copy().view(np.float32)
    e = np.stack([(words[:, 3] >> (8 * d)) & 0xff for d in range(3)], axis=1)
    scale = (e.astype(np.uint32) << 23).view(np.float32)
    imask = (words[:, 3] >> 24) & 0xff
    meta = nodes[:, 24:32]
    q = nodes[:, 32:80].reshape(-1, 3, 2, 8)      # axis, (min, max), slot
    return p, scale, imask, words[:, 4], words[:, 5], meta, q


def check_tlas(nodes, order, world_boxes):
    """Structural invariants of a TLAS over len(order) instances (world_boxes in scene order: (n, 2, 3)), decoded here in
    numpy from the 80-byte node format alone -- nothing of the product's builder is used, so this is the independent half of
    the device-TLAS tests (tests/test_gpu_tlas.py runs it on the nodes the MI355X built). It restates what the reference's
    converter asserts of its own output (BVH8Converter.cpp:21,293,303,322-323): every primitive in exactly one leaf, an
    inner child's meta byte is 0b001xxxxx with its slot + 24 and its imask bit set, a leaf's unary count and offset stay
    below 24 and the leaf offsets of a node run 0, 1, 2, ... in slot order, inner children sit in consecutive node slots
    from base_index_child -- plus what traversal relies on: each quantised child box contains the boxes of everything below it."""
    n = len(order)
    assert sorted(order.tolist()) == list(range(n))                       # a permutation: every instance in exactly one leaf
    p, scale, imask, base_child, base_leaf, meta, q = decode_nodes(nodes)
    seen_nodes, seen_leaves = {0}, set()
    covered = {}                                                            # node -> instances below it
    def visit(k):
        below = []
        inner_rank = 0
        leaf_offsets = [int(meta[k, s]) & 31 for s in range(8) if int(meta[k, s]) and not (imask[k] >> s) & 1]
        assert leaf_offsets == list(range(len(leaf_offsets))) and len(leaf_offsets) <= 24          # num_triangles runs up in slot order (BVH8Converter.cpp:300-303)
        assert all(int(meta[k, s]) for s in range(8) if (imask[k] >> s) & 1)                       # an imask bit names a filled slot
        for s in range(8):
            m = int(meta[k, s])
            if m == 0:
                continue
            lo = p[k] + q[k, :, 0, s] * scale[k]; hi = p[k] + q[k, :, 1, s] * scale[k]
            if (imask[k] >> s) & 1:
                assert m == (0x20 | (24 + s))
                child = int(base_child[k]) + inner_rank; inner_rank += 1
                assert child not in seen_nodes and 0 < child < len(nodes)
                seen_nodes.add(child)
                inside = visit(child)
            else:
                assert (m >> 5) == 1 and (m & 31) < 24                      # one instance per TLAS leaf
                position = int(base_leaf[k]) + (m & 31)
                assert position not in seen_leaves and position < n
                seen_leaves.add(position)
                inside = [int(order[position])]
            for i in inside:                                                # the quantised child box contains its instances' world boxes
                assert (lo <= world_boxes[i, 0] + 1e-4 * np.abs(world_boxes[i, 0])).all() and (hi >= world_boxes[i, 1] - 1e-4 * np.abs(world_boxes[i, 1])).all(), (k, s, i)
            below += inside
        return below
    everything = visit(0)
    assert sorted(everything) == list(range(n)) and seen_leaves == set(range(n)) and seen_nodes == set(range(len(nodes)))


def world_boxes_of(transforms, boxes):
    t = transforms.reshape(-1, 3, 4); b = boxes.reshape(-1, 2, 3)
    corners = np.stack([np.stack([b[:, (c >> d) & 1, d] for d in range(3)], axis=1) for c in range(8)], axis=1)   # (n, 8, 3)
    world = np.einsum("nij,ncj->nci", t[:, :, :3], corners) + t[:, None, :, 3]
    return np.stack([world.min(axis=1), world.max(axis=1)], axis=1)


@pytest.mark.reference_layout
@pytest.mark.parametrize("count", [1, 2, 9, 60, 700])
def test_restated_device_tlas_is_a_valid_tlas_and_traces_like_the_host_built_one(grt, oracle, tmp_path, count):
    grt.config_reset()
    scene = grt.Scene(instanced_scene_file(str(tmp_path / "s"), count=max(count - 3, 0)) if count > 3 else instanced_scene_file(str(tmp_path / "s"), count=0))
    pt = grt.Pathtracer(scene, 96, 64, device=-1); pt.update()
    transforms = pt.array("scene_order_transforms").reshape(-1, 12).copy(); boxes = pt.array("scene_order_boxes").reshape(-1, 6).copy()
    if count <= 3:   # fewer instances than the scene has: build over a prefix (the structure checks only)
        transforms, boxes = transforms[:count], boxes[:count]
    n = transforms.shape[0]
    nodes, order = oracle.tlas_build(transforms, boxes)
    assert 1 <= len(nodes) <= max(1, n - 1) if n > 1 else len(nodes) == 1
    check_tlas(nodes, order, world_boxes_of(transforms, boxes))
    if count <= 3:
        pt.close(); scene.close(); return

    # the same rays through the host-built TLAS and through this one
    host = oracle.SceneView(pt)
    rng = np.random.default_rng(count)
    o, d, _ = host.generate(0, 0, 96 * 64)
    extra_o = rng.uniform(-14, 14, (3, 4000)).astype(np.float32); extra_d = rng.normal(size=(3, 4000)).astype(np.float32); extra_d /= np.linalg.norm(extra_d, axis=0)
    o = np.concatenate([o, extra_o], axis=1); d = np.concatenate([d, extra_d], axis=1)
    hits_host, _ = host.trace(o, d)
    host_indices = pt.array("tlas_indices").copy()

    mine = oracle.SceneView(pt)
    all_nodes = mine.keep["bvh8_nodes"].copy().reshape(-1, 80)
    all_nodes[:2 * n] = 0; all_nodes[:len(nodes)] = nodes
    tables = {name: pt.array("scene_order_" + name).copy() for name in ("roots", "materials", "transforms", "transforms_inv", "transforms_prev")}
    mine.keep["bvh8_nodes"] = np.ascontiguousarray(all_nodes.reshape(-1)); mine.scene.bvh8_nodes = mine.keep["bvh8_nodes"].ctypes.data
    for name, field, width in (("roots", "mesh_bvh_root_indices", 1), ("materials", "mesh_material_ids", 1), ("transforms", "mesh_transforms", 12), ("transforms_inv", "mesh_transforms_inv", 12), ("transforms_prev", "mesh_transforms_prev", 12)):
        a = np.ascontiguousarray(tables[name].reshape(n, width)[order].reshape(-1))
        mine.keep[field] = a; setattr(mine.scene, field, a.ctypes.data)
    hits_mine, _ = mine.trace(o, d)
    hit = hits_host[:, 1] != 0xffffffff
    assert hit.mean() > 0.3 and np.array_equal(hit, hits_mine[:, 1] != 0xffffffff)
    assert np.array_equal(hits_host[:, 1:], hits_mine[:, 1:])                                             # triangle, t bits, (u, v)
    assert np.array_equal(host_indices[hits_host[hit, 0].astype(np.int64)], order[hits_mine[hit, 0].astype(np.int64)])   # the same instance
    pt.close(); scene.close()
