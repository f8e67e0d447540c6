"""The BLAS build on the device (csrc/kernels_blas.hip, rt_build_geometry, config device_blas = 1).

What is checked, and against what:
  * the NODES the MI355X built, decoded here in numpy from the 80-byte format alone (nothing of the product's builder):
    every triangle in exactly one leaf, 1..3 triangles per leaf in unary, leaf offsets running up in slot order, inner
    children in consecutive node slots, every quantised child box containing the vertices of everything below it and having
    a thickness in every axis (the node test is strict: the first version of the build lost rays on axis-aligned walls) -- what
    the reference's converter asserts of its own output (BVH8Converter.cpp:21,293,303,322-323) plus what traversal needs;
  * HITS: the same rays through the device-built trees and through the host-built ones (whose builder is byte-identical
    to the reference's): the same instance, the same ORIGINAL triangle, t bit for bit and the same (u, v); rays that meet the
    shared edge of two triangles may resolve to the other one at a t one ulp away -- they are counted and bounded;
  * the oracle walking the very nodes the device built gives the device's hits bit for bit (instance ids included);
  * frames rendered on device-built trees match the oracle within the image tolerance, light sampling included (the
    light tables name triangles by index: the host remaps them through the build's permutation)."""
import numpy as np
import pytest

from conftest import make_pathtracer

pytestmark = pytest.mark.gpu

REL_L1_TOL = 1e-4


def check_blas(nodes, triangles, root, seen_triangles, is_reference=None):
    """Structural invariants of one BLAS (nodes: (n, 80) uint8 of the whole node array, triangles: (t, 24) float32 device
    triangles in leaf order); counts every triangle it finds in a leaf in `seen_triangles`, returns the number of nodes.
    is_reference[t]: triangle t is a copy in the flattened tree, i.e. possibly one PIECE of an original that early split clipping
    has cut (cpu_config.device_presplit): its leaf box holds the piece, so it lies inside the triangle's box instead of around it."""
    words = nodes.view(np.uint32).reshape(-1, 20)
    origin = words[:, 0:3].copy().view(np.float32)
    scale = (((words[:, 3:4] >> (8 * np.arange(3))) & 0xff).astype(np.uint32) << 23).view(np.float32)
    imask = (words[:, 3] >> 24) & 0xff
    meta = nodes[:, 24:32]
    q = nodes[:, 32:80].reshape(-1, 3, 2, 8)
    p0, e1, e2 = triangles[:, 0:3], triangles[:, 3:6], triangles[:, 6:9]
    corners = np.stack([p0, p0 + e1, p0 + e2], axis=1)                       # (t, 3, 3)
    tri_lo, tri_hi = corners.min(axis=1), corners.max(axis=1)
    visited = [0]

    def visit(k):
        visited[0] += 1
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        inner_rank, leaf_runs = 0, []
        for s in range(8):
            m = int(meta[k, s])
            if m == 0:
                assert not (imask[k] >> s) & 1
                continue
            lo = origin[k] + q[k, :, 0, s] * scale[k]; hi = origin[k] + q[k, :, 1, s] * scale[k]
            if (imask[k] >> s) & 1:
                assert m == (0x20 | (24 + s)), (k, s, m)
                child = int(words[k, 4]) + inner_rank; inner_rank += 1
                assert child > k and child < len(nodes)
                clo, chi = visit(child)
            else:
                unary, offset = m >> 5, m & 31
                assert unary in (1, 3, 7), (k, s, m)
                count = {1: 1, 3: 2, 7: 3}[unary]; leaf_runs.append((offset, count))
                first = int(words[k, 5]) + offset
                assert 0 <= first and first + count <= len(triangles)
                seen_triangles[first:first + count] += 1
                clo, chi = tri_lo[first:first + count].min(axis=0), tri_hi[first:first + count].max(axis=0)
                if is_reference is not None and is_reference[first:first + count].any():
                    # pieces: the box overlaps the triangles' and does not stick out of it by more than the quantisation grid and the padding of flat boxes
                    grid = scale[k] + 0.01
                    assert (hi > lo).all() and (lo >= clo - grid).all() and (hi <= chi + grid).all() and (hi >= clo).all() and (lo <= chi).all(), (k, s, lo, clo, hi, chi)
                    # what the nodes above have to contain is the pieces' own box, which this leaf box is the outward rounding of: less than one grid step of
                    # THIS node per side, one more where a flat box was given its thickness
                    # (an upper bound of the pieces' minimum and a lower bound of their maximum: all the assertions below need)
                    clo, chi = lo + 2 * scale[k], hi - 2 * scale[k]
            assert (hi > lo).all(), (k, s, lo, hi)      # the node test is `tmin < tmax`: a child box of zero thickness is never entered
            slack = 1e-5 * np.maximum(np.abs(clo), np.abs(chi)) + 1e-30
            assert (lo <= clo + slack).all() and (hi >= chi - slack).all(), (k, s, lo, clo, hi, chi)
            lo_all, hi_all = np.minimum(lo_all, clo), np.maximum(hi_all, chi)
        # the leaves of a node share its 24 triangle bits without gaps or overlaps (the builder deals the offsets in slot order; the learned seating, which gives
        # a node's children other slots afterwards, keeps every leaf's offset: the runs are checked sorted)
        expected_offset = 0
        for offset, count in sorted(leaf_runs):
            assert offset == expected_offset, (k, leaf_runs)
            expected_offset += count
        assert expected_offset <= 24
        return lo_all, hi_all

    visit(root)
    return visited[0]


def original_triangle_of(pt):
    reverse = pt.array("reverse_indices").copy()          # original triangle -> device triangle (one entry per ORIGINAL triangle)
    original = np.full(int(reverse.max()) + 1, -1, np.int64)
    original[reverse] = np.arange(len(reverse))
    return original


def rays_for(view, w, h, seed, extent):
    o, d, _ = view.generate(0, 0, w * h)
    rng = np.random.default_rng(seed)
    eo = rng.uniform(-extent, extent, (3, 30000)).astype(np.float32); ed = rng.normal(size=(3, 30000)).astype(np.float32); ed /= np.linalg.norm(ed, axis=0)
    return np.concatenate([o, eo], axis=1), np.concatenate([d, ed], axis=1)


@pytest.mark.parametrize("scene_name,extent", [("cornellbox", 3.0), ("sponza", 60.0)])
def test_device_built_trees_are_valid_and_trace_like_the_host_built_ones(grt, oracle, scene_name, extent):
    w, h = 320, 180
    results = {}
    for device_blas in (1, 0):
        scene, pt = make_pathtracer(grt, scene_name, w, h, 0, device_blas=device_blas)   # (the flattened tree of the device build: seated in the first update, like the host's)
        view = oracle.SceneView(pt)
        if device_blas:
            assert pt.device_blas_build_ms > 0.0
            nodes = pt.array("bvh8_nodes").view(np.uint8).reshape(-1, 80)
            triangles = pt.array("triangles").view(np.float32).reshape(-1, 24)
            roots = pt.array("mesh_bvh_root_indices") & 0x7fffffff
            # one tree per mesh data (instances share them). The triangles of different meshes interleave in leaf order -- leaf
            # positions are dealt level by level over all meshes at once --, so "every triangle in exactly one leaf" is a
            # statement about all trees together
            distinct_roots = sorted(set(int(r) for r in roots))
            # (a) every triangle in exactly one leaf, over ALL nodes behind the TLAS slots at once (vectorised)
            words = nodes.view(np.uint32).reshape(-1, 20)
            meta = nodes[2 * scene.mesh_count:, 24:32].astype(np.int64)
            is_leaf = (meta != 0) & (((words[2 * scene.mesh_count:, 3:4] >> 24) >> np.arange(8)) & 1 == 0)
            count = np.where(is_leaf, np.select([meta >> 5 == 1, meta >> 5 == 3, meta >> 5 == 7], [1, 2, 3], -100), 0)
            assert (count >= 0).all()
            first = words[2 * scene.mesh_count:, 5:6].astype(np.int64) + (meta & 31)
            seen_all = np.zeros(len(triangles) + 4, np.int32)
            for j in range(3):
                np.add.at(seen_all, (first + j)[count > j], 1)
            assert (seen_all[:len(triangles)] == 1).all() and seen_all[len(triangles):].sum() == 0, (int((seen_all == 0).sum()), int((seen_all > 1).sum()))
            # (b) the recursive check (child boxes, numbering, offsets) on a sample of the trees -- all of them on a small scene
            sample = distinct_roots if len(distinct_roots) <= 16 else distinct_roots[::8]
            seen = np.zeros(len(triangles), np.int32)
            aliases = pt.array("alias_mesh_ids")
            is_reference = (aliases >= 0) if len(aliases) == len(triangles) else np.zeros(len(triangles), bool)
            if pt.static_geometry_members:   # the flattened tree, built over the pieces of early split clipping: its boxes hold pieces
                flat_root = pt.static_geometry_top_levels[0]
                if flat_root not in sample: sample = list(sample) + [flat_root]
                assert scene_name != "sponza" or int(is_reference.sum()) > 1.005 * (len(triangles) - int(is_reference.sum()))   # Sponza at the default fraction 0.08: the cut triangles add 1.75 % references
            tree_nodes = sum(check_blas(nodes, triangles, root, seen, is_reference) for root in sample)
            assert seen.max() == 1 and tree_nodes <= len(nodes) - 2 * scene.mesh_count
        o, d = rays_for(view, w, h, 3, extent)
        hits, _ = grt.trace_rays(pt.ctx, o, d)
        want, _ = view.trace(o, d)                         # the oracle on the nodes this context traces
        assert np.array_equal(hits, want)
        original = original_triangle_of(pt)
        hit = hits[:, 1] != 0xffffffff
        triangle = np.where(hit, original[np.where(hit, hits[:, 1], 0).astype(np.int64)], -1)
        results[device_blas] = (hits.copy(), triangle, pt.array("tlas_indices").copy())
        pt.close(); scene.close()
    (a, tri_a, order_a), (b, tri_b, order_b) = results[1], results[0]
    hit_a, hit_b = a[:, 1] != 0xffffffff, b[:, 1] != 0xffffffff
    # (a flat triangle's box is padded as the reference pads it, AABB::fix_if_needed: a zero-thickness box is never entered by
    # the node test `tmin < tmax` -- the first version of the build had exact boxes and lost rays on axis-aligned walls)
    one_sided = hit_a != hit_b
    assert hit_b.mean() > 0.3 and one_sided.mean() < 1e-4, (float(hit_b.mean()), float(one_sided.mean()))
    hit = hit_a & hit_b
    # Where both trees find the same triangle, everything is bit-identical. Where a ray meets the shared edge of two triangles
    # (the diagonal of a Cornell wall: both accept it, their t differ in the last place) the tree decides: whichever is tested
    # first can pull the ray's range in far enough for the node of the other to be culled -- the two candidates differ by an
    # ulp of t, and such rays are a handful per thousand.
    ta, tb = a[:, 2].view(np.float32), b[:, 2].view(np.float32)
    relative = np.abs(ta[hit] - tb[hit]) / tb[hit]
    assert relative.max() < 1e-6, float(relative.max())
    other_triangle = (tri_a != tri_b) & hit
    assert other_triangle.mean() < 3e-3, float(other_triangle.mean())
    same = hit & ~other_triangle
    assert np.array_equal(order_a[a[same, 0].astype(np.int64)], order_b[b[same, 0].astype(np.int64)])   # the same instance
    assert np.array_equal(a[same, 2], b[same, 2]) and np.array_equal(a[same, 3], b[same, 3])     # t bit for bit, (u, v)
    grt.config_reset()


def test_frames_on_device_built_trees_match_the_oracle(grt, oracle):
    """Cornell box (area light: the light tables name triangles by index, remapped through the build's permutation) and
    Sponza, rendered on trees the device built, against the oracle reading those trees back: queue sizes and images."""
    for scene_name, w, h, bounces in (("cornellbox", 160, 120, 5), ("sponza", 256, 144, 4)):
        scene, pt = make_pathtracer(grt, scene_name, w, h, 0, device_blas=1, num_bounces=bounces)
        view = oracle.SceneView(pt)
        frame = oracle.Frame(view)
        for f in range(2):
            if f:
                pt.update()
            pt.render()
            c = pt.counters(); oc = frame.render_sample(pt.sample_index)
            for queue in ("trace", "shadow", "diffuse"):
                got, want = list(getattr(c, queue)[:bounces]), list(getattr(oc, queue)[:bounces])
                assert got[0] == want[0] and all(abs(x - y) <= 2 + 0.002 * y for x, y in zip(got, want)), (scene_name, f, queue, got, want)
            got, want = pt.read_framebuffer()[:, :w, :3], frame.final[:, :w, :3]
            assert np.abs(got - want).sum() / want.sum() < REL_L1_TOL, (scene_name, f)
        assert sum(c.shadow[:bounces]) > 1000
        pt.close(); scene.close()
    grt.config_reset()


def test_device_build_on_awkward_meshes(grt, oracle, tmp_path):
    """Meshes a builder can stumble over, each a shape of one scene: 1, 2, 3 and 4 triangles (the leaf size and one more), 40
    exact copies of one triangle (identical Morton codes: the cut falls back to the middle of the run, and every ray has 40
    candidates at the same t), a fan of 200 slivers sharing a vertex, an axis-aligned flat grid (zero-thickness boxes), a
    cloud whose triangles span 1e-3 .. 1e+2 in size, and instances of one mesh under rotation and non-uniform placement.
    Checked: the node invariants on every tree, hits through the device-built trees equal the oracle walking those trees and
    -- t bit for bit -- the hits through the host-built trees."""
    rng = np.random.default_rng(11)

    def obj(name, triangles):
        with open(tmp_path / name, "w") as f:
            for t in triangles:
                for v in t:
                    f.write("v %r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])))
            for i in range(len(triangles)):
                f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))

    def soup(n, size):
        centres = rng.uniform(-1, 1, (n, 1, 3))
        return centres + rng.normal(size=(n, 3, 3)) * size

    for n in (1, 2, 3, 4):
        obj("few%d.obj" % n, soup(n, 0.4))
    obj("copies.obj", np.repeat(soup(1, 0.5), 40, axis=0))
    fan = np.zeros((200, 3, 3)); angles = np.linspace(0, 2 * np.pi, 201)
    fan[:, 1, 0], fan[:, 1, 1] = np.cos(angles[:-1]), np.sin(angles[:-1]); fan[:, 2, 0], fan[:, 2, 1] = np.cos(angles[1:]), np.sin(angles[1:]); fan[:, 1:, 2] = 0.3
    obj("fan.obj", fan)
    gx, gz = np.meshgrid(np.arange(12.0), np.arange(12.0)); gx, gz = gx.ravel() / 6 - 1, gz.ravel() / 6 - 1
    grid = np.zeros((288, 3, 3)); s = 1 / 6
    for k in range(144):
        grid[2 * k] = [[gx[k], 0, gz[k]], [gx[k] + s, 0, gz[k]], [gx[k] + s, 0, gz[k] + s]]
        grid[2 * k + 1] = [[gx[k], 0, gz[k]], [gx[k] + s, 0, gz[k] + s], [gx[k], 0, gz[k] + s]]
    obj("grid.obj", grid)
    obj("scales.obj", np.concatenate([soup(300, 1e-3), soup(40, 0.3), soup(3, 30.0)]))
    shapes, x = [], -9.0
    for name in ("few1", "few2", "few3", "few4", "copies", "fan", "grid", "scales"):
        shapes.append('<shape type="obj"><string name="filename" value="%s.obj"/><transform name="toWorld"><translate x="%f"/></transform><bsdf type="diffuse"/></shape>' % (name, x)); x += 2.6
    for k in range(6):
        shapes.append('<shape type="obj"><string name="filename" value="scales.obj"/><transform name="toWorld"><scale value="%f"/><rotate y="1" angle="%f"/><rotate x="1" angle="%f"/><translate x="%f" y="%f" z="-6"/></transform><bsdf type="diffuse"/></shape>'
                      % (rng.uniform(0.05, 0.4), rng.uniform(0, 360), rng.uniform(0, 360), rng.uniform(-8, 8), rng.uniform(-2, 2)))
    (tmp_path / "s.xml").write_text('<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="70"/><transform name="toWorld">'
                                    '<lookat origin="0, 3, 14" target="0, 0, 0" up="0, 1, 0"/></transform></sensor>%s</scene>' % "".join(shapes))
    w, h = 256, 144
    results = {}
    for device_blas in (1, 0):
        grt.config_reset(); grt.config_set(device_blas=device_blas)
        scene = grt.Scene(str(tmp_path / "s.xml"))
        pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
        view = oracle.SceneView(pt)
        if device_blas:
            nodes = pt.array("bvh8_nodes").view(np.uint8).reshape(-1, 80)
            triangles = pt.array("triangles").view(np.float32).reshape(-1, 24)
            seen = np.zeros(len(triangles), np.int32)
            aliases = pt.array("alias_mesh_ids")   # copies in the flattened tree: early split clipping may have cut their originals into pieces (the 30-unit triangles of scales.obj)
            is_reference = (aliases >= 0) if len(aliases) == len(triangles) else np.zeros(len(triangles), bool)
            for root in sorted(set(int(r) & 0x7fffffff for r in pt.array("mesh_bvh_root_indices"))):
                check_blas(nodes, triangles, root, seen, is_reference)
            assert (seen == 1).all(), (int((seen == 0).sum()), int((seen > 1).sum()))
        o, d, _ = view.generate(0, 0, w * h)
        eo = rng.uniform(-10, 10, (3, 20000)).astype(np.float32) * np.array([[1.0], [0.3], [0.7]], np.float32)
        ed = rng.normal(size=(3, 20000)).astype(np.float32); ed /= np.linalg.norm(ed, axis=0)
        o, d = np.concatenate([o, eo], axis=1), np.concatenate([d, ed], axis=1)
        if device_blas:
            rays = (o, d)
        else:
            o, d = rays
        hits, _ = grt.trace_rays(pt.ctx, o, d)
        want, _ = view.trace(o, d)
        assert np.array_equal(hits, want)
        results[device_blas] = hits.copy()
        pt.close(); scene.close()
    a, b = results[1], results[0]
    hit = b[:, 1] != 0xffffffff
    assert 0.05 < hit.mean() and np.array_equal(hit, a[:, 1] != 0xffffffff)
    assert np.array_equal(a[hit, 2], b[hit, 2])          # t bit for bit (which of 40 identical copies is hit is the tree's choice)
    grt.config_reset()


def test_a_device_built_flattened_tree_is_seated_like_a_host_built_one(grt, oracle):
    """The device's collapse deals children to octant slots by centre, as the reference's does; the learned seating (host/SlotOrder.cpp) only needs a tree's nodes
    and the triangles of its leaves: the flattened tree the device built is read back, seated for the camera and its nodes go back in place (rt_update_nodes) --
    inside the integrator's first update(). Same hits as the unseated tree (t to the bit), fewer node steps; the device walks what the host view shows."""
    w, h = 640, 360
    results = {}
    for name, config in (("seated", dict(device_blas=1)), ("unseated", dict(device_blas=1, static_slot_learning_rays=0))):
        scene, pt = make_pathtracer(grt, "sponza", w, h, 0, **config)
        assert pt.static_geometry_whole_scene and pt.device_blas_build_ms > 0.0
        assert pt.reseats_completed == (1 if name == "seated" else 0) and not pt.reseat_pending
        view = oracle.SceneView(pt)
        o, d, _ = view.generate(0, 0, w * h)
        hits, _ = grt.trace_rays(pt.ctx, o, d)
        want, stats = view.trace(o, d)
        assert np.array_equal(hits, want)                          # the device walks the tree the host view shows (the re-seated nodes went to both)
        results[name] = (hits, stats.nodes / stats.rays)
        pt.close(); scene.close()
    a, b = results["seated"][0], results["unseated"][0]
    assert np.array_equal(a[:, 2], b[:, 2])                        # the same distance for every ray, to the bit
    assert (a[:, 1] != b[:, 1]).sum() <= 1e-3 * len(a)
    assert results["seated"][1] < 0.97 * results["unseated"][1], (results["seated"][1], results["unseated"][1])
    grt.config_reset()
