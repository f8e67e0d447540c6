"""Flattened static geometry (config merge_static, the default): the instances that stand in the scene with the identity
transform are copied into ONE extra bottom-level tree with a single TLAS leaf; hits on the copies are reported as the
scene's own instances and triangles (include/gpu_raytracer_amd.h: rt_upload_triangle_aliases). The reference has no such
thing -- it walks one BLAS per mesh under the TLAS (Integrator.cpp:101-283, 399-430) -- so the bar is: every ray finds
exactly what it finds in the reference's layout (merge_static = 0), which the other tests pin to the reference's own
kernels. CPU only: the staged arrays and the oracle's walk over them; tests/test_gpu_parity.py runs the HIP kernels over
the same arrays."""
import numpy as np
import pytest

from conftest import make_pathtracer, unpack_hits
from test_tlas import instanced_scene_file


def staged(grt, scene_file, w, h, merge, **config):
    grt.config_reset(); grt.config_set(merge_static=merge, **config)
    scene = grt.Scene(scene_file); grt.config_set(merge_static=merge, **config)
    pt = grt.Pathtracer(scene, w, h, device=-1); pt.update()
    return scene, pt


def rays_for(view, w, h, extent, count, seed):
    rng = np.random.default_rng(seed)
    o, d, _ = view.generate(0, 0, w * h)
    extra_o = rng.uniform(-extent, extent, (3, count)).astype(np.float32)
    extra_d = rng.normal(size=(3, count)).astype(np.float32); extra_d /= np.linalg.norm(extra_d, axis=0)
    return np.concatenate([o, extra_o], axis=1), np.concatenate([d, extra_d], axis=1)


def scene_file_of(grt, name, tmp_path):
    if name == "instances":   # a floor and two emitters, 40 rotated / scaled instances of one mesh
        return instanced_scene_file(str(tmp_path / "s"), count=40)
    if name == "everything":  # textured floor (a scaled file mesh), the same mesh again as a rotated emitter, a rectangle emitter, two spheres
        from scenes import write_scene_with_everything
        from test_loaders import _png_bytes
        return write_scene_with_everything(tmp_path, _png_bytes)
    return grt.scene_path(name)


@pytest.mark.parametrize("name,w,h,extent,identity_instances,flattened,merge", [
    ("cornellbox", 64, 48, 3.0, 8, 8, 1), ("sponza", 96, 54, 14.0, 382, 384, 1), ("sponza", 96, 54, 14.0, 382, 384, 2), ("sponza", 96, 54, 14.0, 382, 382, 3),
    ("instances", 96, 64, 14.0, 3, 3, 1),       # 40 instances of ONE mesh keep their TLAS leaves: an instanced mesh is not copied per instance
    ("everything", 80, 60, 6.0, 3, 5, 1)])      # two transformed instances of a mesh among the five: world-space copies
def test_flattened_static_geometry_traces_like_one_blas_per_mesh(grt, oracle, tmp_path, name, w, h, extent, identity_instances, flattened, merge):
    scene_file = scene_file_of(grt, name, tmp_path)
    scene_ref, pt_ref = staged(grt, scene_file, w, h, 0)
    reference = oracle.SceneView(pt_ref)
    o, d = rays_for(reference, w, h, extent, 6000, 5)
    hits_ref, stats_ref = reference.trace(o, d)
    shadow_ref = reference.trace_shadow(o, d, np.full(o.shape[1], 9.0, np.float32))[0]
    rows_ref = pt_ref.array("tlas_indices").copy(); roots_ref = pt_ref.array("mesh_bvh_root_indices").copy()
    triangles_ref = pt_ref.array("triangles").reshape(-1, 24).copy(); nodes_ref = pt_ref.array("bvh8_nodes").reshape(-1, 80).copy()
    assert pt_ref.static_geometry_members == 0 and pt_ref.array("alias_mesh_ids").size == 0
    materials_ref = np.zeros(scene_ref.mesh_count, np.int32); materials_ref[rows_ref] = pt_ref.array("mesh_material_ids")
    mesh_count = scene_ref.mesh_count
    assert not pt_ref.static_geometry_whole_scene
    pt_ref.close(); scene_ref.close()

    # ("instances": the 39 extra copies of the blob would cost 5 MB; with a 1 MB limit per mesh it stays instanced and the layout is the
    # mixed one -- a TLAS with a leaf for the flattened tree beside 40 instance leaves --, see test_the_flattening_policy_is_a_memory_budget)
    scene, pt = staged(grt, scene_file, w, h, merge, **(dict(static_mesh_copy_limit_mb=1) if name == "instances" else {}))
    members = pt.static_geometry_members
    identity = int((roots_ref < 0).sum())
    # merge_static 1 / 2: every instance (none has moved) whose copies fit the budget; 3: of those, the ones with the identity transform
    assert identity == identity_instances and members == flattened and members >= 2
    flat = oracle.SceneView(pt)

    # ---- the layout ----
    rows = pt.array("tlas_indices"); roots = pt.array("mesh_bvh_root_indices")
    leaves = mesh_count - members + 1
    assert rows.size == mesh_count + 1 == roots.size
    assert (rows[:leaves] == -1).sum() == 1 and (rows[leaves:] >= 0).all()                           # one TLAS leaf for the flattened tree, then a row per member
    assert sorted(rows[rows >= 0].tolist()) == list(range(mesh_count))                               # every scene instance has exactly one row
    triangles = pt.array("triangles").reshape(-1, 24); nodes = pt.array("bvh8_nodes").reshape(-1, 80)
    names_row, names_triangle = pt.array("alias_mesh_ids"), pt.array("alias_triangle_ids")            # one entry per device triangle, -1: not a copy
    assert names_row.size == triangles.shape[0] == names_triangle.size
    first = triangles_ref.shape[0]
    assert (names_row[:first] == -1).all() and (names_triangle[:first] == -1).all() and (names_row[first:] >= 0).all()   # host-built trees: the copies are the tail
    assert np.array_equal(triangles[:first].view(np.uint32), triangles_ref.view(np.uint32))          # everything the reference's layout holds is still there, untouched,
    assert np.array_equal(nodes[2 * mesh_count:nodes_ref.shape[0]], nodes_ref[2 * mesh_count:])      # ... the per-mesh trees included
    alias_rows, alias_triangles = names_row[first:], names_triangle[first:]
    assert (alias_rows >= leaves).all() and (alias_rows < rows.size).all() and (alias_triangles >= 0).all() and (alias_triangles < first).all()
    copies, originals = triangles[first:], triangles[alias_triangles]
    stands_as_loaded = roots[alias_rows] < 0                                                          # copies of identity instances ...
    assert np.array_equal(copies[stands_as_loaded].view(np.uint32), originals[stands_as_loaded].view(np.uint32))   # ... ARE their originals, bit for bit
    assert np.array_equal(copies[:, 9:].view(np.uint32), originals[:, 9:].view(np.uint32))           # the others: the original taken to world space
    placed = pt.array("mesh_transforms").reshape(-1, 3, 4)[alias_rows[~stands_as_loaded]]
    for part, translate in ((slice(0, 3), 1.0), (slice(3, 6), 0.0), (slice(6, 9), 0.0)):             # position_0, edge_1, edge_2
        want = np.einsum("nij,nj->ni", placed[:, :, :3], originals[~stands_as_loaded][:, part]) + translate * placed[:, :, 3]
        assert np.allclose(copies[~stands_as_loaded][:, part], want, rtol=1e-5, atol=1e-5)
    assert stands_as_loaded.all() == (members == identity)
    copied = np.bincount(alias_triangles, minlength=first)
    pairs = np.unique(np.stack([alias_rows, alias_triangles]), axis=1).shape[1]                      # (instance, triangle): once each without spatial splits;
    assert copied.min() >= 0 and (pairs == alias_rows.size if merge == 2 else (pairs < alias_rows.size or name != "sponza"))   # a triangle they cut, once per part (Sponza's long triangles)
    flat_root = int(roots[np.flatnonzero(rows == -1)[0]] & 0x7fffffff)
    # nothing outside the tree: no TLAS, node 0 is a copy of the tree's root and rays start inside it (rt_set_static_geometry)
    assert pt.static_geometry_whole_scene == (members == mesh_count)
    if members == mesh_count:
        assert leaves == 1 and rows[0] == -1 and np.array_equal(nodes[0], nodes[flat_root])
    assert flat_root == nodes_ref.shape[0] and roots[np.flatnonzero(rows == -1)[0]] < 0              # the extra tree sits behind the others; world space
    # each copy names a row whose BLAS holds the original: walk that BLAS' triangle range
    def blas_triangles(root, all_nodes):
        found, todo = [], [root]
        while todo:
            k = todo.pop()
            words = all_nodes[k].view(np.uint32); meta = all_nodes[k][24:32]; imask = int(words[3] >> 24)
            inner = 0
            for s in range(8):
                m = int(meta[s])
                if m == 0: continue
                if (imask >> s) & 1: todo.append(int(words[4]) + inner); inner += 1
                else: found += [int(words[5]) + (m & 31) + j for j in range(bin(m >> 5).count("1"))]
        return found
    for row in np.unique(alias_rows)[:40]:
        own = set(blas_triangles(int(roots[row] & 0x7fffffff), nodes))
        assert set(alias_triangles[alias_rows == row].tolist()) == own

    # ---- the rays ----
    hits, stats = flat.trace(o, d)
    mesh_ref, tri_ref, t_ref, u_ref, v_ref = unpack_hits(hits_ref); mesh, tri, t, u, v = unpack_hits(hits)
    hit = tri_ref != -1
    # An identity instance's copies are its triangles bit for bit, and the ray that meets them is the same world-space ray in
    # both layouts: the same distance, to the bit. A transformed instance's copies are its triangles taken to world space,
    # where the reference's layout takes the ray to object space instead: the same hit up to rounding (a few ulp of t; u, v
    # within one step of their 16-bit quantisation), and a ray that grazes an edge may fall on the other side of it.
    exact = hit & (roots_ref[mesh_ref] < 0)
    both = hit & (tri != -1)
    far_apart = np.zeros(hit.shape, bool); far_apart[both] = np.abs(t[both] - t_ref[both]) > 1e-5 * np.abs(t_ref[both])
    grazing = (hit != (tri != -1)) | (far_apart & ~exact)
    assert hit.mean() > 0.3 and grazing.sum() <= (0 if members == identity else 2e-3 * hit.sum()), int(grazing.sum())
    assert np.array_equal(t.view(np.uint32)[exact & ~grazing], t_ref.view(np.uint32)[exact & ~grazing])
    hit = hit & ~grazing
    # ... to the same triangle of the same scene instance -- except where two triangles lie at exactly the closest distance
    # (a box standing on the floor): a tie goes to whichever the walk meets first, in either layout
    tie = hit & (tri != tri_ref)
    assert tie.sum() <= 1e-3 * hit.sum()
    same = hit & ~tie
    assert np.array_equal(u[same & exact], u_ref[same & exact]) and np.array_equal(v[same & exact], v_ref[same & exact])
    worst = max(int(np.abs(u[same].astype(np.int64) - u_ref[same]).max()), int(np.abs(v[same].astype(np.int64) - v_ref[same]).max()))
    assert worst <= 8, worst                                                                          # of 65535 (a triangle 0.1 wide, 12 units from the origin: 1e-6 of 12 is 1e-4 of the triangle)
    assert np.array_equal(rows[mesh[same]], rows_ref[mesh_ref[same]])
    assert (tri[hit] < first).all() and (rows[mesh[hit]] >= 0).all()                                 # never a copy, never the flattened tree's own row
    by_scene_mesh = np.zeros(mesh_count, np.int32); by_scene_mesh[rows[rows >= 0]] = pt.array("mesh_material_ids")[rows >= 0]
    assert np.array_equal(by_scene_mesh, materials_ref)                                              # rows carry their instance's material
    shadow = flat.trace_shadow(o, d, np.full(o.shape[1], 9.0, np.float32))[0]
    assert (shadow != shadow_ref).sum() <= (0 if members == identity else 1e-3 * shadow.size) and 0.05 < shadow.mean() < 0.999
    if name == "sponza":   # what it is for: fewer nodes and no instance entries on the way to the same hits
        assert stats.nodes < stats_ref.nodes and stats.instances_identity < 0.2 * stats_ref.instances_identity
    pt.close(); scene.close(); grt.config_reset()


def test_frames_do_not_depend_on_the_flattening(grt, oracle):
    """One emitter, so the light tables hold the same single entry in both layouts: every sample draws the same random
    numbers, meets the same triangles and materials -- the rendered frames are the same floats."""
    frames = []
    for merge in (0, 1):
        scene, pt = staged(grt, grt.scene_path("cornellbox"), 40, 30, merge, num_bounces=4)
        view = oracle.SceneView(pt); frame = oracle.Frame(view)
        for s in range(2): frame.render_sample(s)
        frames.append(frame.accumulator(0).copy())
        pt.close(); scene.close()
    assert frames[0].max() > 0.5 and np.array_equal(frames[0], frames[1])
    grt.config_reset()


def test_the_flattened_tree_is_in_breadth_first_order(grt):
    """The first nodes of the flattened tree are its top levels (what every ray walks is one contiguous run). The builder's
    8-wide collapse emits nodes depth-first; bvh8_order_breadth_first renumbers them level by level (children stay consecutive
    in slot order, so traversal does not notice -- every other test of this file walks the renumbered tree)."""
    scene, pt = staged(grt, grt.scene_path("sponza"), 64, 36, 1)
    root, top = pt.static_geometry_top_levels
    nodes = pt.array("bvh8_nodes").view(np.uint32).reshape(-1, 20)
    count = nodes.shape[0] - root
    depth = np.full(count, -1, np.int64); depth[0] = 0
    reached = 1
    for n in range(count):                      # breadth-first: a node's children come after every node of its own level
        assert depth[n] >= 0
        word = nodes[root + n]
        children = bin(int(word[3]) >> 24).count("1")
        if children:
            first = int(word[4]) - root
            assert first == reached             # ... exactly where the nodes reached so far end
            depth[first:first + children] = depth[n] + 1
            reached += children
    assert reached == count and (np.diff(depth) >= 0).all()
    assert top == min(64, int((depth <= 2).sum())) and 9 < top <= 64
    assert np.array_equal(nodes[0], nodes[root])     # node slot 0: the copy of the root rays start in
    pt.close(); scene.close(); grt.config_reset()


def test_light_tables_keep_the_references_order_in_the_flattened_layout(grt, oracle, tmp_path):
    """Several emitters: which of them a random number selects depends on their ORDER in the cumulative distribution -- the
    reference's is the leaf order of its top-level tree (Pathtracer.cpp:503-534). The flattened layout has other instance rows
    but keeps that order (Integrator::reference_tlas_order), so NEE takes the same decisions and the frames of a scene with
    three emitters, one of them a rotated and scaled file mesh, are the same floats in both layouts (all instances here have
    hits that agree to the bit: identity transforms, or transformed ones that no bounce ray grazes differently)."""
    from test_tlas import instanced_scene_file
    path = instanced_scene_file(str(tmp_path / "s"), count=2)      # floor, two emitters of different power, two blobs
    tables, frames = [], []
    for merge in (0, 1):
        scene, pt = staged(grt, path, 48, 32, merge, num_bounces=3)
        assert (pt.static_geometry_members > 0) == bool(merge)
        rows = pt.array("tlas_indices")
        cdf = pt.array("light_mesh_cumulative_probability").copy(); spans = pt.array("light_mesh_triangle_span").copy()
        meshes = rows[pt.array("light_mesh_transform_indices")]     # scene mesh of every light entry, in distribution order
        tables.append((cdf, spans.reshape(-1, 2)[:, 1] - spans.reshape(-1, 2)[:, 0], meshes.copy()))
        view = oracle.SceneView(pt); frame = oracle.Frame(view)
        for s in range(2): frame.render_sample(s)
        frames.append(frame.accumulator(0).copy())
        pt.close(); scene.close()
    assert len(tables[0][0]) == 2 and np.array_equal(tables[0][0], tables[1][0]) and np.array_equal(tables[0][1], tables[1][1]) and np.array_equal(tables[0][2], tables[1][2])
    rel = np.abs(frames[0] - frames[1]).sum() / frames[0].sum()
    assert frames[0].max() > 0.1 and rel < 1e-5, rel
    grt.config_reset()


def test_a_member_that_moves_leaves_the_flattened_tree(grt, oracle):
    """The tree is rebuilt without it (here inside update(): set_flatten_asynchronously(False); the background rebuild has the next
    test); it keeps a TLAS leaf of its own even when it comes to rest."""
    scene, pt = staged(grt, grt.scene_path("cornellbox"), 48, 36, 1)
    pt.set_flatten_asynchronously(False)
    assert pt.static_geometry_members == 8 and pt.array("tlas_indices").size == 9 and pt.static_geometry_whole_scene
    triangles_before = pt.array("triangles").size // 24
    scene.set_mesh_transform(6, (0.25, 0.0, -0.1), (0.0, 0.0, 0.0, 1.0), 1.0)
    pt.invalidate("scene"); pt.update()
    rows = pt.array("tlas_indices").copy()
    assert pt.static_geometry_members == 7 and not pt.static_geometry_whole_scene and sorted(rows.tolist()[:2]) == [-1, 6] and sorted(rows[2:].tolist()) == [0, 1, 2, 3, 4, 5, 7]
    assert pt.array("triangles").size // 24 < triangles_before                 # its copies are gone
    view = oracle.SceneView(pt)
    o, d = rays_for(view, 48, 36, 3.0, 3000, 9)
    hits, _ = view.trace(o, d)
    scene.set_mesh_transform(6, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), 1.0)   # standing still again does not bring it back
    pt.invalidate("scene"); pt.update()
    assert pt.static_geometry_members == 7
    scene.set_mesh_transform(6, (0.25, 0.0, -0.1), (0.0, 0.0, 0.0, 1.0), 1.0)
    for other in (0, 1, 2, 3, 4, 5):                                          # everything but one instance moves: nothing left to flatten
        scene.set_mesh_transform(other, (0.0, 0.01, 0.0), (0.0, 0.0, 0.0, 1.0), 1.0)
    pt.invalidate("scene"); pt.update()
    assert pt.static_geometry_members == 0 and not pt.static_geometry_whole_scene and sorted(pt.array("tlas_indices").tolist()) == list(range(8))
    pt.close(); scene.close()

    scene, pt = staged(grt, grt.scene_path("cornellbox"), 48, 36, 0)
    scene.set_mesh_transform(6, (0.25, 0.0, -0.1), (0.0, 0.0, 0.0, 1.0), 1.0)
    pt.invalidate("scene"); pt.update()
    hits_ref, _ = oracle.SceneView(pt).trace(o, d)
    rows_ref = pt.array("tlas_indices").copy()
    hit = hits_ref[:, 1] != 0xffffffff
    assert np.array_equal(hits[:, 2], hits_ref[:, 2])                                                  # the same distance to the bit (the mover is walked in object space in both layouts)
    same = hit & (hits[:, 1] == hits_ref[:, 1])                                                        # (ties: the box that moved still stands ON the floor)
    assert same.sum() >= 0.999 * hit.sum() and np.array_equal(hits[same, 3], hits_ref[same, 3])
    assert np.array_equal(rows[hits[same, 0].astype(np.int64)], rows_ref[hits_ref[same, 0].astype(np.int64)])
    pt.close(); scene.close(); grt.config_reset()


def test_a_member_that_moves_is_taken_out_beside_the_frame_loop(grt, oracle):
    """The default: update() does NOT wait for the new tree. The frame in which the move is noticed, and every frame until a worker
    thread has built the tree of the members that are left, is staged in the reference's layout (a TLAS leaf per instance; their
    trees and triangles never left the device arrays) -- hits against the plain reference staging of the same scene: identical --;
    the first update() after the worker is done installs the flattened layout, with the tree the worker built."""
    import time
    scene, pt = staged(grt, grt.scene_path("cornellbox"), 48, 36, 1)
    assert pt.static_geometry_members == 8 and pt.reflatten_in_progress == 0
    scene.set_mesh_transform(6, (0.25, 0.0, -0.1), (0.0, 0.0, 0.0, 1.0), 1.0)
    pt.invalidate("scene")
    started = time.perf_counter(); pt.update(); waited = time.perf_counter() - started
    assert pt.static_geometry_members == 0 and pt.reflatten_in_progress in (1, 2) and sorted(pt.array("tlas_indices").tolist()) == list(range(8))
    assert pt.array("alias_mesh_ids").size == 0
    view = oracle.SceneView(pt)
    o, d = rays_for(view, 48, 36, 3.0, 3000, 9)
    interim, _ = view.trace(o, d)
    rows_interim = pt.array("tlas_indices").copy()
    deadline = time.time() + 30
    while pt.reflatten_in_progress != 2 and time.time() < deadline: time.sleep(0.01)
    assert pt.reflatten_in_progress == 2
    pt.update()                                                               # nothing was invalidated: the finished build alone brings the new layout in
    assert pt.static_geometry_members == 7 and pt.reflattens_completed == 1 and pt.reflatten_in_progress == 0
    rows = pt.array("tlas_indices").copy()
    assert sorted(rows.tolist()[:2]) == [-1, 6] and sorted(rows[2:].tolist()) == [0, 1, 2, 3, 4, 5, 7]
    flattened, _ = oracle.SceneView(pt).trace(o, d)
    hit = interim[:, 1] != 0xffffffff
    assert hit.mean() > 0.3 and np.array_equal(flattened[:, 2], interim[:, 2])           # the same distances to the bit in the interim and in the new layout
    same = hit & (flattened[:, 1] == interim[:, 1])
    assert same.sum() >= 0.999 * hit.sum() and np.array_equal(rows[flattened[same, 0].astype(np.int64)], rows_interim[interim[same, 0].astype(np.int64)])
    # a second mover while nothing is pending, then a third one WHILE the worker runs: its tree is out of date when it arrives and is not taken
    scene.set_mesh_transform(5, (0.0, 0.05, 0.0), (0.0, 0.0, 0.0, 1.0), 1.0); pt.invalidate("scene"); pt.update()
    scene.set_mesh_transform(4, (0.0, 0.05, 0.0), (0.0, 0.0, 0.0, 1.0), 1.0); pt.invalidate("scene"); pt.update()
    deadline = time.time() + 30
    while pt.reflatten_in_progress == 1 and time.time() < deadline: time.sleep(0.01)
    pt.update()
    assert pt.static_geometry_members == 5 and sorted(pt.array("tlas_indices")[4:].tolist()) == [0, 1, 2, 3, 7]
    final, _ = oracle.SceneView(pt).trace(o, d)
    assert (final[:, 1] != 0xffffffff).mean() > 0.3
    pt.close(); scene.close(); grt.config_reset()


def test_the_flattening_is_left_alone_where_it_does_not_apply(grt, tmp_path):
    for config in ({"bvh_type": 2}, {"bvh_type": 4}, {"device_tlas": 1}):          # CWBVH with a host-built TLAS only (device_tlas needs a device: no-op on the host)
        scene, pt = staged(grt, grt.scene_path("cornellbox"), 16, 16, 1, **config)
        if "bvh_type" in config:
            assert pt.static_geometry_members == 0 and pt.array("alias_mesh_ids").size == 0 and pt.array("tlas_indices").size == 8
        pt.close(); scene.close()
    grt.config_reset()


def test_the_flattening_policy_is_a_memory_budget(grt, tmp_path):
    """Which instances are copied into the flattened tree is decided in bytes (Config.h static_mesh_copy_limit_mb /
    static_copy_budget_mb; ~176 B per copied triangle), not by a count of instances: an instanced mesh joins while the copies
    BEYOND its first stay under the per-mesh limit, everything joins until the total budget is reached; what stays outside keeps
    its TLAS leaf. The extra device bytes are reported (static_geometry_bytes)."""
    from test_tlas import instanced_scene_file
    path = instanced_scene_file(str(tmp_path / "s"), count=40)          # floor, two emitters, 40 instances of one 768-triangle blob
    grt.config_reset()
    scene = grt.Scene(path); scene.wait_until_loaded()
    blob_triangles = max(scene.mesh_data_array(m, "triangles", np.float32).size // 24 for m in range(scene.mesh_data_count))
    scene.close()
    cases = [({}, 3 + 40),                                                                        # 39 extra copies of the blob: 39 x 768 x 176 B = 5 MB, under the default 64 MB
             (dict(static_mesh_copy_limit_mb=1), 3),                                              # ... over 1 MB: the blob stays instanced
             (dict(static_copy_budget_mb=1), 3 + int((1048576 - 3 * 2 * 176) // (blob_triangles * 176)))]   # the total budget: the rectangles and as many blobs as fit
    for config, members in cases:
        scene, pt = staged(grt, path, 32, 24, 1, **config)
        assert pt.static_geometry_members == members, (config, pt.static_geometry_members, members)
        copies = int((pt.array("alias_mesh_ids") >= 0).sum())
        assert pt.static_geometry_bytes >= copies * (96 + 48 + 8) and pt.static_geometry_bytes < copies * 176 * 2 + 4096
        assert (pt.array("tlas_indices") >= 0).sum() == scene.mesh_count                          # every scene instance still has exactly one row
        pt.close(); scene.close()
    grt.config_reset()


# ---- the builder of the flattened tree on its own (host/StaticBVHBuilder.cpp) ------------------------------------------------

def build_static(grt, triangles, threads=0):
    """(n, 3, 3) vertices -> BVH2 nodes (box min, box max, left_or_first, count | axis << 30), leaf -> triangle indices"""
    import ctypes
    lib = grt.host_lib()
    t24 = np.zeros((len(triangles), 24), np.float32); t24[:, :9] = np.asarray(triangles, np.float32).reshape(-1, 9)
    handle = lib.grt_build_static_bvh(t24.ctypes.data, len(triangles), threads)
    assert handle, lib.grt_last_error()

    def array(name, dtype):
        size = ctypes.c_size_t(0); p = lib.grt_built_array(handle, name.encode(), ctypes.byref(size))
        return np.frombuffer((ctypes.c_char * size.value).from_address(p), dtype=dtype).copy() if size.value else np.zeros(0, dtype)
    nodes = array("bvh2_nodes", np.uint8).reshape(-1, 32); indices = array("bvh2_indices", np.int32); wide = array("bvh8_nodes", np.uint8)
    lib.grt_built_free(handle)
    return nodes, indices, wide


def check_static_tree(nodes, indices, triangles, rng):
    """What traversal relies on: one reference per leaf, children inside their parent, and -- the point of a spatial split --
    every point of a triangle lies in the box of at least one of the leaves that reference it."""
    boxes = nodes[:, :24].view(np.float32).reshape(-1, 2, 3); link = nodes[:, 24:28].view(np.int32).ravel(); word = nodes[:, 28:32].view(np.uint32).ravel()
    count = word & 0x3fffffff
    leaves_of = {}
    todo, seen, leaf_order = [0], 0, []
    while todo:
        k = todo.pop(); seen += 1
        if count[k]:
            assert count[k] == 1                                                   # BVH8Converter wants one primitive per binary leaf
            leaf_order.append(int(link[k])); leaves_of.setdefault(int(indices[link[k]]), []).append(k)
            continue
        for child in (link[k], link[k] + 1):
            assert (boxes[child, 0] >= boxes[k, 0] - 1e-4).all() and (boxes[child, 1] <= boxes[k, 1] + 1e-4).all(), (k, child)
            todo.append(int(child))
    assert sorted(leaf_order) == list(range(len(indices))) and seen == len(nodes) - 1     # every node reached once (node 1 is the alignment dummy)
    assert sorted(leaves_of) == list(range(len(triangles)))                              # every triangle referenced
    for t in rng.choice(len(triangles), min(len(triangles), 400), replace=False):
        w = rng.dirichlet((1, 1, 1), 60).astype(np.float32)
        points = w @ np.asarray(triangles[t], np.float32)
        covered = np.zeros(len(points), bool)
        for k in leaves_of[int(t)]:
            eps = 1e-4 * (1.0 + np.abs(boxes[k]).max())
            covered |= ((points >= boxes[k, 0] - eps) & (points <= boxes[k, 1] + eps)).all(axis=1)
        assert covered.all(), (int(t), len(leaves_of[int(t)]))
    return max(len(v) for v in leaves_of.values())


def test_static_bvh_builder_on_soups_slivers_and_degenerate_input(grt):
    rng = np.random.default_rng(3)
    soup = rng.uniform(-1, 1, (3000, 1, 3)) + rng.normal(size=(3000, 3, 3)) * 0.05
    # long thin triangles crossing a cloud of small ones: what spatial splits are for
    slivers = np.concatenate([soup[:1500], rng.uniform(-1, 1, (60, 1, 3)) * [1, 0.02, 1] + rng.normal(size=(60, 3, 3)) * [2.0, 0.01, 0.01]])
    cases = {
        "soup": soup, "slivers": slivers,
        "one": soup[:1], "two": soup[:2], "three": soup[:3],
        "copies": np.repeat(soup[:1], 64, axis=0),                                   # identical references: no plane separates them
        "points": np.repeat(rng.uniform(-1, 1, (50, 1, 3)), 3, axis=1),              # zero-area triangles
        "flat": np.concatenate([rng.uniform(-1, 1, (500, 3, 2)), np.zeros((500, 3, 1))], axis=2),   # all in the plane z = 0
        "scales": np.concatenate([rng.uniform(-1, 1, (800, 1, 3)) + rng.normal(size=(800, 3, 3)) * 1e-3, rng.normal(size=(4, 3, 3)) * 30.0]),
    }
    for name, triangles in cases.items():
        nodes, indices, wide = build_static(grt, triangles)
        assert wide.size % 80 == 0 and wide.size > 0
        most = check_static_tree(nodes, indices, triangles, rng)
        assert len(indices) >= len(triangles) and len(indices) <= 3 * len(triangles) + 8, (name, len(indices))   # splits duplicate references, within reason
        if name == "slivers":
            assert most > 1                                                       # the long ones were cut
        if name in ("one", "two", "three", "copies", "points"):
            assert len(indices) == len(triangles)                                 # nothing to gain from cutting these
    nodes, indices, wide = build_static(grt, np.zeros((0, 3, 3)))
    assert len(indices) == 0 and wide.size == 0


def test_static_bvh_builder_does_not_depend_on_the_thread_count(grt):
    """The pieces handed to the threads are subtrees of ONE tree that every thread count splits the same way."""
    rng = np.random.default_rng(4)
    triangles = rng.uniform(-1, 1, (20000, 1, 3)) + rng.normal(size=(20000, 3, 3)) * 0.03
    reference = build_static(grt, triangles, threads=1)
    for threads in (2, 3, 8):
        again = build_static(grt, triangles, threads=threads)
        assert all(np.array_equal(a, b) for a, b in zip(reference, again)), threads


def test_svgf_frames_with_a_moving_camera_do_not_depend_on_the_flattening(grt, oracle):
    """The g-buffer's instance ids are table rows, which differ between the layouts, but every row names the same scene
    instance in every frame: reprojection, its consistency tests and the history lengths come out the same, and with one
    emitter so do the filtered frames, float for float."""
    runs = []
    for merge in (0, 1):
        scene, pt = staged(grt, grt.scene_path("cornellbox"), 64, 48, merge, num_bounces=3, enable_svgf=1, enable_taa=1)
        view = oracle.SceneView(pt); frame = oracle.Frame(view)
        frames, histories = [], []
        for f in range(4):
            if f:
                scene.set_camera((0.02 * f, 1.0 + 0.01 * f, 6.8), (0.0, 0.004 * f, 0.0, 1.0)); pt.update()
                view.scene.camera = oracle.SceneView(pt).scene.camera
            vp = pt.view_projection()
            for i in range(16):
                view.scene.view_projection[i] = vp[0][i]; view.scene.view_projection_prev[i] = vp[1][i]
            frame.render_sample(pt.sample_index)
            frames.append(frame.final[:, :64, :3].copy()); histories.append(frame.buffers["hl"].reshape(48, -1)[:, :64].copy())
        runs.append((frames, histories))
        pt.close(); scene.close()
    for f in range(4):
        assert np.array_equal(runs[0][1][f], runs[1][1][f]), f
        assert np.array_equal(runs[0][0][f], runs[1][0][f]), f
    assert runs[0][1][3].mean() > 2.0 and (runs[0][1][3] == 0).any()          # most pixels reproject over the four frames, some are disoccluded
    grt.config_reset()


def test_static_bvh_builder_survives_non_finite_vertices(grt):
    """A broken mesh must not hang the build (the splits fall back to halving lists whose boxes cannot be compared)."""
    rng = np.random.default_rng(1)
    triangles = rng.uniform(-1, 1, (2000, 1, 3)) + rng.normal(size=(2000, 3, 3)) * 0.05
    for poison in (np.nan, np.inf, 3e38):
        broken = triangles.copy(); broken[5, 1, 2] = poison; broken[100, 0, 0] = poison
        nodes, indices, wide = build_static(grt, broken)
        assert len(indices) >= 2000 and sorted(set(indices.tolist())) == list(range(2000)) and wide.size > 0
    # ... nor the 8-wide collapse behind it: boxes of non-finite vertices make every candidate of its cost table NaN or +inf; an inconsistent table once sent
    # gather_children past a node's eight slots (a stack overflow found with one +inf vertex among 20 000 triangles; BVH.cpp: fill_cost_table halves the budget then)
    big = rng.uniform(-1, 1, (20000, 1, 3)) + rng.normal(size=(20000, 3, 3)) * 0.03
    for poison in (np.inf, np.nan):
        broken = big.copy(); broken[5, 0, 0] = poison
        nodes, indices, wide = build_static(grt, broken)
        assert sorted(set(indices.tolist())) == list(range(20000)) and wide.size > 0 and wide.size % 80 == 0


def test_early_split_clipping_covers_every_triangle_with_the_boxes_of_its_pieces(grt):
    """StaticBVHBuilder::presplit (in front of the device's Morton-order BLAS build, which has no spatial splits): a triangle longer than the limit
    is cut into pieces, each a reference with the piece's box. What the tree needs of those boxes: every point of a triangle lies in the box of
    one of ITS pieces (a ray that hits the triangle there walks into a leaf that holds it), no piece's box sticks out of the triangle's own box
    by more than the ulp it is widened by, pieces are no longer than the limit, and a triangle within the limit stays one reference."""
    import ctypes
    lib = grt.host_lib()
    lib.grt_static_presplit.restype = ctypes.c_int
    lib.grt_static_presplit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(7)
    n = 400
    tris = np.zeros((n, 24), np.float32)
    centre = rng.uniform(-10, 10, (n, 1, 3))
    size = np.where(rng.random((n, 1, 1)) < 0.15, rng.uniform(5, 40, (n, 1, 1)), rng.uniform(0.01, 0.5, (n, 1, 1)))   # a few huge ones among the small
    corners = (centre + rng.normal(size=(n, 3, 3)) * size).astype(np.float32)
    corners[::17, :, 1] = corners[::17, :1, 1]                                   # axis-aligned (flat) ones, like a floor
    tris[:, 0:9] = corners.reshape(n, 9)
    tris[:, 9:18] = np.tile(np.array([0, 1, 0], np.float32), (n, 3))             # (normals / uvs ride along untouched)
    limit = 2.0
    source = np.zeros(64 * n, np.int32); boxes = np.zeros((64 * n, 6), np.float32)
    count = lib.grt_static_presplit(tris.ctypes.data, n, ctypes.c_float(limit), source.ctypes.data, boxes.ctypes.data, len(source))
    assert n < count <= len(source)
    source, boxes = source[:count], boxes[:count]
    assert (np.diff(source) >= 0).all() and set(source.tolist()) == set(range(n))   # pieces of a triangle are consecutive, every triangle has some
    lo_t, hi_t = corners.min(axis=1), corners.max(axis=1)
    pieces = np.bincount(source, minlength=n)
    extent_t = (hi_t - lo_t).max(axis=1)
    assert (pieces[extent_t <= limit] == 1).all() and (pieces[extent_t > 2 * limit] >= 2).all() and pieces.max() <= 64
    pad = 0.01 + 1e-5 * np.abs(np.concatenate([lo_t, hi_t], axis=1)).max(axis=1)    # a flat box is padded by 0.001 (AABB::fix_if_needed), a cut one widened by an ulp
    assert (boxes[:, :3] >= lo_t[source] - pad[source, None]).all() and (boxes[:, 3:] <= hi_t[source] + pad[source, None]).all()
    assert (boxes[:, 3:] > boxes[:, :3]).all()                                       # no box of zero thickness (the node test is strict)
    uncapped = pieces[source] < 60
    assert ((boxes[:, 3:] - boxes[:, :3]).max(axis=1)[uncapped] <= limit + 0.01).all()
    # coverage: random points of every triangle
    for t in np.flatnonzero(pieces > 1)[:60]:
        u = rng.random((300, 2)); flip = u.sum(axis=1) > 1; u[flip] = 1 - u[flip]
        pts = corners[t, 0] + u[:, :1] * (corners[t, 1] - corners[t, 0]) + u[:, 1:] * (corners[t, 2] - corners[t, 0])
        mine = boxes[source == t]
        tol = 1e-5 * np.abs(corners[t]).max()                                    # the points themselves are rounded
        inside = ((pts[:, None, :] >= mine[None, :, :3] - tol) & (pts[:, None, :] <= mine[None, :, 3:] + tol)).all(axis=2).any(axis=1)
        assert inside.all(), (int(t), int((~inside).sum()))


def test_learned_slot_order_reseats_children_and_nothing_else(grt, oracle):
    """host/SlotOrder.cpp (config static_slot_learning_rays, on by default): the flattened tree's children are dealt to the octant slots by what seeded sample
    rays say. Only the ORDER of a walk may change: same node count, same multiset of child boxes / leaf contents per tree, identical closest hits and occlusion
    answers on every compared ray (exact ties in t aside: none among these rays), fewer node steps for the camera's rays, and a tree that is a pure function
    of scene, configuration and camera (built twice: byte-identical)."""
    w, h = 192, 108
    def build(**config):
        scene, pt = staged(grt, grt.scene_path("sponza"), w, h, 1, **config)
        view = oracle.SceneView(pt)
        return scene, pt, view
    scene_a, pt_a, plain = build(static_slot_learning_rays=0)
    o, d = rays_for(plain, w, h, 14.0, 20000, 11)
    hits_a, stats_a = plain.trace(o, d)
    far = np.full(o.shape[1], 7.0, np.float32)
    occluded_a = plain.trace_shadow(o, d, far)[0]
    nodes_a = pt_a.array("bvh8_nodes").reshape(-1, 80).copy(); first_a = 2 * scene_a.mesh_count
    aliases_a = (pt_a.array("alias_mesh_ids").copy(), pt_a.array("alias_triangle_ids").copy()); triangles_a = pt_a.array("triangles").copy()
    pt_a.close(); scene_a.close()

    scene_b, pt_b, learned = build(static_slot_learning_rays=300000)
    hits_b, stats_b = learned.trace(o, d)
    occluded_b = learned.trace_shadow(o, d, far)[0]
    nodes_b = pt_b.array("bvh8_nodes").reshape(-1, 80).copy()
    assert nodes_a.shape == nodes_b.shape and not np.array_equal(nodes_a, nodes_b)
    assert np.array_equal(triangles_a.view(np.uint32), pt_b.array("triangles").view(np.uint32))      # triangles, their order and what the copies stand for: untouched
    assert np.array_equal(aliases_a[0], pt_b.array("alias_mesh_ids")) and np.array_equal(aliases_a[1], pt_b.array("alias_triangle_ids"))
    # per node: the children are the same children in other seats -- compare each node's sorted (meta kind / count, six box bytes) rows, nodes sorted too
    def child_rows(nodes):
        rows = []
        for n in nodes[first_a:]:
            meta, imask = n[24:32], n[15]
            kids = []
            for s in range(8):
                if meta[s] == 0: continue
                kind = 255 if (imask >> s) & 1 else int(meta[s] >> 5)                                # inner, or the leaf's unary triangle count
                kids.append((kind,) + tuple(int(n[32 + 8 * k + s]) for k in range(6)))
            rows.append((tuple(int(v) for v in n[:15]), tuple(sorted(kids))))
        return sorted(rows)
    assert child_rows(nodes_a) == child_rows(nodes_b)
    ta, tb = unpack_hits(hits_a), unpack_hits(hits_b)
    assert np.array_equal(hits_a[:, 2], hits_b[:, 2])                                                 # t, to the bit
    same = (hits_a[:, :2] == hits_b[:, :2]).all(axis=1)
    assert same.mean() > 0.999, same.mean()                                                           # (instance, triangle): exact ties between coplanar triangles may resolve differently
    assert np.array_equal(occluded_a, occluded_b)
    camera = slice(0, w * h)
    steps_a, steps_b = plain.trace(o[:, camera], d[:, camera])[1].nodes, learned.trace(o[:, camera], d[:, camera])[1].nodes
    assert steps_b < 0.97 * steps_a, (steps_a, steps_b)                                               # the camera's rays: measurably shorter walks (-10 % at 1080p on the device)
    pt_b.close(); scene_b.close()

    scene_c, pt_c, _ = build(static_slot_learning_rays=300000)
    assert np.array_equal(nodes_b, pt_c.array("bvh8_nodes").reshape(-1, 80))
    pt_c.close(); scene_c.close()


def _wide_child_rows(wide):
    """Per 80-byte node: (header without imask, its children as sorted (kind, six box bytes) rows); the list sorted -- what a re-seating may not change."""
    rows = []
    for n in wide.reshape(-1, 80):
        meta, imask = n[24:32], n[15]
        kids = []
        for s in range(8):
            if meta[s] == 0: continue
            kind = 255 if (imask >> s) & 1 else int(meta[s] >> 5)
            kids.append((kind,) + tuple(int(n[32 + 8 * k + s]) for k in range(6)))
        rows.append((tuple(int(v) for v in n[:15]), tuple(sorted(kids))))
    return sorted(rows)


def test_slot_learner_on_soups_slivers_and_degenerate_input_and_any_thread_count(grt):
    """bvh8_learn_slot_order (host/SlotOrder.cpp) on what a loader can hand it: it must not crash, must only re-seat children (the same children per node, the same
    node count), and must give the same bytes on 1, 3 and 8 threads (seeded samples, integer scores)."""
    import ctypes
    lib = grt.host_lib()
    rng = np.random.default_rng(8)
    soup = rng.uniform(-1, 1, (4000, 1, 3)) + rng.normal(size=(4000, 3, 3)) * 0.05
    broken = soup[:300].copy(); broken[7, 1, 2] = np.nan; broken[11, 0, 0] = np.inf
    one_bad_vertex = soup.copy(); one_bad_vertex[5, 0, 0] = np.inf   # (every sample ray that touches it is dropped: a ray of NaNs would walk the whole tree)
    cases = {
        "soup": soup, "one": soup[:1], "four": soup[:4], "copies": np.repeat(soup[:1], 64, axis=0),
        "points": np.repeat(rng.uniform(-1, 1, (50, 1, 3)), 3, axis=1),
        "flat": np.concatenate([rng.uniform(-1, 1, (500, 3, 2)), np.zeros((500, 3, 1))], axis=2),
        "non-finite": broken, "one infinite vertex": one_bad_vertex,
    }
    def learned(triangles, rays, threads):
        t24 = np.zeros((len(triangles), 24), np.float32); t24[:, :9] = np.asarray(triangles, np.float32).reshape(-1, 9)
        handle = lib.grt_build_static_bvh(t24.ctypes.data, len(triangles), 0)
        assert handle, lib.grt_last_error()
        def wide():
            size = ctypes.c_size_t(0); p = lib.grt_built_array(handle, b"bvh8_nodes", ctypes.byref(size))
            return np.frombuffer((ctypes.c_char * size.value).from_address(p), dtype=np.uint8).copy() if size.value else np.zeros(0, np.uint8)
        before = wide()
        assert lib.grt_built_learn_slot_order(handle, rays, threads) == 0, lib.grt_last_error()
        after = wide()
        lib.grt_built_free(handle)
        return before, after
    for name, triangles in cases.items():
        before, after = learned(triangles, 20000, 0)
        assert before.shape == after.shape, name
        assert _wide_child_rows(before) == _wide_child_rows(after), name
    before, one = learned(soup, 60000, 1)
    assert not np.array_equal(before, one)                      # (a soup of 4 000 triangles does get re-seated)
    for threads in (3, 8):
        assert np.array_equal(one, learned(soup, 60000, threads)[1]), threads
    assert np.array_equal(learned(soup, 0, 0)[0], learned(soup, 0, 0)[1])     # no rays: nothing moves


def test_the_skipping_walk_finds_the_reference_walks_hits_in_fewer_node_steps(grt, oracle):
    """rt_set_skip_behind_hit (config skip_behind_hit, default on, one-tree scenes only): a stack entry carries a 16-bit lower bound of the entry distance of the
    children left in it and is dropped unvisited when that bound is not in front of the hit held. The oracle restates both walks; on the same tree and rays the
    skipping one finds the same closest hits (t to the bit; only which of two coplanar triangles at exactly that t is named can differ), tests the same
    triangles, fetches a tenth fewer nodes; shadow rays do not change at all (their limit never moves)."""
    scene, pt = make_pathtracer(grt, "sponza", 96, 54, -1)
    assert pt.static_geometry_whole_scene and pt.skip_behind_hit
    view = oracle.SceneView(pt)
    o, d = rays_for(view, 96, 54, 14.0, 120000, 5)
    hits_skip, stats_skip = view.trace(o, d)
    md = np.random.default_rng(2).uniform(0.5, 12.0, o.shape[1]).astype(np.float32)
    shadow_skip, shadow_stats_skip = view.trace_shadow(o, d, md)
    grt.config_set(skip_behind_hit=0)
    assert not pt.skip_behind_hit
    walk = oracle.SceneView(pt)                                      # the same arrays, the reference's walk
    hits_walk, stats_walk = walk.trace(o, d)
    shadow_walk, shadow_stats_walk = walk.trace_shadow(o, d, md)
    mesh_a, tri_a, t_a, u_a, v_a = unpack_hits(hits_skip); mesh_b, tri_b, t_b, u_b, v_b = unpack_hits(hits_walk)
    assert (tri_b != -1).mean() > 0.5
    assert np.array_equal(t_a.view(np.uint32), t_b.view(np.uint32))   # every ray: the same distance, to the bit (a miss is a miss)
    tie = tri_a != tri_b
    assert tie.sum() <= 1e-3 * tie.size, int(tie.sum())
    assert np.array_equal(hits_skip[~tie], hits_walk[~tie])
    assert stats_skip.rays == stats_walk.rays == o.shape[1]
    assert stats_skip.nodes < 0.93 * stats_walk.nodes, (stats_skip.nodes, stats_walk.nodes)
    assert abs(int(stats_skip.triangles) - int(stats_walk.triangles)) <= 1e-4 * stats_walk.triangles   # a dropped visit could enter no child: no triangle test goes with it
    assert np.array_equal(shadow_skip, shadow_walk) and shadow_stats_skip.nodes == shadow_stats_walk.nodes and shadow_stats_skip.triangles == shadow_stats_walk.triangles
    pt.close(); scene.close()
    # a scene that keeps a TLAS walks the reference's way whatever the wish
    scene, pt = make_pathtracer(grt, "sponza", 96, 54, -1, merge_static=0)
    assert not pt.skip_behind_hit
    pt.close(); scene.close(); grt.config_reset()


FAR_VIEWPOINT = ((-129.707321, 17.916590, 43.054050), (0.011467, 0.408287, 0.005129, -0.912762))   # the reference's ninth Sponza point of view (Util/PerfTest.h:30-40): 138 units from where the scene's own camera stands


def _node_steps_of_a_sample(oracle, pt):
    oc = oracle.Frame(oracle.SceneView(pt)).render_sample(0)
    return oc.trace_stats.nodes / oc.trace_stats.rays, oc.trace_stats.triangles / oc.trace_stats.rays


def test_the_flattened_tree_is_seated_again_when_the_camera_has_travelled(grt, oracle):
    """config static_reseat_distance (0.1 of the flattened geometry's diagonal): the seating of the tree's children is trained on paths from the camera as it
    stood (static_slot_learning_viewpoint); when the camera has travelled further than that, a worker seats a copy of the tree for the new viewpoint beside the
    frame loop and the nodes are swapped in between two frames (rt_update_nodes). What the re-seated tree costs the new viewpoint's rays is what a tree BUILT
    for that viewpoint costs them, and less than the stale seating; boxes, leaves and triangles do not change, so hits cannot."""
    import time
    w, h = 320, 180
    scene, pt = make_pathtracer(grt, "sponza", w, h, -1)
    assert pt.static_geometry_whole_scene and pt.reseats_completed == 0
    view = oracle.SceneView(pt); o, d = rays_for(view, w, h, 150.0, 40000, 9)
    hits_before, _ = view.trace(o, d)
    nodes_before = pt.array("bvh8_nodes").copy()
    near = scene.get_camera()
    scene.set_camera((near[0][0] + 20.0, near[0][1], near[0][2]), tuple(near[1])); pt.update()      # 20 units: not far enough
    assert pt.reseats_completed == 0 and not pt.reseat_pending
    scene.set_camera(*FAR_VIEWPOINT); pt.update()
    assert pt.reseat_pending                                                                        # ... beside the frame loop: this update did not wait
    deadline = time.time() + 120
    while pt.reseats_completed == 0:
        assert time.time() < deadline
        time.sleep(0.05); pt.update()
    assert pt.reseats_completed == 1 and not pt.reseat_pending and pt.last_reseat_seconds > 0.0
    nodes_after = pt.array("bvh8_nodes")
    assert nodes_after.size == nodes_before.size and (nodes_after != nodes_before).any()
    root = pt.static_geometry_top_levels[0]
    assert np.array_equal(pt.array("tlas_nodes").view(np.uint8).reshape(-1, 80)[0], nodes_after.view(np.uint8).reshape(-1, 80)[root])   # node 0, where rays start, is the re-seated root
    view = oracle.SceneView(pt)
    hits_after, _ = view.trace(o, d)
    mesh_a, tri_a, t_a, _, _ = unpack_hits(hits_before); mesh_b, tri_b, t_b, _, _ = unpack_hits(hits_after)
    assert np.array_equal(t_a.view(np.uint32), t_b.view(np.uint32)) and (tri_a != tri_b).sum() <= 1e-3 * tri_a.size
    reseated = _node_steps_of_a_sample(oracle, pt)
    pt.update(); assert pt.reseats_completed == 1                                                   # seated for where the camera stands: nothing more to do
    pt.close(); scene.close()
    # a tree built for the far viewpoint from the start
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("sponza")); scene.set_camera(*FAR_VIEWPOINT)
    pt = grt.Pathtracer(scene, w, h, device=-1); pt.update()
    fresh = _node_steps_of_a_sample(oracle, pt)
    pt.close(); scene.close()
    # ... and the seating left as it was (static_reseat_distance 0)
    scene, pt = make_pathtracer(grt, "sponza", w, h, -1, static_reseat_distance=0)
    scene.set_camera(*FAR_VIEWPOINT); pt.update(); pt.update()
    assert pt.reseats_completed == 0 and not pt.reseat_pending
    stale = _node_steps_of_a_sample(oracle, pt)
    pt.close(); scene.close(); grt.config_reset()
    assert abs(reseated[0] - fresh[0]) <= 0.01 * fresh[0], (reseated, fresh)
    assert reseated[0] < 0.98 * stale[0] and reseated[1] < stale[1], (reseated, stale)
