"""Flattened static geometry on the MI355X (config merge_static, the default; tests/test_static_geometry.py has the CPU side):
the HIP traversal kernels over the flattened layout against the oracle walking the same arrays -- bit for bit, as every other
trace test -- and against the same kernels over the reference's layout (one BLAS per mesh), at the benchmark's size."""
import numpy as np
import pytest

from conftest import make_pathtracer, unpack_hits
from test_gpu_parity import secondary_rays
from test_static_geometry import rays_for

pytestmark = pytest.mark.gpu


def test_flattened_sponza_on_the_device_finds_what_the_reference_layout_finds(grt, oracle):
    w, h = 1920, 1080
    results = {}
    for merge in (1, 0):
        scene, pt = make_pathtracer(grt, "sponza", w, h, 0, merge_static=merge)
        assert pt.static_geometry_members == (384 if merge else 0)
        view = oracle.SceneView(pt)
        if merge:
            o, d = rays_for(view, w, h, 14.0, 400000, 21)
            hits, _ = grt.trace_rays(pt.ctx, o, d)
            so, sd = secondary_rays(view, o, d, hits, 3)                   # bounce rays from the hit points: the incoherent case
            md = np.full(so.shape[1], 6.0, np.float32)
            rays = (o, d, so, sd, md)
        o, d, so, sd, md = rays
        hits, _ = grt.trace_rays(pt.ctx, o, d)
        bounce, _ = grt.trace_rays(pt.ctx, so, sd)
        occluded, _ = grt.trace_shadow_rays(pt.ctx, so, sd, md)
        # against the oracle on the same arrays (a slice: the oracle is one to two orders slower)
        pick = np.random.default_rng(1).choice(so.shape[1], 150000, replace=False)
        assert np.array_equal(bounce[pick], view.trace(so[:, pick], sd[:, pick])[0])
        assert np.array_equal(hits[:200000], view.trace(o[:, :200000], d[:, :200000])[0])
        assert np.array_equal(occluded[pick].astype(bool), view.trace_shadow(so[:, pick], sd[:, pick], md[pick])[0].astype(bool))
        results[merge] = (hits.copy(), bounce.copy(), occluded.copy(), pt.array("tlas_indices").copy(), int((pt.array("alias_mesh_ids") < 0).sum()) if merge else pt.array("triangles").size // 24, pt.array("mesh_bvh_root_indices").copy())
        pt.close(); scene.close()
    roots_reference_layout = results[0][5]
    for which in (0, 1):
        a, b = results[1][which], results[0][which]
        mesh_a, tri_a, t_a, u_a, v_a = unpack_hits(a); mesh_b, tri_b, t_b, u_b, v_b = unpack_hits(b)
        hit = tri_b != -1
        # 382 of the 384 instances stand with the identity transform: their copies are their triangles bit for bit, the hit
        # is the same to the bit. The other two are copied in world space (the reference's layout takes the ray to object
        # space instead): the same hit up to rounding.
        exact = hit & (roots_reference_layout[mesh_b] < 0)
        both = hit & (tri_a != -1)
        far_apart = np.zeros(hit.shape, bool); far_apart[both] = np.abs(t_a[both] - t_b[both]) > 1e-5 * np.abs(t_b[both])
        grazing = (hit != (tri_a != -1)) | (far_apart & ~exact)
        assert hit.mean() > 0.5 and grazing.sum() <= 1e-4 * hit.sum(), int(grazing.sum())
        assert exact.sum() > 0.95 * hit.sum() and np.array_equal(t_a.view(np.uint32)[exact], t_b.view(np.uint32)[exact])   # the same distance, to the bit, for ~2.4 M rays
        hit = hit & ~grazing
        tie = hit & (tri_a != tri_b)                                                       # two triangles at exactly the closest distance: the walk's order decides
        assert tie.sum() <= 1e-3 * hit.sum(), int(tie.sum())
        same = hit & ~tie
        assert np.array_equal(u_a[same & exact], u_b[same & exact]) and np.array_equal(v_a[same & exact], v_b[same & exact])
        assert np.array_equal(results[1][3][mesh_a[same]], results[0][3][mesh_b[same]])    # the same scene instance
        assert (tri_a[hit] < results[1][4]).all()                                          # never a copy
    assert (results[1][2] != results[0][2]).sum() <= 1e-5 * results[0][2].size             # any-hit: the same rays are occluded (but for a ray grazing one of the two transformed instances)
    grt.config_reset()


def test_seating_and_the_skipping_walk_change_the_walk_and_not_the_hits(grt, oracle):
    """The same rays at the benchmark's size through three builds of the flattened Sponza tree on the device: (a) what ships -- children seated by the
    slot learner, closest-hit rays skipping stack entries behind their hit (rt_set_skip_behind_hit); (b) the same tree walked the reference's way;
    (c) the tree as the collapse leaves it (no learned seating), the reference's walk. All three name the same distance for every ray, to the bit;
    the few rays that name another triangle sit on two coplanar triangles at exactly that distance. Then the counting launches: (a) fetches fewer
    nodes than (b), both exactly what the oracle counts for its restatement of that walk."""
    w, h = 1920, 1080
    runs = {}
    rays = None
    for name, config in (("ships", {}), ("reference_walk", dict(skip_behind_hit=0)), ("unseated", dict(skip_behind_hit=0, static_slot_learning_rays=0))):
        scene, pt = make_pathtracer(grt, "sponza", w, h, 0, **config)
        assert pt.static_geometry_whole_scene and pt.skip_behind_hit == (name == "ships")
        lib = grt.device_lib(); lib.rt_get_skip_behind_hit.argtypes = [__import__("ctypes").c_void_p]
        assert lib.rt_get_skip_behind_hit(pt.ctx) == (1 if name == "ships" else 0)
        view = oracle.SceneView(pt)
        if rays is None:
            o, d = rays_for(view, w, h, 14.0, 400000, 33)
            first, _ = grt.trace_rays(pt.ctx, o, d)
            so, sd = secondary_rays(view, o, d, first, 3)
            rays = (o, d, so, sd)
        o, d, so, sd = rays
        hits, _ = grt.trace_rays(pt.ctx, o, d)
        bounce, _ = grt.trace_rays(pt.ctx, so, sd)
        few, _ = grt.trace_rays(pt.ctx, so[:, :6000], sd[:, :6000])             # a launch small enough for the 8-lanes-per-ray engine
        assert np.array_equal(few, bounce[:6000])
        pick = np.random.default_rng(3).choice(so.shape[1], 100000, replace=False)
        want, oracle_stats = view.trace(so[:, pick], sd[:, pick])
        assert np.array_equal(bounce[pick], want)                              # the device against the oracle's restatement of the SAME walk: bit for bit
        runs[name] = (hits, bounce, oracle_stats.nodes / oracle_stats.rays)
        if name != "unseated":                                                 # the counting launches of one sample, against the oracle's counters
            frame = oracle.Frame(view)
            grt.set_trace_statistics(pt.ctx, True); pt.render(); stats = grt.get_trace_statistics(pt.ctx); grt.set_trace_statistics(pt.ctx, False)
            oc = frame.render_sample(pt.sample_index)
            assert abs(stats["closest"]["rays"] - oc.trace_stats.rays) <= 1e-5 * oc.trace_stats.rays   # (a handful of paths per million end elsewhere: sinf / logf differ by ulps between the device library and glibc)
            for key, ref in (("nodes", oc.trace_stats.nodes), ("triangles", oc.trace_stats.triangles)):
                assert abs(stats["closest"][key] - ref) <= 1e-3 * ref, (name, key)
            assert abs(stats["shadow"]["nodes"] - oc.shadow_stats.nodes) <= 2e-3 * oc.shadow_stats.nodes
            runs[name] += (stats["closest"]["nodes"] / stats["closest"]["rays"],)
        pt.close(); scene.close()
    for other in ("reference_walk", "unseated"):
        for which in (0, 1):
            a, b = runs["ships"][which], runs[other][which]
            mesh_a, tri_a, t_a, u_a, v_a = unpack_hits(a); mesh_b, tri_b, t_b, u_b, v_b = unpack_hits(b)
            assert np.array_equal(t_a.view(np.uint32), t_b.view(np.uint32)), other   # ~2.5 M + ~1.5 M rays: the same distance, to the bit
            tie = tri_a != tri_b
            assert tie.sum() <= 1e-3 * tie.size, (other, int(tie.sum()))
            assert np.array_equal(a[~tie], b[~tie])
    assert runs["ships"][2] < 0.92 * runs["reference_walk"][2] < 0.98 * runs["unseated"][2], [runs[k][2] for k in runs]   # incoherent rays, node steps per ray
    assert runs["ships"][3] < 0.92 * runs["reference_walk"][3]                                                            # the frame's own rays
    grt.config_reset()


def test_flattened_scene_with_moving_instances_renders_like_the_oracle(grt, oracle, tmp_path):
    """A floor and two emitters flattened (one TLAS leaf), 40 rotated / scaled instances of one mesh beside them in the TLAS
    (an instanced mesh is not copied per instance): frames and queue sizes against the oracle. Then the floor starts to
    move: the tree is rebuilt without it and the frames still agree; then an emitter: nothing is left to flatten."""
    from test_tlas import instanced_scene_file
    from test_gpu_parity import compare_frames
    grt.config_reset(); grt.config_set(num_bounces=4, static_mesh_copy_limit_mb=1)   # (39 extra copies of the blob: 5 MB -- over this limit, so it stays instanced)
    scene = grt.Scene(instanced_scene_file(str(tmp_path / "s"), count=40)); grt.config_set(num_bounces=4, static_mesh_copy_limit_mb=1)
    pt = grt.Pathtracer(scene, 192, 128, device=0); pt.update()
    pt.set_flatten_asynchronously(False)    # (the rebuild inside update(): deterministic member counts; the background rebuild has its own test below)
    assert pt.static_geometry_members == 3 and pt.array("tlas_indices").size == 44 and not pt.static_geometry_whole_scene
    assert sorted(pt.array("tlas_indices")[41:].tolist()) == [0, 1, 2] and (pt.array("tlas_indices")[:41] == -1).sum() == 1
    compare_frames(grt, oracle, pt, 2, 192, 128)
    scene.set_mesh_transform(0, (0.5, -0.25, 0.0), (0.0, 0.0, 0.0, 1.0), 1.0)
    for mesh in (5, 20):                                                      # (instances that had leaves of their own anyway)
        position, rotation, scale = scene.mesh_transform(mesh)
        scene.set_mesh_transform(mesh, (position[0], position[1] + 1.0, position[2]), rotation, scale * 1.1)
    pt.invalidate("scene"); pt.update()
    assert pt.static_geometry_members == 2 and pt.array("tlas_indices").size == 44 and sorted(pt.array("tlas_indices")[42:].tolist()) == [1, 2]
    compare_frames(grt, oracle, pt, 2, 192, 128)
    scene.set_mesh_transform(1, (0.0, -0.5, 0.0), (0.0, 0.0, 0.0, 1.0), 1.0)
    pt.invalidate("scene"); pt.update()
    assert pt.static_geometry_members == 0 and sorted(pt.array("tlas_indices").tolist()) == list(range(43))
    compare_frames(grt, oracle, pt, 2, 192, 128)
    pt.close(); scene.close(); grt.config_reset()


def test_world_space_copies_of_transformed_instances_render_like_the_oracle(grt, oracle, tmp_path):
    """The scene with everything: five instances, two of them a scaled and a rotated instance of one file mesh -- all
    flattened, the transformed ones as world-space copies, no TLAS. Light sampling, MIS, textures and media behind it see the
    scene's own instances (the emitter that is a copy is found by index in the light tables)."""
    from scenes import write_scene_with_everything
    from test_loaders import _png_bytes
    from test_gpu_parity import compare_frames
    grt.config_reset()
    scene = grt.Scene(write_scene_with_everything(tmp_path, _png_bytes)); scene.set_sky_scale(0.3)
    grt.config_set(num_bounces=6)
    pt = grt.Pathtracer(scene, 160, 120, device=0); pt.update()
    assert pt.static_geometry_members == 5 and pt.static_geometry_whole_scene
    compare_frames(grt, oracle, pt, 3, 160, 120, luts=grt.read_luts(pt.ctx))   # (rough dielectric + conductor: the device's Kulla-Conty tables)
    pt.close(); scene.close(); grt.config_reset()


def test_pixel_query_names_the_scene_instance_behind_a_flattened_hit(grt):
    answers = []
    for merge in (1, 0):
        scene, pt = make_pathtracer(grt, "cornellbox", 128, 128, 0, merge_static=merge)
        assert pt.static_geometry_members == (8 if merge else 0)
        pt.set_pixel_query(64, 64); pt.render(); pt.update()
        _, mesh, triangle, status = pt.pixel_query
        assert status == 0 and 0 <= mesh < 8
        answers.append((mesh, triangle))
        pt.close(); scene.close()
    assert answers[0] == answers[1]
    grt.config_reset()


def test_alias_and_entry_errors_are_reported(grt):
    import ctypes
    lib = grt.device_lib()
    lib.rt_upload_triangle_aliases.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.rt_set_static_geometry.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    ctx = ctypes.c_void_p()
    assert lib.rt_create(0, ctypes.byref(ctx)) == 0
    assert lib.rt_set_static_geometry(ctx, 0) == 0                                   # nothing to change
    assert lib.rt_set_static_geometry(ctx, 1) != 0 and b"CWBVH" in lib.rt_last_error(ctx)          # no geometry yet
    names = np.full(4, -1, np.int32)
    assert lib.rt_upload_triangle_aliases(ctx, names.ctypes.data, names.ctypes.data) != 0 and b"no geometry" in lib.rt_last_error(ctx)
    lib.rt_destroy(ctx)
    scene, pt = make_pathtracer(grt, "cornellbox", 32, 32, 0, merge_static=0)
    count = pt.array("triangles").size // 24
    rows, triangles = np.full(count, -1, np.int32), np.full(count, -1, np.int32)
    rows[5], triangles[5] = 0, 7; rows[7], triangles[7] = 0, 2                        # copy 5 names triangle 7, itself a copy
    assert lib.rt_upload_triangle_aliases(pt.ctx, rows.ctypes.data, triangles.ctypes.data) != 0 and b"not itself a copy" in lib.rt_last_error(pt.ctx)
    rows[7] = -1; triangles[5] = count                                               # out of range
    assert lib.rt_upload_triangle_aliases(pt.ctx, rows.ctypes.data, triangles.ctypes.data) != 0
    triangles[5] = 7
    assert lib.rt_upload_triangle_aliases(pt.ctx, rows.ctypes.data, triangles.ctypes.data) == 0    # a legal alias: hits on triangle 5 report (row 0, triangle 7)
    assert lib.rt_upload_triangle_aliases(pt.ctx, None, None) == 0                   # and cleared again
    pt.render()
    pt.close(); scene.close(); grt.config_reset()


def test_a_moving_member_does_not_stall_the_frame_loop_for_the_rebuild(grt, oracle):
    """Sponza, all 384 instances flattened; one of them (a vase) starts to move in the middle of a frame loop. Round 3 rebuilt the
    tree of the other 383 inside update(): a frame as long as the host build (0.2-0.8 s). Now that frame and the following ones
    are rendered in the reference's layout while a worker thread builds, and the frame that installs the new tree pays staging +
    upload only. Measured here: the longest frame of both variants (the background one has to be several times shorter), and the
    frames after the switch still agree with the oracle."""
    import ctypes
    import time
    from test_gpu_parity import compare_frames
    from test_gpu_full_size import record
    lib = grt.device_lib(); lib.rt_synchronize.argtypes = [ctypes.c_void_p]
    longest = {}
    for background in (True, False):
        scene, pt = make_pathtracer(grt, "sponza", 640, 360, 0, num_bounces=4)
        pt.set_flatten_asynchronously(background)
        assert pt.static_geometry_whole_scene and pt.static_geometry_members == 384

        def frame():
            started = time.perf_counter()
            pt.update(); pt.render(); assert lib.rt_synchronize(pt.ctx) == 0
            return time.perf_counter() - started
        steady = sorted(frame() for _ in range(30))[15]
        mover = 100
        position, rotation, scale = scene.mesh_transform(mover)
        times = []
        for k in range(1, 400):
            if k <= 5:                                                        # it moves for five frames, then stands still
                scene.set_mesh_transform(mover, (position[0] + 0.05 * k, position[1], position[2]), rotation, scale)
                pt.invalidate("scene")
            times.append(frame())
            if k > 5 and pt.static_geometry_members == 383 and pt.reflatten_in_progress == 0 and len(times) > 20:
                break
            if background and pt.reflatten_in_progress == 1:
                time.sleep(0.002)                                              # (a frame loop with a display would idle here too; the worker needs the cores)
        assert pt.static_geometry_members == 383 and not pt.static_geometry_whole_scene
        if background:
            assert pt.reflattens_completed >= 1
        longest[background] = max(times)
        record("moving member, %s rebuild" % ("background" if background else "in-line"), steady_frame_ms=steady * 1e3, longest_frame_ms=max(times) * 1e3, frames=len(times), build_s=pt.static_geometry_build_seconds)
        pt.invalidate("scene"); pt.update()                                    # (accumulation starts again at sample 0, where compare_frames' oracle starts)
        assert pt.sample_index == 0 and pt.static_geometry_members == 383
        compare_frames(grt, oracle, pt, 2, 640, 360)
        pt.close(); scene.close()
    assert longest[True] < 0.6 * longest[False], longest
    grt.config_reset()
