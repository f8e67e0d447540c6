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
        assert pt.static_geometry_members == (382 if merge else 0)
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
        results[merge] = (hits.copy(), bounce.copy(), occluded.copy(), pt.array("tlas_indices").copy(), int((pt.array("alias_mesh_ids") < 0).sum()) if merge else pt.array("triangles").size // 24)
        pt.close(); scene.close()
    for which in (0, 1):
        a, b = results[1][which], results[0][which]
        mesh_a, tri_a, t_a, u_a, v_a = unpack_hits(a); mesh_b, tri_b, t_b, u_b, v_b = unpack_hits(b)
        hit = tri_b != -1
        assert hit.mean() > 0.5 and np.array_equal(hit, tri_a != -1)
        assert np.array_equal(t_a.view(np.uint32), t_b.view(np.uint32))                    # the same distance, to the bit, for every one of ~2.5 M rays
        tie = hit & (tri_a != tri_b)                                                       # two triangles at exactly the closest distance: the walk's order decides
        assert tie.sum() <= 1e-3 * hit.sum(), int(tie.sum())
        same = hit & ~tie
        assert np.array_equal(u_a[same], u_b[same]) and np.array_equal(v_a[same], v_b[same])
        assert np.array_equal(results[1][3][mesh_a[same]], results[0][3][mesh_b[same]])    # the same scene instance
        assert (tri_a[hit] < results[1][4]).all()                                          # never a copy
    assert np.array_equal(results[1][2], results[0][2])                                    # any-hit: the same rays are occluded
    grt.config_reset()


def test_flattened_scene_with_moving_instances_renders_like_the_oracle(grt, oracle, tmp_path):
    """Static floor and emitters flattened, 40 transformed instances beside them in the TLAS: frames and queue sizes against
    the oracle, then one of the static instances starts to move -- the flattening dissolves and the frames still agree."""
    from test_tlas import instanced_scene_file
    from test_gpu_parity import compare_frames
    grt.config_reset(); grt.config_set(num_bounces=4)
    scene = grt.Scene(instanced_scene_file(str(tmp_path / "s"), count=40)); grt.config_set(num_bounces=4)
    pt = grt.Pathtracer(scene, 192, 128, device=0); pt.update()
    assert pt.static_geometry_members == 3 and pt.array("tlas_indices").size == scene.mesh_count + 1
    compare_frames(grt, oracle, pt, 2, 192, 128)
    scene.set_mesh_transform(0, (0.0, -0.5, 0.0), (0.0, 0.0, 0.0, 1.0), 1.0)
    pt.invalidate("scene"); pt.update()
    assert pt.static_geometry_members == 0 and pt.array("tlas_indices").size == scene.mesh_count
    compare_frames(grt, oracle, pt, 2, 192, 128)
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
