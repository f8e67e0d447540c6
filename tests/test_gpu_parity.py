"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer / index / traversal results must be bit-exact; floating-point images must
agree within the stated tolerances (transcendental functions differ by a few ulp between glibc
and the device math library, nothing else does)."""
import os

import numpy as np
import pytest

from conftest import make_pathtracer, unpack_hits

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "render_golden.npz")

# image tolerances (relative L1 over the frame, and fraction of pixels off by more than 1 %)
REL_L1_TOL = 1e-4
OUTLIER_FRACTION_TOL = 2e-3


def secondary_rays(view, o, d, hits, seed):
    """Incoherent rays leaving the primary hit points (cosine-ish random directions)."""
    rng = np.random.default_rng(seed)
    _, tri, t, _, _ = unpack_hits(hits)
    ok = tri >= 0
    org = (o + d * np.where(ok, t, 1.0).astype(np.float32) * np.float32(0.999)).astype(np.float32)
    dirs = rng.normal(size=o.shape).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=0)
    return org[:, ok], dirs[:, ok]


@pytest.mark.reference_layout
def test_gpu_frames_equal_the_references_own_kernels(grt, oracle):
    """Closes the loop without the restated oracle in between: the HIP kernels on the MI355X against the reference's
    Pathtracer.cu executed on the CPU (oracle/ref/ref_cuda_harness.cpp, prebuilt into oracle/_ref) on the same
    staged arrays -- queue sizes per bounce and frames."""
    if oracle.ref_lib() is None or not hasattr(oracle.ref_lib(), "ref_cuda_frame_create"):
        pytest.skip("oracle/_ref was built without the reference's device code")
    scene, pt = make_pathtracer(grt, "cornellbox", 96, 64, 0, num_bounces=5)
    view = oracle.SceneView(pt)
    theirs = oracle.ReferenceFrame(view)
    for f in range(3):
        if f:
            pt.update()
        pt.render()
        c = pt.counters()
        rc = theirs.render_sample(pt.sample_index)
        for name in ("trace", "shadow", "diffuse"):
            got, want = list(getattr(c, name)[:5]), [int(v) for v in rc[name][:5]]
            assert got[0] == want[0] and all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(got, want)), (f, name, got, want)
        got, want = pt.read_framebuffer()[:, :96, :3], theirs.final[:, :96, :3]
        assert np.abs(got - want).sum() / want.sum() < REL_L1_TOL, f
    theirs.close(); pt.close(); scene.close()


@pytest.mark.parametrize("scene_name,w,h", [("cornellbox", 256, 256), ("sponza", 640, 360)])
def test_trace_hits_are_bit_exact(grt, oracle, scene_name, w, h):
    scene, pt = make_pathtracer(grt, scene_name, w, h, 0)
    view = oracle.SceneView(pt)
    n = w * h
    o, d, px = grt.generate_rays(pt.ctx, 0, 0, n)
    oo, od, opx = view.generate(0, 0, n)
    assert np.array_equal(px, opx) and np.array_equal(o, oo)
    assert np.allclose(d, od, atol=3e-7, rtol=0)            # Box-Muller jitter: logf / sinf / cosf
    hits_cpu, stats = view.trace(oo, od)
    hits_gpu, _ = grt.trace_rays(pt.ctx, oo, od)
    assert np.array_equal(hits_gpu, hits_cpu)               # mesh id, triangle id, t bits, quantised u,v
    so, sd = secondary_rays(view, oo, od, hits_cpu, 1)
    hits_cpu2, _ = view.trace(so, sd)
    hits_gpu2, _ = grt.trace_rays(pt.ctx, so, sd)
    assert np.array_equal(hits_gpu2, hits_cpu2)
    md = np.full(so.shape[1], 25.0, np.float32)
    occ_cpu, _ = view.trace_shadow(so, sd, md)
    occ_gpu, _ = grt.trace_shadow_rays(pt.ctx, so, sd, md)
    assert np.array_equal(occ_gpu, occ_cpu) and 0 < occ_cpu.mean() < 1
    pt.close(); scene.close()


def test_trace_edge_cases(grt, oracle):
    """Empty batch, a single ray, rays that miss everything, a ray starting on a surface."""
    scene, pt = make_pathtracer(grt, "cornellbox", 64, 64, 0)
    view = oracle.SceneView(pt)
    o = np.array([[0, 0, 0, 0.3], [1, 1, 50, 0.0], [6.8, -20, 0, 0.5]], np.float32)
    d = np.array([[0, 0, 1, 0], [0, 1, 0, 1], [-1, 0, 0, 0]], np.float32)
    hits_cpu, _ = view.trace(o, d)
    hits_gpu, _ = grt.trace_rays(pt.ctx, o, d)
    assert np.array_equal(hits_gpu, hits_cpu)
    assert hits_gpu[1, 1] == 0xffffffff and hits_gpu[2, 1] == 0xffffffff   # misses keep INVALID
    one_gpu, _ = grt.trace_rays(pt.ctx, o[:, :1], d[:, :1])
    assert np.array_equal(one_gpu, hits_cpu[:1])
    none_gpu, _ = grt.trace_rays(pt.ctx, np.zeros((3, 0), np.float32), np.zeros((3, 0), np.float32))
    assert none_gpu.shape == (0, 4)
    pt.close(); scene.close()


@pytest.mark.reference_layout
def test_instanced_scene_trace_is_bit_exact(grt, oracle, tmp_path):
    """TLAS/BLAS with non-identity transforms (rotation, non-unit scale): object-space rays."""
    (tmp_path / "blob.obj").write_text(blob_obj(12))
    shapes = []
    rng = np.random.default_rng(5)
    for i in range(40):
        x, y, z = rng.uniform(-8, 8, 3)
        shapes.append('<shape type="obj"><string name="filename" value="blob.obj"/><transform name="toWorld"><scale value="%f"/>'
                      '<rotate y="1" angle="%f"/><rotate x="1" angle="%f"/><translate x="%f" y="%f" z="%f"/></transform><bsdf type="diffuse"/></shape>'
                      % (rng.uniform(0.5, 2.0), rng.uniform(0, 360), rng.uniform(0, 360), x, y, z))
    (tmp_path / "s.xml").write_text('<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="70"/><transform name="toWorld">'
                                    '<lookat origin="0, 0, 25" target="0, 0, 0" up="0, 1, 0"/></transform></sensor>%s</scene>' % "".join(shapes))
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "s.xml"))
    pt = grt.Pathtracer(scene, 256, 160, device=0); pt.update()
    view = oracle.SceneView(pt)
    o, d, _ = view.generate(0, 0, 256 * 160)
    hits_cpu, stats = view.trace(o, d)
    hits_gpu, _ = grt.trace_rays(pt.ctx, o, d)
    assert stats.instances_transformed > 0
    assert np.array_equal(hits_gpu, hits_cpu)
    assert (hits_cpu[:, 1] != 0xffffffff).mean() > 0.2
    pt.close(); scene.close()


def test_animated_instances_do_not_drain_the_pipeline_or_mix_scene_versions(grt, oracle, tmp_path):
    """Per-frame TLAS rebuild (Integrator::build_tlas with enable_scene_update): every frame the host
    uploads a new TLAS and new instance tables; they go into a ring of versions with asynchronous
    copies, so frames stay in flight. (a) each frame, rendered and read on its own, matches the oracle on
    the scene state of that frame; (b) six frames accumulated with explicit sample indices while the
    instances move give the same image bit for bit with 1 and with 3 frames in flight, i.e. every frame
    traced the TLAS version that was current when it was submitted."""
    import ctypes
    (tmp_path / "blob.obj").write_text(blob_obj(10))
    rng = np.random.default_rng(9)
    shapes = ['<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="-90"/><scale value="14"/><translate y="-3"/></transform><bsdf type="diffuse"/></shape>',
              '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><scale value="4"/><translate y="12"/></transform><emitter type="area"><rgb name="radiance" value="20, 20, 20"/></emitter></shape>']
    for i in range(30):
        x, y, z = rng.uniform(-6, 6, 3)
        shapes.append('<shape type="obj"><string name="filename" value="blob.obj"/><transform name="toWorld"><scale value="%f"/><translate x="%f" y="%f" z="%f"/></transform><bsdf type="diffuse"/></shape>' % (rng.uniform(0.6, 1.4), x, y, z))
    (tmp_path / "s.xml").write_text('<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="60"/><transform name="toWorld">'
                                    '<lookat origin="0, 4, 22" target="0, 0, 0" up="0, 1, 0"/></transform></sensor>%s</scene>' % "".join(shapes))
    w, h = 200, 120
    lib = grt.device_lib()

    def move(scene, frame):
        for m in range(2, scene.mesh_count):
            pos, _, scale = base[m]
            a = 0.35 * frame + 0.2 * m
            scene.set_mesh_transform(m, [pos[0] + 0.4 * np.sin(a), pos[1], pos[2] + 0.4 * np.cos(a)], [0.0, float(np.sin(a / 2)), 0.0, float(np.cos(a / 2))], scale)

    # (a) frame by frame against the oracle
    grt.config_reset(); grt.config_set(num_bounces=3)
    scene = grt.Scene(str(tmp_path / "s.xml"))
    base = [scene.mesh_transform(m) for m in range(scene.mesh_count)]
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    for frame in range(3):
        move(scene, frame); pt.invalidate("scene"); pt.update()
        assert pt.sample_index == 0
        pt.render()
        view = oracle.SceneView(pt); ref = oracle.Frame(view)
        oc = ref.render_sample(0); c = pt.counters()
        assert all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(list(c.trace[:3]), list(oc.trace[:3])))
        got, want = pt.read_framebuffer()[:, :w, :3], ref.final[:, :w, :3]
        assert np.abs(got - want).sum() / want.sum() < REL_L1_TOL, frame
    pt.close(); scene.close()

    # (b) frames in flight: explicit sample indices so that the accumulator depends on every frame
    images = []
    for in_flight in (1, 3):
        grt.config_reset(); grt.config_set(num_bounces=3)
        scene = grt.Scene(str(tmp_path / "s.xml"))
        pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
        grt.set_samples_in_flight(pt.ctx, in_flight)
        for frame in range(40):      # more frames than the scene ring has versions (12): it wraps around under frames in flight
            move(scene, frame); pt.invalidate("scene"); pt.update()
            assert lib.rt_render_sample(pt.ctx, frame) == 0
        images.append(pt.read_framebuffer().copy())
        pt.close(); scene.close()
    assert np.array_equal(images[0], images[1]) and images[0][..., :3].max() > 0.0


def blob_obj(n):
    """Small closed lumpy sphere (n x 2n quads) as OBJ text."""
    lines = []
    for i in range(n + 1):
        th = np.pi * i / n
        for j in range(2 * n):
            ph = np.pi * j / n
            r = 1.0 + 0.15 * np.sin(3 * th) * np.cos(2 * ph)
            lines.append("v %f %f %f" % (r * np.sin(th) * np.cos(ph), r * np.cos(th), r * np.sin(th) * np.sin(ph)))
    for i in range(n):
        for j in range(2 * n):
            a = i * 2 * n + j + 1; b = i * 2 * n + (j + 1) % (2 * n) + 1
            c = (i + 1) * 2 * n + (j + 1) % (2 * n) + 1; e = (i + 1) * 2 * n + j + 1
            lines.append("f %d %d %d %d" % (a, b, c, e))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("scene_name,w,h,bvh_type", [("cornellbox", 256, 256, 2), ("cornellbox", 256, 256, 4),
                                                      ("sponza", 480, 270, 2), ("sponza", 480, 270, 4), ("sponza", 480, 270, 1)])
def test_binary_and_4_wide_bvh_trace_is_bit_exact_and_config_1_renders(grt, oracle, scene_name, w, h, bvh_type):
    """bvh_type = BVH (BVH2.h, BASELINE config #1 on the device), BVH4 (BVH4.h) and SBVH (1: the
    binary kernels over the spatial-split tree): primary and incoherent rays through
    kernel_trace_bvh2 / kernel_trace_bvh4 give the oracle's hits bit for bit (the host permutes the
    triangles by the BVH2 indices, so ids differ from the CWBVH run), shadow rays agree, and a full
    frame matches the oracle's render. Sponza's meshes are file-loaded, so their trees are
    leaf-collapsed (several triangles per leaf) and, for SBVH, reference triangles more than once."""
    scene, pt = make_pathtracer(grt, scene_name, w, h, 0, bvh_type=bvh_type, num_bounces=4)
    if bvh_type == 1:
        assert pt.array("triangles").size // 24 > 262687   # duplicated references
        bvh_type = 2
    view = oracle.SceneView(pt, bvh_type=bvh_type)
    o, d, _ = view.generate(0, 0, w * h)
    hits_cpu, stats = view.trace(o, d)
    hits_gpu, _ = grt.trace_rays(pt.ctx, o, d)
    assert np.array_equal(hits_gpu, hits_cpu) and (hits_cpu[:, 1] != 0xffffffff).mean() > 0.5
    # incoherent rays: from the primary hit points into seeded random directions, and as shadow rays
    rng = np.random.default_rng(11)
    t = hits_cpu[:, 2].view(np.float32)
    ok = hits_cpu[:, 1] != 0xffffffff
    origin = (o + d * np.where(ok, t, 0.0) * 0.999)[:, ok][:, :60000]
    direction = rng.normal(size=origin.shape).astype(np.float32)
    direction /= np.linalg.norm(direction, axis=0)
    h2_cpu, _ = view.trace(origin, direction)
    h2_gpu, _ = grt.trace_rays(pt.ctx, origin, direction)
    assert np.array_equal(h2_gpu, h2_cpu)
    max_dist = rng.uniform(0.05, 5.0, origin.shape[1]).astype(np.float32)
    occ_gpu, _ = grt.trace_shadow_rays(pt.ctx, origin, direction, max_dist)
    assert np.array_equal(occ_gpu.astype(bool), view.trace_shadow(origin, direction, max_dist)[0].astype(bool))
    pt.close(); scene.close()
    if scene_name == "cornellbox":   # BASELINE config #1: 512 x 512, one sample, binary SAH BVH
        scene, pt = make_pathtracer(grt, "cornellbox", 512, 512, 0, bvh_type=bvh_type)
        view = oracle.SceneView(pt, bvh_type=bvh_type); frame = oracle.Frame(view)
        pt.render(); c = pt.counters(); oc = frame.render_sample(pt.sample_index)
        nb = pt.device_config().num_bounces
        assert c.trace[0] == oc.trace[0] == 512 * 512
        assert all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(list(c.trace[:nb]), list(oc.trace[:nb])))
        got, want = pt.read_framebuffer()[:, :512, :3], frame.final[:, :512, :3]
        assert np.abs(got - want).sum() / want.sum() < REL_L1_TOL
        pt.close(); scene.close()


def test_random_samples_are_bit_exact(grt, oracle):
    scene, pt = make_pathtracer(grt, "cornellbox", 300, 200, 0)
    view = oracle.SceneView(pt)
    px = np.arange(0, pt.pitch * 200, 13, dtype=np.uint32)
    for dim, bounce, sample in ((0, 0, 0), (1, 0, 3), (2, 7, 100), (5, 12, 4095), (6, 13, 77), (3, 127, 9), (4, 1, 4096), (5, 2, 100000)):
        got = grt.random_samples(pt.ctx, dim, px, bounce, sample)
        want = view.random(dim, px, bounce, sample)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (dim, bounce, sample)
    pt.close(); scene.close()


def compare_frames(grt, oracle, pt, frames, w, h, luts=None):
    view = oracle.SceneView(pt, luts=luts)
    frame = oracle.Frame(view)
    nb = pt.device_config().num_bounces
    for f in range(frames):
        if f:
            pt.update()
        pt.render()
        c = pt.counters()
        oc = frame.render_sample(pt.sample_index)
        # queue sizes: every path makes the same decisions on both sides, except the handful whose
        # pdf / roulette comparison sits within a few ulp of its threshold (sinf/cosf/logf/... differ
        # by ulps between glibc and the device library): allow 0.2 % + 2 rays per queue.
        for name in ("trace", "shadow", "diffuse", "plastic", "dielectric", "conductor"):
            got_q, want_q = list(getattr(c, name)[:nb]), list(getattr(oc, name)[:nb])
            assert got_q[0] == want_q[0], name
            assert all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(got_q, want_q)), (name, got_q, want_q)
        got, want = pt.read_framebuffer()[:, :w, :3], frame.final[:, :w, :3]
        assert np.isfinite(got).all()
        rel = np.abs(got - want).sum() / want.sum()
        outliers = (np.abs(got - want).max(axis=2) > 0.01 * (want.max(axis=2) + 1e-3)).mean()
        assert rel < REL_L1_TOL and outliers < OUTLIER_FRACTION_TOL, (f, rel, outliers)
    return frame


def test_cornell_render_matches_oracle_and_golden(grt, oracle):
    scene, pt = make_pathtracer(grt, "cornellbox", 48, 48, 0, num_bounces=4)
    frame = compare_frames(grt, oracle, pt, 2, 48, 48)
    g = np.load(GOLDEN)
    got = pt.read_framebuffer()[:, :48, :3]
    assert np.abs(got - g["image"]).sum() / g["image"].sum() < REL_L1_TOL
    pt.close(); scene.close()


def test_cornell_render_box_filter_many_bounces(grt, oracle):
    scene, pt = make_pathtracer(grt, "cornellbox", 160, 120, 0, num_bounces=12, reconstruction_filter=0)
    compare_frames(grt, oracle, pt, 3, 160, 120)
    pt.close(); scene.close()


def test_sponza_render_matches_oracle(grt, oracle):
    scene, pt = make_pathtracer(grt, "sponza", 320, 180, 0, num_bounces=6)
    compare_frames(grt, oracle, pt, 2, 320, 180)
    pt.close(); scene.close()


def test_sponza_plastic_variant_matches_oracle(grt, oracle):
    """Sponza has no plastic; odd material indices become roughplastic alpha=0.3."""
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("sponza"))
    for i in range(1, scene.material_count, 2):
        if scene.material_type(i) == grt.MATERIAL_DIFFUSE:
            scene.set_material(i, grt.MATERIAL_PLASTIC, None, 0.3)
    grt.config_set(num_bounces=5)
    pt = grt.Pathtracer(scene, 320, 180, device=0); pt.update()
    frame = compare_frames(grt, oracle, pt, 2, 320, 180)
    assert sum(pt.counters().plastic[:5]) > 0
    pt.close(); scene.close()


def test_sponza_texture_path_albedo_normal_position_aovs(grt, oracle):
    """The texture unit (a17): bounce-0 ALBEDO is the anisotropic / trilinear fetch of the diffuse map
    with ray-cone gradients, nothing else; NORMAL and POSITION are the interpolated surface frame.
    One sample, so accumulator == this sample's AOV. Tolerance: 2e-3 absolute on albedo (8-bit texels,
    bilinear weights differ by ulps) for all but 0.1 % of the pixels (LOD / probe-count decisions that
    sit on a rounding edge), 1e-4 relative on position."""
    scene, pt = make_pathtracer(grt, "sponza", 480, 270, 0, num_bounces=2)
    for aov in (grt.AOV_ALBEDO, grt.AOV_NORMAL, grt.AOV_POSITION):
        pt.aov_enable(aov)
    pt.update()
    assert sum(1 for t in pt.textures() if t[0].size > 4) == 19   # the real maps are loaded, not the 1x1 fallback
    view = oracle.SceneView(pt)
    frame = oracle.Frame(view)
    pt.render()
    frame.render_sample(pt.sample_index)
    w = 480
    got_albedo, want_albedo = pt.read_aov(grt.AOV_ALBEDO)[:, :w, :3], frame.accumulator(grt.AOV_ALBEDO)[:, :w, :3]
    assert want_albedo.std() > 0.05                                  # textured, not a constant
    bad = (np.abs(got_albedo - want_albedo).max(axis=2) > 2e-3).mean()
    assert bad < 1e-3, bad
    got_n, want_n = pt.read_aov(grt.AOV_NORMAL)[:, :w, :3], frame.accumulator(grt.AOV_NORMAL)[:, :w, :3]
    assert (np.abs(got_n - want_n).max(axis=2) > 1e-4).mean() < 1e-4
    got_p, want_p = pt.read_aov(grt.AOV_POSITION)[:, :w, :3], frame.accumulator(grt.AOV_POSITION)[:, :w, :3]
    assert (np.abs(got_p - want_p).max(axis=2) > 1e-4 * (1.0 + np.abs(want_p).max(axis=2))).mean() < 1e-4
    pt.close(); scene.close()


@pytest.mark.parametrize("scene_name,w,h,radius", [("cornellbox", 160, 120, 0.5), ("sponza", 320, 180, 2.0)])
def test_ambient_occlusion_integrator_matches_oracle(grt, oracle, scene_name, w, h, radius):
    """The reference's second integrator (AO.cpp / AO.cu) through the host class AO and
    rt_render_ao_sample: 3 progressive samples. Occlusion is binary per sample, so pixels are either
    equal to ~1e-7 or differ by a multiple of 1/n; the occlusion-ray count must agree within the queue
    tolerance and all but 0.1 % of the pixels must agree."""
    grt.config_reset()
    scene = grt.Scene(grt.scene_path(scene_name))
    ao = grt.AO(scene, w, h, device=0, radius=radius)
    ao.aov_enable(grt.AOV_NORMAL); ao.aov_enable(grt.AOV_POSITION)
    ao.update()
    view = oracle.SceneView(ao); frame = oracle.Frame(view)
    for f in range(3):
        if f:
            ao.update()
        ao.render()
        c = ao.counters()
        oc = frame.render_ao_sample(ao.sample_index, radius)
        assert c.trace[0] == oc.trace[0] == w * h
        assert abs(c.shadow[0] - oc.shadow[0]) <= 2 + 0.002 * oc.shadow[0]
    got, want = ao.read_framebuffer()[:, :w, :3], frame.final[:, :w, :3]
    assert got.min() >= 0.0 and got.max() <= 1.0 + 1e-6 and 0.05 < want.mean() < 0.999   # partly occluded
    assert (np.abs(got - want).max(axis=2) > 1e-5).mean() < 1e-3
    for aov, tol in ((grt.AOV_NORMAL, 1e-4), (grt.AOV_POSITION, 1e-3)):
        g, o = ao.read_aov(aov)[:, :w, :3], frame.accumulator(aov)[:, :w, :3]
        assert (np.abs(g - o).max(axis=2) > tol * (1.0 + np.abs(o).max(axis=2))).mean() < 1e-3, aov
    ao.close(); scene.close()


def test_feature_toggles_match_oracle(grt, oracle):
    for cfg in (dict(enable_next_event_estimation=0), dict(enable_multiple_importance_sampling=0), dict(enable_russian_roulette=0), dict(reconstruction_filter=1, enable_mipmapping=0)):
        scene, pt = make_pathtracer(grt, "cornellbox", 96, 64, 0, num_bounces=5, **cfg)
        compare_frames(grt, oracle, pt, 2, 96, 64)
        pt.close(); scene.close()


def test_edge_sizes_and_sample_ranges_match_oracle(grt, oracle):
    """Degenerate launch sizes and the RNG's table boundary, each against the oracle:
    a 1x1 and a 33x17 frame (pitch 64: most of a row is padding), a single bounce, no bounce at all,
    and a batch of samples that straddles sample_index 4095 -> 4096, where random<Dim>() switches
    from the PMJ table to the hash fallback (Sampling.h:48-58)."""
    import ctypes
    for w, h, bounces in ((1, 1, 3), (33, 17, 1), (40, 24, 0)):
        scene, pt = make_pathtracer(grt, "cornellbox", w, h, 0, num_bounces=bounces)
        view = oracle.SceneView(pt); frame = oracle.Frame(view)
        pt.render(); c = pt.counters(); oc = frame.render_sample(pt.sample_index)
        assert list(c.trace[:bounces]) == list(oc.trace[:bounces]), (w, h, bounces)
        got, want = pt.read_framebuffer()[:, :w, :3], frame.final[:, :w, :3]
        assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-5 * (1.0 + np.abs(want).max()), (w, h, bounces)
        pt.close(); scene.close()
    scene, pt = make_pathtracer(grt, "cornellbox", 96, 64, 0, num_bounces=4)
    view = oracle.SceneView(pt); frame = oracle.Frame(view)
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    assert lib.rt_render_samples(pt.ctx, 4094, 4) == 0, lib.rt_last_error(pt.ctx)
    for sample in range(4094, 4098):
        frame.render_sample(sample)
    got, want = pt.read_framebuffer()[:, :96, :3], frame.final[:, :96, :3]
    rel = np.abs(got - want).sum() / want.sum()
    assert rel < REL_L1_TOL, rel
    pt.close(); scene.close()


def test_render_is_deterministic_and_split_invariant(grt):
    """Size-independent properties at the full BASELINE frame size (1920x1080 Sponza):
    the same sample rendered twice is bit-identical, and rendering the frame as two pixel ranges
    (the multi-GPU split) gives bit-identical pixels; ray counts only shrink along the bounces."""
    scene, pt = make_pathtracer(grt, "sponza", 1920, 1080, 0, num_bounces=4)
    pt.render(); a = pt.read_framebuffer().copy(); ca = pt.counters()
    pt.render(); b = pt.read_framebuffer().copy(); cb = pt.counters()
    assert np.array_equal(a, b) and list(ca.trace[:4]) == list(cb.trace[:4])
    assert ca.trace[0] == 1920 * 1080 and all(ca.trace[i + 1] <= ca.trace[i] for i in range(3))
    assert all(ca.shadow[i] <= ca.trace[i] for i in range(4))
    assert np.isfinite(a).all() and a[..., :3].min() >= 0.0

    half = 1920 * 536
    pt.set_pixel_range(0, half); pt.render(); top = pt.read_framebuffer().copy(); c1 = pt.counters()
    pt.set_pixel_range(half, 1920 * 1080 - half); pt.render(); both = pt.read_framebuffer().copy(); c2 = pt.counters()
    assert np.array_equal(both, a)                      # second range filled in the rest
    assert np.array_equal(top[:536], a[:536])
    assert [c1.trace[i] + c2.trace[i] for i in range(4)] == list(ca.trace[:4])
    pt.close(); scene.close()


def test_batch_size_does_not_change_the_image(grt):
    """The reference's BATCH_SIZE loop (Pathtracer.cpp:746-796) vs one whole-frame batch: same pixels."""
    import ctypes
    scene, pt = make_pathtracer(grt, "cornellbox", 300, 200, 0, num_bounces=4)
    lib = grt.device_lib()
    lib.rt_set_batch_size.argtypes = [ctypes.c_void_p, ctypes.c_int]
    pt.render(); whole = pt.read_framebuffer().copy(); c0 = pt.counters()
    assert lib.rt_set_batch_size(pt.ctx, 7777) == 0
    pt.render(); pieces = pt.read_framebuffer().copy(); c1 = pt.counters()
    assert np.array_equal(whole, pieces) and list(c0.trace[:4]) == list(c1.trace[:4])
    pt.close(); scene.close()


def test_samples_in_flight_do_not_change_the_image(grt):
    """Six progressive samples submitted back to back with 1, 2 and 4 samples in flight
    (rt_set_samples_in_flight; shadow rays on the side stream in all of them): the accumulated
    image and the last sample's queue sizes are bit-identical."""
    images, queues = [], []
    for in_flight in (1, 2, 4):
        scene, pt = make_pathtracer(grt, "cornellbox", 320, 240, 0, num_bounces=5)
        grt.set_samples_in_flight(pt.ctx, in_flight)
        for f in range(6):
            if f:
                pt.update()
            pt.render()
        images.append(pt.read_framebuffer().copy())
        c = pt.counters()
        queues.append(list(c.trace[:5]) + list(c.shadow[:5]))
        pt.close(); scene.close()
    assert np.array_equal(images[0], images[1]) and np.array_equal(images[0], images[2])
    assert queues[0] == queues[1] == queues[2] and queues[0][0] == 320 * 240
    assert np.isfinite(images[0]).all() and images[0][..., :3].max() > 0.0


def test_sample_batches_equal_single_samples(grt):
    """rt_render_samples(first, count) renders `count` samples of every pixel as one wavefront
    (virtual pixel index = sample * frame_pixels + pixel). Six samples as 6 x 1, 2 x 3 and 4 + 2:
    the accumulated image is bit-identical and the queue totals add up."""
    import ctypes
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    images, rays = [], []
    for plan in ([1] * 6, [3, 3], [4, 2]):
        scene, pt = make_pathtracer(grt, "cornellbox", 320, 240, 0, num_bounces=5)
        first, total = 0, np.zeros(10, np.int64)
        for count in plan:
            assert lib.rt_render_samples(pt.ctx, first, count) == 0, lib.rt_last_error(pt.ctx)
            c = pt.counters()
            total += np.array(list(c.trace[:5]) + list(c.shadow[:5]), np.int64)
            first += count
        images.append(pt.read_framebuffer().copy()); rays.append(total)
        pt.close(); scene.close()
    assert np.array_equal(images[0], images[1]) and np.array_equal(images[0], images[2])
    assert np.array_equal(rays[0], rays[1]) and np.array_equal(rays[0], rays[2]) and rays[0][0] == 6 * 320 * 240
    # sample batches x pixel batches (rt_set_batch_size): queue capacity is pixels-per-batch x samples
    scene, pt = make_pathtracer(grt, "cornellbox", 320, 240, 0, num_bounces=5)
    lib.rt_set_batch_size.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.rt_set_batch_size(pt.ctx, 5000) == 0
    assert lib.rt_render_samples(pt.ctx, 0, 4) == 0 and lib.rt_render_samples(pt.ctx, 4, 2) == 0
    assert np.array_equal(pt.read_framebuffer(), images[0])
    pt.close(); scene.close()
    # the host class: render_samples(6) == 6 x (update, render), and the progression continues after it
    scene, pt = make_pathtracer(grt, "cornellbox", 320, 240, 0, num_bounces=5)
    pt.render_samples(6)
    assert pt.sample_index == 5 and np.array_equal(pt.read_framebuffer(), images[0])
    pt.update(); assert pt.sample_index == 6
    pt.close(); scene.close()
    # and on the textured 1080p scene, split over tiles as one rank of four would render it
    scene, pt = make_pathtracer(grt, "sponza", 1920, 1080, 0, num_bounces=3)
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
    assert lib.rt_set_pixel_tiles(pt.ctx, 1920 * 8, 1, 4) == 0
    for s in range(4):
        assert lib.rt_render_sample(pt.ctx, s) == 0
    single = pt.read_framebuffer().copy()
    pt.close(); scene.close()
    scene, pt = make_pathtracer(grt, "sponza", 1920, 1080, 0, num_bounces=3)
    assert lib.rt_set_pixel_tiles(pt.ctx, 1920 * 8, 1, 4) == 0
    assert lib.rt_render_samples(pt.ctx, 0, 4) == 0
    batched = pt.read_framebuffer().copy()
    assert np.array_equal(single, batched) and single[8:16, :1920, :3].max() > 0.0 and single[0:8].max() == 0.0
    pt.close(); scene.close()


def test_pixel_query_returns_the_primary_hit(grt, oracle):
    """Integrator::set_pixel_query protocol (Integrator.h:266-277, Integrator.cpp:483-495): armed before
    a render, answered at the update() after it, mesh id translated from TLAS order to the scene's."""
    scene, pt = make_pathtracer(grt, "cornellbox", 200, 160, 0, num_bounces=3)
    view = oracle.SceneView(pt)
    o, d, px = view.generate(pt.sample_index, 0, 200 * 160)
    hits, _ = view.trace(o, d)
    tlas_indices = pt.array("tlas_indices")
    for (x, y) in ((100, 80), (20, 30), (180, 140)):
        pt.set_pixel_query(x, y)
        assert pt.pixel_query[3] == 1                       # pending
        pt.render()
        assert pt.pixel_query[3] == 2                       # output ready
        pt.update()
        _, mesh, tri, status = pt.pixel_query
        assert status == 0
        row = 160 - y                                       # window y is top-down
        ray = row * 200 + x                                 # generate() order: scan lines of `width` pixels
        assert int(px[ray]) == x + row * pt.pitch
        want_mesh, want_tri = int(hits[ray, 0]), int(hits[ray, 1])
        if want_tri == 0xffffffff:
            assert (mesh, tri) == (-1, -1)
        else:
            assert (mesh, tri) == (int(tlas_indices[want_mesh]), want_tri)
    pt.close(); scene.close()


def test_device_errors_are_reported(grt):
    import ctypes
    lib = grt.device_lib()
    ctx = ctypes.c_void_p()
    assert lib.rt_create(0, ctypes.byref(ctx)) == 0
    assert lib.rt_render_sample(ctx, 0) != 0 and b"not uploaded" in lib.rt_last_error(ctx)
    assert lib.rt_create(9999, ctypes.byref(ctypes.c_void_p())) != 0
    lib.rt_destroy(ctx)


def test_tile_split_pack_unpack_rebuilds_the_frame(grt):
    """The multi-GPU path on one GPU: render the tiles of 3 virtual ranks one after the other
    (rt_set_pixel_tiles), pack each rank's tiles, concatenate as an all-gather would, unpack:
    the frame must equal the single-range render bit for bit."""
    import ctypes
    import importlib
    import torch
    parallel = importlib.import_module("gpu_raytracer_amd.parallel")
    W, H, world = 400, 230, 3
    scene, pt = make_pathtracer(grt, "cornellbox", W, H, 0, num_bounces=4)
    lib, ctx = grt.device_lib(), pt.ctx
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_pack_pixels.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_unpack_pixels.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_stream_wait_for_context.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rt_context_wait_for_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    pt.render()
    full = pt.read_framebuffer().copy()
    rays_full = sum(pt.counters().trace[:4])

    gathered = torch.zeros((0, 4), device="cuda")
    rays = 0
    for rank in range(world):
        split = parallel.TileSplit(rank, world, W, H)
        assert lib.rt_set_pixel_tiles(ctx, split.tile_pixels, rank, world) == 0
        pt.render()
        rays += sum(pt.counters().trace[:4])
        packed = torch.zeros((split.local_pixels, 4), device="cuda")
        # the fill runs on torch's stream, the pack on the tracer's: order them on the device, both ways
        torch_stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.rt_context_wait_for_stream(ctx, torch_stream) == 0
        assert lib.rt_pack_pixels(ctx, packed.data_ptr(), split.tile_pixels, rank, world, split.tiles_per_rank) == 0
        assert lib.rt_stream_wait_for_context(ctx, torch_stream) == 0
        gathered = torch.cat([gathered, packed])
        # the torch-side unpack used by bench.py agrees with the device-side one (checked below)
    assert rays == rays_full
    split = parallel.TileSplit(0, world, W, H)
    assert np.array_equal(split.unpack(gathered).cpu().numpy(), full[:, :W, :])
    # wipe the frame, then scatter the gathered tiles back on the device
    lib.rt_set_pixel_range(ctx, 0, 0)
    zeros = torch.zeros_like(gathered)
    torch.cuda.synchronize()
    assert lib.rt_unpack_pixels(ctx, zeros.data_ptr(), split.tile_pixels, world, split.tiles_per_rank) == 0
    assert lib.rt_synchronize(ctx) == 0 and not pt.read_framebuffer().any()
    assert lib.rt_unpack_pixels(ctx, gathered.data_ptr(), split.tile_pixels, world, split.tiles_per_rank) == 0
    assert lib.rt_synchronize(ctx) == 0
    assert np.array_equal(pt.read_framebuffer()[:, :W], full[:, :W])
    pt.close(); scene.close()


def test_trace_statistics_equal_the_oracle_counters(grt, oracle):
    """N_node / N_tri / N_inst of the roofline: the counting kernel visits exactly the nodes and
    triangles the sequential algorithm visits."""
    scene, pt = make_pathtracer(grt, "sponza", 240, 135, 0, num_bounces=3)
    view = oracle.SceneView(pt)
    frame = oracle.Frame(view)
    grt.set_trace_statistics(pt.ctx, True)
    pt.render()
    stats = grt.get_trace_statistics(pt.ctx)
    grt.set_trace_statistics(pt.ctx, False)
    oc = frame.render_sample(pt.sample_index)
    assert stats["closest"]["rays"] == oc.trace_stats.rays == sum(pt.counters().trace[:3])
    for key, ref in (("nodes", oc.trace_stats.nodes), ("triangles", oc.trace_stats.triangles), ("instances_identity", oc.trace_stats.instances_identity)):
        assert abs(stats["closest"][key] - ref) <= 1e-3 * ref, key   # identical up to the few ulp-diverged paths
    assert abs(stats["shadow"]["nodes"] - oc.shadow_stats.nodes) <= 2e-3 * oc.shadow_stats.nodes
    assert abs(stats["closest"]["algorithmic_bytes"] - oc.trace_stats.algorithmic_bytes()) <= 1e-3 * oc.trace_stats.algorithmic_bytes()
    pt.close(); scene.close()


def test_bench_spawns_its_own_ranks_for_a_tile_split_run(grt):
    """`python bench.py --gpus 2` with no launcher around it (the way the driver starts it) re-executes itself under
    torch.distributed.run. Both ranks share GPU 0 here and gather through gloo (BENCH_SHARE_GPU / BENCH_DIST_BACKEND:
    one GPU on the test box); tile split, pack, all-gather, unpack and the max-over-ranks timing all run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["value"] > 0 and out["config"]["ranks"] == 2
    assert out["config"]["rays_per_step"] > 1920 * 1080      # both ranks' rays are in the total


def _render_plan(grt, scene_name, w, h, scheduler, plan, config, prepare=None, aovs=()):
    """Submits `plan` = [(first sample, count), ...] back to back (nothing is read in between) and returns the
    accumulated image, the AOV accumulators, the counters of the last submission and the completion count."""
    import ctypes
    scene, pt = make_pathtracer(grt, scene_name, w, h, 0, **config)
    for aov in aovs:
        pt.aov_enable(aov)
    if aovs:
        pt.update()
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    grt.set_scheduler(pt.ctx, scheduler)
    if prepare:
        prepare(lib, pt)
    completed = []
    for first, count in plan:
        assert lib.rt_render_samples(pt.ctx, first, count) == 0, lib.rt_last_error(pt.ctx)
        completed.append(grt.submissions_completed(pt.ctx))
    image = pt.read_framebuffer().copy()
    extra = [pt.read_aov(aov).copy() for aov in aovs]
    c = pt.counters()
    nb = pt.device_config().num_bounces
    queues = [list(getattr(c, name)[:nb]) for name in ("trace", "shadow", "diffuse", "plastic", "dielectric", "conductor")]
    done = grt.submissions_completed(pt.ctx)
    pt.close(); scene.close()
    return image, extra, queues, completed, done


def test_merged_wavefront_equals_the_slot_scheduler(grt):
    """RT_SCHEDULER_MERGED: consecutive submissions feed one wavefront (every launch carries the rays of all submissions
    in flight, each at its own bounce). Per path nothing changes, so images, AOVs and the per-bounce queue sizes of a
    submission are bit-identical to the slot scheduler's -- with submissions of different sizes back to back, whole
    frames restarted at sample 0, a pixel range, a tile split, and more submissions than the wavefront admits at once."""
    import ctypes
    cases = [
        ("cornellbox", 320, 240, [(0, 1), (1, 4), (5, 2), (7, 3), (10, 1), (11, 1), (12, 4)], dict(num_bounces=5), None, ()),
        ("sponza", 640, 360, [(0, 4), (0, 4), (0, 4)], dict(num_bounces=6), None, (grt.AOV_ALBEDO, grt.AOV_NORMAL, grt.AOV_POSITION)),
        ("cornellbox", 200, 150, [(s, 1) for s in range(40)], dict(num_bounces=12), None, ()),   # 40 submissions, 12 in flight
    ]
    def tiles(lib, pt):
        lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
        assert lib.rt_set_pixel_tiles(pt.ctx, 640 * 8, 1, 4) == 0
    def pixel_range(lib, pt):
        assert lib.rt_set_pixel_range(pt.ctx, 640 * 100 + 17, 640 * 120 + 5) == 0
    cases.append(("sponza", 640, 360, [(0, 2), (2, 2), (4, 4)], dict(num_bounces=4), tiles, ()))
    # ragged in both directions: the merged wavefront walks the frame in 8 x 8 patches along bands of 8 scan lines (kernel_generate_stream) -- 203 = 25 patches + 3 columns,
    # 157 = 19 bands + 5 lines --, whole and as rank 1 of 3 with 8-row tiles (whose last tile is the clipped band); the slot scheduler walks scan lines
    def ragged_tiles(lib, pt):
        lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
        assert lib.rt_set_pixel_tiles(pt.ctx, 203 * 8, 1, 3) == 0
    cases.append(("cornellbox", 203, 157, [(0, 3), (3, 2)], dict(num_bounces=5), None, ()))
    cases.append(("cornellbox", 203, 157, [(0, 3), (3, 2)], dict(num_bounces=5), ragged_tiles, ()))
    cases.append(("sponza", 640, 360, [(0, 3), (3, 1)], dict(num_bounces=4), pixel_range, ()))
    for scene_name, w, h, plan, config, prepare, aovs in cases:
        merged = _render_plan(grt, scene_name, w, h, "merged", plan, config, prepare, aovs)
        slots = _render_plan(grt, scene_name, w, h, "slots", plan, config, prepare, aovs)
        label = (scene_name, plan[:3])
        assert np.array_equal(merged[0], slots[0]) and merged[0][..., :3].max() > 0.0, label
        for a, b in zip(merged[1], slots[1]):
            assert np.array_equal(a, b), label
        assert merged[2] == slots[2], (label, merged[2], slots[2])
        # submissions complete in order while later ones are made; reading completes the rest
        assert merged[3] == sorted(merged[3]) and merged[3][-1] <= len(plan), (label, merged[3])
        assert merged[4] == len(plan), label


def test_small_pipelined_submissions_share_an_iteration(grt):
    """The tiles of one rank of an 8-way split are 1/8 of a frame: with frame pipelining on, such submissions generate
    their rays at once but wait for company, so that the iterations stay as large as those of a whole frame
    (RT_STREAM_BATCH_PATHS). Nothing completes until the iteration is enqueued -- by the submission that fills the batch,
    by rt_advance, by a camera change or by any call that flushes -- and the image and the per-bounce ray counts are those
    of the same submissions made one per iteration."""
    import ctypes
    W, H, BOUNCES, FRAMES = 192, 128, 5, 11
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3

    def render(pipelining):
        scene, pt = make_pathtracer(grt, "cornellbox", W, H, 0, num_bounces=BOUNCES)
        assert lib.rt_set_pixel_tiles(pt.ctx, W * 8, 1, 8) == 0      # rank 1 of 8: tiles 1, 9 of 16
        grt.set_frame_pipelining(pt.ctx, pipelining)
        completed, counters = [], None
        for f in range(FRAMES):
            assert lib.rt_render_samples(pt.ctx, 2 * f, 2) == 0
            completed.append(grt.submissions_completed(pt.ctx))
        if pipelining:
            # 8 submissions per iteration: #0-7 entered iteration 0 together, #8-10 still wait for theirs
            assert completed == [0] * FRAMES
            for k in range(BOUNCES - 1):
                grt.advance(pt.ctx)
            assert grt.submissions_completed(pt.ctx) == 8          # iterations 0..4 done: the first batch is complete
            grt.advance(pt.ctx)
            assert grt.submissions_completed(pt.ctx) == FRAMES     # #8-10 entered iteration 1
        counters = pt.counters()
        image = pt.read_framebuffer().copy()
        trace = list(counters.trace[:BOUNCES]), list(counters.shadow[:BOUNCES])
        pt.close(); scene.close()
        return image, trace

    batched, batched_counts = render(True)
    single, single_counts = render(False)
    assert np.array_equal(batched, single)
    assert batched_counts == single_counts
    assert np.abs(batched[8:16, :W]).max() > 0 and np.abs(batched[0:8, :W]).max() == 0   # only this rank's tiles

    # a camera change enqueues the waiting submissions first: they are shaded with the camera they were generated with
    scene, pt = make_pathtracer(grt, "cornellbox", W, H, 0, num_bounces=BOUNCES)
    assert lib.rt_set_pixel_tiles(pt.ctx, W * 8, 1, 8) == 0
    grt.set_frame_pipelining(pt.ctx, True)
    assert lib.rt_render_samples(pt.ctx, 0, 2) == 0
    scene.set_camera((0.2, 1.0, 6.5), (0.0, 0.0, 0.0, 1.0)); pt.update()   # -> iteration 0
    for k in range(BOUNCES - 1):                                            # iterations 1..4
        grt.advance(pt.ctx)
    assert grt.submissions_completed(pt.ctx) == 1
    pt.close(); scene.close()


def test_a_declared_burst_of_whole_frames_shares_iterations(grt):
    """rt_set_stream_batch: an application that submits a burst of WHOLE frames and reads nothing in between declares the burst's
    paths; its submissions then wait for one another, enter the wavefront together and complete together after num_bounces
    iterations (not submissions + num_bounces - 1) -- with the image and the per-bounce ray counts of the same submissions made
    one per iteration. A sixth submission that no longer fits the declared burst starts the next iteration."""
    import ctypes
    W, H, BOUNCES, FRAMES, SPP = 192, 128, 5, 5, 2
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]

    def render(burst, frames=FRAMES):
        scene, pt = make_pathtracer(grt, "cornellbox", W, H, 0, num_bounces=BOUNCES)
        if burst:
            grt.set_frame_pipelining(pt.ctx, True)
            grt.set_stream_batch(pt.ctx, FRAMES * SPP * W * H)
        completed = []
        for f in range(frames):
            assert lib.rt_render_samples(pt.ctx, SPP * f, SPP) == 0
            completed.append(grt.submissions_completed(pt.ctx))
        if burst:
            assert completed == [0] * frames                       # the fifth submission enqueued iteration 0 for all five
            for k in range(BOUNCES - 2):
                grt.advance(pt.ctx)
            assert grt.submissions_completed(pt.ctx) == 0          # iterations 0 .. 3: nobody has passed the last bounce
            grt.advance(pt.ctx)
            assert grt.submissions_completed(pt.ctx) == FRAMES     # iteration 4: all five at once
            if frames > FRAMES:
                grt.advance(pt.ctx)                                # the sixth entered iteration 1 (enqueued by the first rt_advance)
                assert grt.submissions_completed(pt.ctx) == frames
        counters = pt.counters()
        image = pt.read_framebuffer().copy()
        counts = list(counters.trace[:BOUNCES]), list(counters.shadow[:BOUNCES])
        pt.close(); scene.close()
        return image, counts

    together, together_counts = render(True)
    one_by_one, one_by_one_counts = render(False)
    assert np.isfinite(together).all() and together[..., :3].max() > 0.0
    assert np.array_equal(together, one_by_one)
    assert together_counts == one_by_one_counts
    six, _ = render(True, FRAMES + 1)
    six_one_by_one, _ = render(False, FRAMES + 1)
    assert np.array_equal(six, six_one_by_one)


def test_merged_wavefront_advances_without_new_samples(grt):
    """rt_advance runs one iteration without new samples: a frame loop learns from rt_submissions_completed when a frame
    may be packed. With frame pipelining on, rt_pack_pixels follows the completed submissions only."""
    import ctypes
    scene, pt = make_pathtracer(grt, "cornellbox", 160, 120, 0, num_bounces=6)
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    assert lib.rt_render_samples(pt.ctx, 0, 2) == 0 and lib.rt_render_samples(pt.ctx, 2, 2) == 0
    assert grt.submissions_completed(pt.ctx) == 0
    for k in range(4):
        grt.advance(pt.ctx)
    assert grt.submissions_completed(pt.ctx) == 1      # born at iteration 0, last bounce at iteration 5
    grt.advance(pt.ctx)
    assert grt.submissions_completed(pt.ctx) == 2
    grt.advance(pt.ctx)                                 # nothing in flight: a no-op
    assert grt.submissions_completed(pt.ctx) == 2
    four = pt.read_framebuffer().copy()
    pt.close(); scene.close()
    scene, pt = make_pathtracer(grt, "cornellbox", 160, 120, 0, num_bounces=6)
    for s in range(4):
        assert lib.rt_render_samples(pt.ctx, s, 1) == 0
    assert np.array_equal(pt.read_framebuffer(), four)
    pt.close(); scene.close()
