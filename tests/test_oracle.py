"""Pins the CPU oracle itself (no GPU): traversal against brute force, hashing against an
independent numpy restatement, and the committed golden render."""
import os

import numpy as np
import pytest

from conftest import make_pathtracer, unpack_hits

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "render_golden.npz")


def brute_force(tris, origins, dirs):
    """Double-precision Moeller-Trumbore against every triangle (closest t per ray)."""
    p0, e1, e2 = tris[:, 0:3].astype(np.float64), tris[:, 3:6].astype(np.float64), tris[:, 6:9].astype(np.float64)
    best_t = np.full(origins.shape[1], np.inf)
    best_id = np.full(origins.shape[1], -1)
    for r in range(origins.shape[1]):
        o, d = origins[:, r].astype(np.float64), dirs[:, r].astype(np.float64)
        h = np.cross(d, e2)
        a = (e1 * h).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            f = 1.0 / a
            s = o - p0
            u = f * (s * h).sum(1)
            q = np.cross(s, e1)
            v = f * (q * d).sum(1)
            t = f * (e2 * q).sum(1)
        ok = (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0)
        if ok.any():
            tt = np.where(ok, t, np.inf)
            best_id[r] = int(tt.argmin()); best_t[r] = tt.min()
    return best_id, best_t


def test_bvh8_and_bvh2_traversal_agree_with_brute_force(grt, oracle):
    scene, pt = make_pathtracer(grt, "cornellbox", 32, 32, -1)
    view8 = oracle.SceneView(pt, bvh_type=8)
    o, d, _ = view8.generate(0, 0, 32 * 32)
    rng = np.random.default_rng(3)
    o2 = np.tile(np.array([[0.1], [1.0], [0.2]], np.float32), (1, 500))
    d2 = rng.normal(size=(3, 500)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=0)
    O, D = np.concatenate([o, o2], 1), np.concatenate([d, d2], 1)
    hits8, stats8 = view8.trace(O, D)
    tris = pt.array("triangles").reshape(-1, 24)
    bid, bt = brute_force(tris, O, D)
    _, tid, t, _, _ = unpack_hits(hits8)
    miss = bid < 0
    assert ((tid == -1) == miss).all()
    assert np.allclose(t[~miss], bt[~miss], rtol=2e-5)
    same = tid[~miss] == bid[~miss]
    assert same.mean() > 0.995    # ties on shared edges may pick the neighbour triangle
    assert stats8.rays == O.shape[1] and stats8.nodes > 0

    # binary BVH (BASELINE config #1): same closest hits
    pt.close(); scene.close()
    scene, pt = make_pathtracer(grt, "cornellbox", 32, 32, -1, bvh_type=2)
    view2 = oracle.SceneView(pt, bvh_type=2)
    hits2, _ = view2.trace(O, D)
    _, tid2, t2, _, _ = unpack_hits(hits2)
    assert ((tid2 == -1) == miss).all() and np.allclose(t2[~miss], bt[~miss], rtol=2e-5)
    # shadow rays: occluded iff brute force finds a hit closer than max_distance
    md = np.full(O.shape[1], 3.0, np.float32)
    occ, _ = view2.trace_shadow(O, D, md)
    occ8, _ = view8.trace_shadow(O, D, md)
    expect = (~miss) & (bt < 3.0)
    assert (occ.astype(bool) == expect).mean() > 0.999 and (occ8 == occ).all()
    pt.close(); scene.close()

    # 4-wide BVH (BVH4.h): same closest hits, same occlusion; the host uses the BVH2 triangle order for it
    scene, pt = make_pathtracer(grt, "cornellbox", 32, 32, -1, bvh_type=4)
    view4 = oracle.SceneView(pt, bvh_type=4)
    hits4, stats4 = view4.trace(O, D)
    assert np.array_equal(hits4[:, 1:], hits2[:, 1:]) and stats4.nodes > 0   # triangle id, t bits, u/v: identical to the binary BVH
    occ4, _ = view4.trace_shadow(O, D, md)
    assert (occ4 == occ).all()
    pt.close(); scene.close()


def test_instanced_traversal_uses_object_space_rays(grt, oracle, tmp_path):
    """Non-identity instance: transform_inv applied without renormalising, so t is world distance (BVH8.h:222-228)."""
    (tmp_path / "tri.obj").write_text("v -1 -1 0\nv 1 -1 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "s.xml").write_text("""<scene version="0.5.0"><shape type="obj"><string name="filename" value="tri.obj"/>
      <transform name="toWorld"><scale value="3"/><translate z="-10"/></transform><bsdf type="diffuse"/></shape></scene>""")
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "s.xml"))
    pt = grt.Pathtracer(scene, 16, 16, device=-1); pt.update()
    view = oracle.SceneView(pt)
    o = np.array([[0.0], [0.0], [5.0]], np.float32); d = np.array([[0.0], [0.0], [-1.0]], np.float32)
    hits, stats = view.trace(o, d)
    mesh, tri, t, u, v = unpack_hits(hits)
    assert tri[0] == 0 and abs(t[0] - 15.0) < 1e-5 and stats.instances_transformed == 1
    pt.close(); scene.close()


# ---- independent numpy restatement of the integer sample-index path (Util.h:104-149, Sampling.h:44-84)
def np_pcg(seed):
    seed = np.uint32(seed)
    with np.errstate(over="ignore"):
        state = seed * np.uint32(747796405) + np.uint32(2891336453)
        word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
    return (word >> np.uint32(22)) ^ word


def np_permute(index, length, seed):
    u = np.uint32
    mask = u(length - 1); index = u(index); seed = u(seed)
    with np.errstate(over="ignore"):
        index ^= seed; index *= u(0xe170893d); index ^= seed >> u(16); index ^= (index & mask) >> u(4)
        index ^= seed >> u(8); index *= u(0x0929eb3f); index ^= seed >> u(23); index ^= (index & mask) >> u(1)
        index *= u(1) | seed >> u(27); index *= u(0x6935fa69); index ^= (index & mask) >> u(11)
        index *= u(0x74dcb303); index ^= (index & mask) >> u(2); index *= u(0x9e501cc3); index ^= (index & mask) >> u(2)
        index *= u(0xc860a3df); index &= mask; index ^= index >> u(5)
        return (index + seed) & mask


def test_random_sample_index_path_is_bit_exact(grt, oracle):
    scene, pt = make_pathtracer(grt, "cornellbox", 200, 100, -1)
    view = oracle.SceneView(pt)
    pmj = pt.array("pmj_samples").reshape(64, 4096, 2)
    bn = pt.array("blue_noise").reshape(16, 128, 128, 2)
    pitch = pt.pitch
    rng = np.random.default_rng(7)
    for dim_enum, bounce, sample in ((0, 0, 0), (5, 2, 17), (6, 12, 4095), (2, 20, 1000), (4, 127, 3)):
        px = rng.integers(0, pitch * 100, 64).astype(np.uint32)
        got = view.random(dim_enum, px, bounce, sample)
        for k, pixel in enumerate(px):
            with np.errstate(over="ignore"):
                h = np_pcg((np.uint32(pixel) * np.uint32(7) + np.uint32(dim_enum)) * np.uint32(128) + np.uint32(bounce))
            dim = dim_enum + 5 * bounce
            si = sample if dim < 64 else int(np_permute(sample, 4096, h))
            s = pmj[dim % 64, si].copy()
            x, y = (int(pixel) % pitch) % 128, (int(pixel) // pitch) % 128
            b = bn[dim % 16, y, x].astype(np.float32) * np.float32(1.0 / 255.0)
            s = s + b
            s = np.where(s >= 1.0, s - np.float32(1.0), s).astype(np.float32)
            assert got[k].view(np.uint32).tolist() == s.view(np.uint32).tolist()
    pt.close(); scene.close()


def test_pmj_table_is_a_stratified_02_sequence(grt):
    """Every power-of-two prefix of each sequence has one sample per elementary interval (pmj02 property)."""
    scene, pt = make_pathtracer(grt, "cornellbox", 32, 32, -1)
    pmj = pt.array("pmj_samples").reshape(64, 4096, 2)
    assert pmj.min() >= 0.0 and pmj.max() < 1.0
    for seq in (0, 1, 31, 63):
        for log_n in (2, 4, 6, 10, 12):
            n = 1 << log_n
            pts = pmj[seq, :n].astype(np.float64)
            for a in range(log_n + 1):           # strata of 2^a x 2^(log_n - a)
                nx, ny = 1 << a, 1 << (log_n - a)
                cell = np.floor(pts[:, 0] * nx).astype(int) * ny + np.floor(pts[:, 1] * ny).astype(int)
                assert np.unique(cell).size == n, (seq, log_n, a)
    pt.close(); scene.close()


@pytest.mark.reference_layout
def test_oracle_matches_committed_golden_render(grt, oracle):
    g = np.load(GOLDEN)
    scene, pt = make_pathtracer(grt, "cornellbox", 48, 48, -1, num_bounces=4)
    view = oracle.SceneView(pt)
    o, d, _ = view.generate(0, 0, 48 * 48)
    assert np.array_equal(o, g["ray_origin"]) and np.allclose(d, g["ray_direction"], atol=2e-7)
    hits, stats = view.trace(g["ray_origin"], g["ray_direction"])
    assert np.array_equal(hits, g["hits"]) and stats.nodes == int(g["nodes"]) and stats.triangles == int(g["triangles"])
    frame = oracle.Frame(view)
    for s in range(2):
        c = frame.render_sample(s)
        assert (list(c.trace[:4]) + list(c.shadow[:4])) == g["counters"][s].tolist()
    rel = np.abs(frame.final[:, :48, :3] - g["image"]).sum() / g["image"].sum()
    assert rel < 1e-5
    # physical sanity of the golden: light visible, image finite, red/green wall colour bleeding present
    img = g["image"]
    assert np.isfinite(img).all() and img.max() > 10.0 and img[:, :8, 0].mean() != img[:, -8:, 0].mean()
    pt.close(); scene.close()


def test_accumulation_quirk_divides_by_sample_index(grt, oracle):
    """Frame 1 overwrites frame 0 (AOV.h:35-41 divides by n, not n+1): acc after samples 0,1 == sample 1."""
    scene, pt = make_pathtracer(grt, "cornellbox", 24, 24, -1, num_bounces=3)
    view = oracle.SceneView(pt)
    a = oracle.Frame(view); a.render_sample(0); a.render_sample(1)
    b = oracle.Frame(view); b.render_sample(1)
    assert np.allclose(a.final, b.final, rtol=1e-5, atol=2e-5)  # acc + (fb - acc) / 1 rounds at the scale of the overwritten sample
    pt.close(); scene.close()


def test_oracle_ambient_occlusion_properties(grt, oracle):
    """AO integrator restatement (AO.cu): occlusion is 0/1 per sample; an occlusion ray of
    (almost) zero length always escapes, so every pixel that hits geometry gets exactly 1 and
    every other pixel 0; longer rays can only be occluded more often."""
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("cornellbox"))
    ao = grt.AO(scene, 48, 36, device=-1)
    ao.update()
    view = oracle.SceneView(ao)
    means = []
    for radius in (1e-4, 0.25, 1.0, 100.0):
        frame = oracle.Frame(view)
        c = frame.render_ao_sample(0, radius)
        img = frame.final[:, :48, 0]
        assert set(np.unique(img)).issubset({0.0, 1.0}) and c.trace[0] == 48 * 36 and c.shadow[0] <= c.trace[0]
        if radius == 1e-4:
            assert int(img.sum()) == c.shadow[0]
        means.append(float(img.mean()))
    assert means[0] >= means[1] >= means[2] >= means[3] and means[3] < means[0]
    # three samples average to multiples of 1/3 (frame 1 overwrites frame 0: AOV.h:35-46 divides by sample_index)
    frame = oracle.Frame(view)
    for s in range(4):
        frame.render_ao_sample(s, 0.5)
    vals = np.unique(np.round(frame.final[:, :48, 0] * 3.0, 4))
    assert set(vals).issubset({0.0, 1.0, 2.0, 3.0})
    ao.close(); scene.close()


FURNACE_XML = ('<scene version="0.5.0"><integrator type="path"><integer name="maxDepth" value="24"/></integrator>'
               '<sensor type="perspective"><float name="fov" value="30"/><transform name="toWorld"><lookat origin="0, 0, 6" target="0, 0, 0" up="0, 1, 0"/></transform></sensor>'
               '<shape type="sphere"><float name="radius" value="1"/>%(bsdf)s</shape>'
               '<shape type="sphere"><float name="radius" value="0.6"/><transform name="toWorld"><translate x="1.2" y="0.8" z="0.5"/></transform>%(bsdf)s</shape></scene>')


@pytest.mark.parametrize("bsdf,lo,hi", [
    ('<bsdf type="diffuse"><rgb name="reflectance" value="1, 1, 1"/></bsdf>', 0.998, 1.0001),
    ('<bsdf type="roughplastic"><rgb name="diffuseReflectance" value="1, 1, 1"/><float name="alpha" value="0.3"/></bsdf>', 0.97, 1.01)])
def test_white_furnace(grt, oracle, tmp_path, bsdf, lo, hi):
    """Physical pin of the BSDF restatement, independent of any reference output: two white objects
    under the constant white sky, Russian roulette off -- every path that escapes carries
    throughput = product of f * cos / pdf. A lossless diffuse surface must return exactly the sky
    (1.0 up to the paths cut at 24 bounces); the plastic of BSDF.h:72-190 is built to conserve energy.
    A mismatch between a sample() and its pdf, or between eval() and the sampled lobe, shows up here."""
    (tmp_path / "f.xml").write_text(FURNACE_XML % {"bsdf": bsdf})
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "f.xml"))
    grt.config_set(enable_russian_roulette=0)
    pt = grt.Pathtracer(scene, 40, 40, device=-1); pt.update()
    frame = oracle.Frame(oracle.SceneView(pt))
    for s in range(17):          # sample 0 is overwritten by sample 1 (AOV.h:35-46): 16 samples count
        frame.render_sample(s)
    img = frame.final[:, :40, :3]
    assert lo <= img.mean() <= hi, img.mean()
    assert np.isfinite(img).all() and img.min() > 0.7
    pt.close(); scene.close()


FORM_FACTOR_XML = ('<scene version="0.5.0"><integrator type="path"><integer name="maxDepth" value="2"/></integrator>'
                   '<sensor type="perspective"><float name="fov" value="2"/><transform name="toWorld"><lookat origin="0.001, 0.5, 0" target="0, 0, 0" up="0, 0, 1"/></transform></sensor>'
                   '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="-90"/><scale value="20"/></transform><bsdf type="diffuse"><rgb name="reflectance" value="1, 1, 1"/></bsdf></shape>'
                   '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><scale value="0.5"/><translate y="1"/></transform><emitter type="area"><rgb name="radiance" value="1, 1, 1"/></emitter></shape></scene>')


@pytest.mark.parametrize("toggles", [{}, {"enable_multiple_importance_sampling": 0}, {"enable_next_event_estimation": 0}],
                         ids=["nee+mis", "nee", "bsdf-sampling"])
def test_direct_lighting_matches_the_analytic_form_factor(grt, oracle, tmp_path, toggles):
    """Physical pin of the light sampling (Pathtracer.cu:354-422, 1040-1120): a white diffuse floor
    seen from straight above, lit by a 1x1 unit-radiance square 1 above it. The radiance leaving the
    point under the light's centre is the form factor of the square, 4 * F(1/2, 1/2) with
    F(X, Y) = (X/sqrt(1+X^2) atan(Y/sqrt(1+X^2)) + Y/sqrt(1+Y^2) atan(X/sqrt(1+Y^2))) / (2 pi)
    = 0.23944. All three estimators the config can select (light sampling with and without MIS,
    BSDF sampling alone) must converge to it: a wrong light pdf, MIS weight or double count does not."""
    x = 0.5 / np.sqrt(1.25)
    analytic = 4.0 * (2.0 * x * np.arctan(x)) / (2.0 * np.pi)
    (tmp_path / "ff.xml").write_text(FORM_FACTOR_XML)
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "ff.xml"))
    scene.set_sky_scale(0.0)     # the default constant sky would light the floor as well
    grt.config_set(num_bounces=2, enable_russian_roulette=0, **toggles)
    pt = grt.Pathtracer(scene, 12, 12, device=-1); pt.update()
    frame = oracle.Frame(oracle.SceneView(pt))
    for s in range(513):
        frame.render_sample(s)
    mean = float(frame.final[:, :12, :3].mean())
    assert abs(mean - analytic) < 0.01 * analytic, (mean, analytic)
    pt.close(); scene.close(); grt.config_reset()


def write_sliver_scene(tmp_path, n=600, seed=9):
    """A file-loaded mesh of long thin overlapping triangles (what spatial splits are for), placed twice."""
    rng = np.random.default_rng(seed)
    p0 = rng.random((n, 3)) * 4 - 2
    p1 = p0 + rng.random((n, 3)) * 4 - 2
    p2 = p0 + rng.random((n, 3)) * 0.3
    with open(tmp_path / "slivers.obj", "w") as f:
        for a, b, c in zip(p0, p1, p2):
            f.write("v %.6f %.6f %.6f\nv %.6f %.6f %.6f\nv %.6f %.6f %.6f\n" % (*a, *b, *c))
        for i in range(n):
            f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    (tmp_path / "s.xml").write_text('<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="60"/><transform name="toWorld"><lookat origin="0, 0, 9" target="0, 0, 0" up="0, 1, 0"/></transform></sensor>'
                                    '<shape type="obj"><string name="filename" value="slivers.obj"/></shape>'
                                    '<shape type="obj"><string name="filename" value="slivers.obj"/><transform name="toWorld"><rotate y="1" angle="40"/><translate x="5" y="0.5" z="-1"/></transform></shape></scene>')
    return str(tmp_path / "s.xml")


@pytest.mark.reference_layout
def test_spatial_split_and_collapsed_trees_trace_like_the_cwbvh(grt, oracle, tmp_path):
    """bvh_type = SBVH / BVH / BVH4 on a file-loaded mesh: leaves hold several triangles
    (BVHCollapser.cpp) and, with spatial splits, a triangle sits in several leaves (the device
    triangle array then has one copy per reference, Integrator.cpp:128-152). Every tree must find
    the same closest hits (t bit-exact: the same Moeller-Trumbore on the same triangle data) and
    the same occlusion as the CWBVH."""
    path = write_sliver_scene(tmp_path)
    rng = np.random.default_rng(5)
    results = {}
    for bvh_type in (8, 2, 1, 4):
        grt.config_reset()
        scene = grt.Scene(path)
        grt.config_set(bvh_type=bvh_type)
        pt = grt.Pathtracer(scene, 24, 24, device=-1); pt.update()
        view = oracle.SceneView(pt, bvh_type={8: 8, 2: 2, 1: 2, 4: 4}[bvh_type])
        if bvh_type == 8:
            o, d, _ = view.generate(0, 0, 24 * 24)
            o2 = rng.normal(size=(3, 800)).astype(np.float32) * 3
            d2 = rng.normal(size=(3, 800)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=0)
            O, D = np.ascontiguousarray(np.concatenate([o, o2], 1)), np.ascontiguousarray(np.concatenate([d, d2], 1))
        hits, stats = view.trace(O, D)
        occ, _ = view.trace_shadow(O, D, np.full(O.shape[1], 2.5, np.float32))
        tris = pt.array("triangles").reshape(-1, 24)
        mesh_id, tid, t, u, v = unpack_hits(hits)
        results[bvh_type] = dict(t=t.copy(), hit=tid >= 0, occ=occ.copy(), tri_count=tris.shape[0],
                                 p0=np.where((tid >= 0)[:, None], tris[np.maximum(tid, 0), 0:3], 0), nodes=stats.nodes)
        pt.close(); scene.close()
    want = results[8]
    assert want["hit"].mean() > 0.2
    assert results[2]["tri_count"] == want["tri_count"] == results[4]["tri_count"] == 600
    assert results[1]["tri_count"] > 600                      # references duplicated by spatial splits
    for bvh_type in (2, 1, 4):
        got = results[bvh_type]
        assert np.array_equal(got["hit"], want["hit"]) and np.array_equal(got["occ"], want["occ"]), bvh_type
        assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32)), bvh_type
        same_triangle = (got["p0"] == want["p0"]).all(axis=1)
        assert same_triangle.mean() > 0.995, bvh_type          # equal-t ties between overlapping slivers may resolve differently
    grt.config_reset()


def _rectangle_form_factor(x0, x1, z0, z1, h):
    """Form factor from a horizontal rectangle [x0, x1] x [z0, z1] at height h to the point below the origin
    (parallel differential element): inclusion-exclusion over the closed form for a corner rectangle."""
    def corner(x, z):
        a, b = np.sqrt(h * h + x * x), np.sqrt(h * h + z * z)
        return (x / a * np.arctan(z / a) + z / b * np.arctan(x / b)) / (2.0 * np.pi)
    return corner(x1, z1) - corner(x0, z1) - corner(x1, z0) + corner(x0, z0)


@pytest.mark.parametrize("toggles", [{}, {"enable_multiple_importance_sampling": 0}, {"enable_next_event_estimation": 0}],
                         ids=["nee+mis", "nee", "bsdf-sampling"])
def test_two_lights_of_different_power_size_and_instance_scale(grt, oracle, tmp_path, toggles):
    """Pins the two-level light CDF and its pdf (Pathtracer.cpp:384-534 on the host, Pathtracer.cu:465-555 on the
    device): mesh weight = luminance x area x scale^2, triangle picked by area, pdf = power d^2 / (cos total).
    A unit-radiance 1x1 shape at height 1 and a three-times brighter file-loaded quad that is placed off-centre
    through an instance transform with scale 0.5 must add up to L1 F1 + L2 F2 at the floor point under the
    origin for every estimator; a weight that ignored the instance scale or the emission would bias NEE."""
    (tmp_path / "quad.obj").write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nf 1 2 3\nf 1 3 4\n")
    (tmp_path / "s.xml").write_text(
        '<scene version="0.5.0"><integrator type="path"><integer name="maxDepth" value="2"/></integrator>'
        '<sensor type="perspective"><float name="fov" value="2"/><transform name="toWorld"><lookat origin="0.001, 0.5, 0" target="0, 0, 0" up="0, 0, 1"/></transform></sensor>'
        '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="-90"/><scale value="20"/></transform><bsdf type="diffuse"><rgb name="reflectance" value="1, 1, 1"/></bsdf></shape>'
        '<shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><scale value="0.5"/><translate y="1"/></transform><emitter type="area"><rgb name="radiance" value="1, 1, 1"/></emitter></shape>'
        '<shape type="obj"><string name="filename" value="quad.obj"/><transform name="toWorld"><scale value="0.5"/><translate x="1.5" y="0.8" z="0.25"/></transform>'
        '<emitter type="area"><rgb name="radiance" value="3, 3, 3"/></emitter></shape></scene>')
    analytic = 1.0 * _rectangle_form_factor(-0.5, 0.5, -0.5, 0.5, 1.0) + 3.0 * _rectangle_form_factor(1.0, 2.0, -0.25, 0.75, 0.8)
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "s.xml"))
    scene.set_sky_scale(0.0)
    grt.config_set(num_bounces=2, enable_russian_roulette=0, **toggles)
    pt = grt.Pathtracer(scene, 12, 12, device=-1); pt.update()
    assert abs(scene.mesh_transform(2)[2] - 0.5) < 1e-6                        # the quad is an instance with scale 0.5, not baked
    frame = oracle.Frame(oracle.SceneView(pt))
    for s in range(769):
        frame.render_sample(s)
    mean = float(frame.final[:, :12, :3].mean())
    assert abs(mean - analytic) < 0.012 * analytic, (mean, analytic)
    pt.close(); scene.close(); grt.config_reset()


def test_russian_roulette_does_not_change_the_expectation(grt, oracle):
    """Russian roulette (Pathtracer.cu:199-218) terminates paths with probability 1 - p and divides the survivors by
    p: the mean image must not move (8 bounces, so that most paths reach the bounces where it is active)."""
    means = {}
    for label, cfg in (("rr", {}), ("no rr", dict(enable_russian_roulette=0))):
        scene, pt = make_pathtracer(grt, "cornellbox", 20, 20, -1, num_bounces=8, **cfg)
        frame = oracle.Frame(oracle.SceneView(pt))
        for s in range(385):
            frame.render_sample(s)
        means[label] = float(frame.final[:, :20, :3].mean())
        pt.close(); scene.close()
    grt.config_reset()
    assert abs(means["rr"] - means["no rr"]) < 0.02 * means["no rr"], means


def _reference_frame(oracle, view):
    if oracle.ref_lib() is None or not hasattr(oracle.ref_lib(), "ref_cuda_frame_create"):
        pytest.skip("oracle/_ref not built with the reference's device code (no /root/reference on this machine)")
    return oracle.ReferenceFrame(view)


@pytest.mark.reference_layout
def test_oracle_equals_the_references_own_kernels_run_on_the_cpu(grt, oracle):
    """THE pin of the restated device path: the reference's Pathtracer.cu (every kernel and header, compiled
    verbatim for the host through oracle/ref/cuda_shim and executed one CUDA thread at a time by
    oracle/ref/ref_cuda_harness.cpp) renders the Cornell box from the same staged arrays as the oracle.
    Queue sizes per bounce must be identical and the frames agree to float noise: the two differ only in where
    the compilers fuse multiply-adds."""
    for config in (dict(num_bounces=5), dict(num_bounces=3, enable_next_event_estimation=0), dict(num_bounces=3, enable_multiple_importance_sampling=0),
                   dict(num_bounces=6, enable_russian_roulette=0, reconstruction_filter=0), dict(num_bounces=4, reconstruction_filter=1)):
        scene, pt = make_pathtracer(grt, "cornellbox", 64, 48, -1, **config)
        view = oracle.SceneView(pt)
        ours, theirs = oracle.Frame(view), _reference_frame(oracle, view)
        nb = config["num_bounces"]
        for s in range(3):
            oc, rc = ours.render_sample(s), theirs.render_sample(s)
            for queue in ("trace", "shadow", "diffuse"):
                got, want = list(getattr(oc, queue)[:nb]), [int(v) for v in rc[queue][:nb]]
                assert all(abs(a - b) <= 1 + 0.002 * b for a, b in zip(got, want)), (config, s, queue, got, want)
            a, b = ours.final[:, :64, :3], theirs.final[:, :64, :3]
            assert np.abs(a - b).sum() / b.sum() < 2e-5, (config, s)
            assert (np.abs(a - b).max(axis=2) > 0.01 * (b.max(axis=2) + 1e-3)).mean() < 2e-3, (config, s)
        theirs.close(); pt.close(); scene.close()
    grt.config_reset()


def _synthetic_luts(seed=5):
    """Smooth tables in (0, 1) with the shapes of the Kulla-Conty LUTs: both renderers read the same numbers, so
    they need not be the true albedos (integrating those on the CPU takes minutes)."""
    rng = np.random.default_rng(seed)
    def smooth(shape):
        grid = np.meshgrid(*[np.linspace(0, 1, n) for n in shape], indexing="ij")
        return (0.55 + 0.35 * np.cos(sum((i + 1.3) * a for i, a in enumerate(grid)) * 1.7 + rng.random())).astype(np.float32)
    return [smooth((16, 16, 16)), smooth((16, 16, 16)), smooth((16, 16)), smooth((16, 16)), smooth((32, 32)), smooth((32,))]


def _compare_with_reference_kernels(oracle, pt, w, samples, rel_tol, outlier_tol, luts=None, bvh_type=8):
    view = oracle.SceneView(pt, bvh_type=bvh_type, luts=luts)
    ours, theirs = oracle.Frame(view), _reference_frame(oracle, view)
    nb = pt.device_config().num_bounces
    totals = {}
    for s in range(samples):
        oc, rc = ours.render_sample(s), theirs.render_sample(s)
        for queue in ("trace", "shadow", "diffuse", "plastic", "dielectric", "conductor"):
            got, want = list(getattr(oc, queue)[:nb]), [int(v) for v in rc[queue][:nb]]
            assert got[0] == want[0] and all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(got, want)), (s, queue, got, want)
            totals[queue] = totals.get(queue, 0) + sum(want)
        a, b = ours.final[:, :w, :3], theirs.final[:, :w, :3]
        assert np.isfinite(b).all()
        assert np.abs(a - b).sum() / b.sum() < rel_tol, (s, np.abs(a - b).sum() / b.sum())
        assert (np.abs(a - b).max(axis=2) > 0.01 * (b.max(axis=2) + 1e-3)).mean() < outlier_tol, s
    theirs.close()
    return totals


@pytest.mark.reference_layout
def test_reference_kernels_on_sponza_textures_instances_and_plastic(grt, oracle):
    """Same pin on Sponza: 384 instances through the TLAS, 19 mip-mapped textures (ray-cone LOD, anisotropic
    bounce-0 lookups -- both sides filter with the software texture unit, the reference's NVIDIA unit being the one
    thing that cannot run here), and the variant whose odd materials are rough plastic."""
    scene, pt = make_pathtracer(grt, "sponza", 96, 54, -1, num_bounces=4)
    totals = _compare_with_reference_kernels(oracle, pt, 96, 2, 3e-4, 2e-3)
    assert totals["diffuse"] > 10000 and totals["shadow"] > 5000
    pt.close(); scene.close()

    grt.config_reset()
    scene = grt.Scene(grt.scene_path("sponza"))
    for i in range(1, scene.material_count, 2):
        if scene.material_type(i) == grt.MATERIAL_DIFFUSE:
            scene.set_material(i, grt.MATERIAL_PLASTIC, None, 0.3)
    grt.config_set(num_bounces=4)
    pt = grt.Pathtracer(scene, 96, 54, device=-1); pt.update()
    totals = _compare_with_reference_kernels(oracle, pt, 96, 2, 3e-4, 2e-3)
    assert totals["plastic"] > 5000
    pt.close(); scene.close(); grt.config_reset()


@pytest.mark.reference_layout
def test_reference_kernels_on_dielectric_conductor_and_medium(grt, oracle, tmp_path):
    """Rough dielectric with a scattering medium inside, a smooth dielectric and a rough conductor (BSDF.h:192-525,
    the medium branch of kernel_sort, Kulla-Conty energy compensation): queue sizes per material and bounce and the
    frames of the reference's kernels and the oracle coincide."""
    from test_gpu_materials_svgf import GLASS_SCENE
    (tmp_path / "glass.xml").write_text(GLASS_SCENE)
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "glass.xml"))
    pt = grt.Pathtracer(scene, 96, 64, device=-1); pt.update()
    totals = _compare_with_reference_kernels(oracle, pt, 96, 3, 2e-5, 1e-3, luts=_synthetic_luts())
    assert totals["dielectric"] > 1000 and totals["conductor"] > 300
    pt.close(); scene.close(); grt.config_reset()


@pytest.mark.parametrize("bvh_type", [2, 4])
def test_reference_binary_and_4_wide_kernels(grt, oracle, bvh_type):
    """kernel_trace_bvh2 / bvh4 and their shadow variants (BVH2.h, BVH4.h) of the reference, on the trees the host
    builds for those types, against the oracle's traversal of the same trees."""
    scene, pt = make_pathtracer(grt, "cornellbox", 64, 48, -1, bvh_type=bvh_type, num_bounces=4)
    _compare_with_reference_kernels(oracle, pt, 64, 2, 2e-5, 2e-3, bvh_type=bvh_type)
    pt.close(); scene.close(); grt.config_reset()


@pytest.mark.reference_layout
@pytest.mark.parametrize("taa", [1, 0])
def test_reference_svgf_and_taa_kernels(grt, oracle, taa):
    """SVGF (reproject, spatial variance, six a-trous iterations, finalize) and TAA of the reference (SVGF.h, TAA.h)
    over five frames with a static camera, against the oracle's restatement: filtered frames and history lengths."""
    scene, pt = make_pathtracer(grt, "cornellbox", 64, 48, -1, num_bounces=3, enable_svgf=1, enable_taa=taa)
    view = oracle.SceneView(pt)
    ours, theirs = oracle.Frame(view), _reference_frame(oracle, view)
    for f in range(5):
        if f:
            pt.update()
        vp = pt.view_projection()
        for i in range(16):
            view.scene.view_projection[i] = vp[0][i]
            view.scene.view_projection_prev[i] = vp[1][i]
        ours.render_sample(pt.sample_index); theirs.render_sample(pt.sample_index)
        a, b = ours.final[:, :64, :3], theirs.final[:, :64, :3]
        assert np.isfinite(b).all() and b.mean() > 0.01
        # The edge-stopping weights are exp(-|dz| / (sigma_z |grad z . dp| + 1e-8)): on the Cornell walls that face the
        # camera the depth gradient is zero up to rounding, so a last-bit difference in a depth (the two builds fuse
        # multiply-adds differently) switches single taps on or off. The frames still have to agree to 0.25 % overall.
        assert np.abs(a - b).sum() / b.sum() < 2.5e-3, (f, np.abs(a - b).sum() / b.sum())
        assert (np.abs(a - b).max(axis=2) > 0.02 * (b.max(axis=2) + 1e-3)).mean() < 6e-2, f   # six a-trous passes spread each switched tap
        assert np.array_equal(ours.buffers["hl"].reshape(48, -1)[:, :64], theirs.history_length()[:, :64]), f
    theirs.close(); pt.close(); scene.close(); grt.config_reset()


@pytest.mark.reference_layout
def test_reference_kulla_conty_table_kernels(grt, oracle):
    """kernel_integrate_dielectric / _conductor (100 000 samples per cell) and the two averaging kernels of the
    reference (KullaConty.h:83-240) against oracle_luts.cpp: a spread of cells of each table, and the averages of
    whole (synthetic) tables."""
    scene, pt = make_pathtracer(grt, "cornellbox", 16, 16, -1)
    view = oracle.SceneView(pt)
    theirs = _reference_frame(oracle, view)
    for entering in (True, False):
        for first in (0, 273, 1911, 4090):
            want = theirs.integrate_dielectric_cells(entering, first, 3)
            got = view.integrate_dielectric_cells(entering, first, 3)
            assert np.allclose(got, want, rtol=2e-5, atol=2e-6), (entering, first, got, want)
            assert (want >= 0).all() and (want <= 1.0001).all()
    for first in (0, 500, 1021):
        want = theirs.integrate_conductor_cells(first, 3)
        got = view.integrate_conductor_cells(first, 3)
        assert np.allclose(got, want, rtol=2e-5, atol=2e-6), (first, got, want)
    luts = _synthetic_luts()
    assert np.allclose(oracle.average_dielectric(luts[0]), theirs.average_dielectric(luts[0]), rtol=1e-6, atol=1e-7)
    assert np.allclose(oracle.average_conductor(luts[4]), theirs.average_conductor(luts[4]), rtol=1e-6, atol=1e-7)
    theirs.close(); pt.close(); scene.close()


@pytest.mark.reference_layout
@pytest.mark.parametrize("scene_name,w,h,radius", [("cornellbox", 64, 48, 0.5), ("sponza", 96, 54, 1.5)])
def test_reference_ambient_occlusion_kernels(grt, oracle, scene_name, w, h, radius):
    """The reference's AO integrator (Src/CUDA/AO.cu, verbatim, run on the CPU through oracle/_ref/libref_ao.so)
    against the oracle's restatement: the same number of occlusion rays and the same 0 / 1 / fractional image."""
    if oracle.ref_ao_lib() is None:
        pytest.skip("oracle/_ref/libref_ao.so not built (no /root/reference on this machine)")
    grt.config_reset()
    scene = grt.Scene(grt.scene_path(scene_name))
    ao = grt.AO(scene, w, h, device=-1, radius=radius); ao.update()
    view = oracle.SceneView(ao)
    ours, theirs = oracle.Frame(view), oracle.ReferenceAOFrame(view)
    for s in range(4):
        oc = ours.render_ao_sample(s, radius)
        primary, occlusion = theirs.render_ao_sample(s, radius)
        assert primary == w * h and abs(occlusion - oc.shadow[0]) <= 1 + 0.001 * occlusion, (s, occlusion, oc.shadow[0])
        a, b = ours.final[:, :w, :3], theirs.final[:, :w, :3]
        assert np.abs(a - b).sum() / max(b.sum(), 1e-6) < 2e-3, (s, np.abs(a - b).sum() / b.sum())
        assert (np.abs(a - b).max(axis=2) > 1e-3).mean() < 3e-3, s
    assert 0.05 < theirs.final[:, :w, 0].mean() < 0.99
    theirs.close(); ao.close(); scene.close()


@pytest.mark.reference_layout
def test_reference_kernels_thin_lens_hdr_sky_and_instances(grt, oracle, tmp_path):
    """More of kernel_generate / kernel_sort through the reference's own code: a thin-lens camera (aperture sampling),
    an HDR environment map with structure (sample_sky on misses at every bounce), and instanced file meshes with
    rotation + uniform scale (object-space traversal, normal transforms, the plastic BSDF)."""
    from scenes import write_thin_lens_hdr_scene
    write_thin_lens_hdr_scene(tmp_path)
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "s.xml"), sky=str(tmp_path / "sky.hdr"))
    pt = grt.Pathtracer(scene, 64, 40, device=-1); pt.update()
    assert pt.camera().aperture_radius > 0.1 and pt.sky()[1:3] == (32, 16)
    totals = _compare_with_reference_kernels(oracle, pt, 64, 3, 2e-5, 1e-3)
    assert totals["plastic"] > 200 and totals["shadow"] == 0               # lit by the sky alone
    pt.close(); scene.close(); grt.config_reset()


@pytest.mark.reference_layout
def test_reference_svgf_reprojection_with_a_moving_camera(grt, oracle):
    """SVGF temporal reprojection under camera motion (kernel_svgf_reproject: previous screen positions from the
    g-buffer, bilinear history taps, consistency tests, disocclusions): history lengths identical, frames within the
    edge-stopping noise."""
    scene, pt = make_pathtracer(grt, "cornellbox", 64, 48, -1, num_bounces=3, enable_svgf=1, enable_taa=1)
    view = oracle.SceneView(pt)
    ours, theirs = oracle.Frame(view), _reference_frame(oracle, view)
    for f in range(4):
        if f:
            scene.set_camera((0.02 * f, 1.0 + 0.01 * f, 6.8), (0.0, 0.004 * f, 0.0, 1.0)); pt.update()
            view.scene.camera = oracle.SceneView(pt).scene.camera
        vp = pt.view_projection()
        for i in range(16):
            view.scene.view_projection[i] = vp[0][i]; view.scene.view_projection_prev[i] = vp[1][i]
        ours.render_sample(pt.sample_index); theirs.render_sample(pt.sample_index)
        a, b = ours.final[:, :64, :3], theirs.final[:, :64, :3]
        assert np.abs(a - b).sum() / b.sum() < 6e-3, (f, np.abs(a - b).sum() / b.sum())
        history = theirs.history_length()[:, :64]
        assert np.array_equal(ours.buffers["hl"].reshape(48, -1)[:, :64], history), f
        if f:
            assert 0.8 * f < history.mean() <= f and (history == 0).any()      # most pixels reproject, some are disoccluded
    theirs.close(); pt.close(); scene.close(); grt.config_reset()


@pytest.mark.reference_layout
def test_oracle_matches_golden_frames_rendered_by_the_reference_kernels(grt, oracle):
    """The committed fixture tests/golden/reference_kernels_golden.npz holds frames and queue sizes produced by the
    reference's own Pathtracer.cu on the CPU (tests/golden/make_golden.py --only-reference-kernels): the oracle has to
    reproduce them, also on a machine where oracle/_ref cannot be built."""
    golden = np.load(os.path.join(os.path.dirname(GOLDEN), "reference_kernels_golden.npz"))
    # (the Sponza fixture was rendered from uncompressed textures; BC1 textures against the reference's kernels: test_reference_*)
    cases = (("cornell", "cornellbox", dict(num_bounces=5), 64, 48), ("cornell_no_nee", "cornellbox", dict(num_bounces=4, enable_next_event_estimation=0), 64, 48),
             ("sponza", "sponza", dict(num_bounces=3, enable_block_compression=0), 80, 45))
    for name, scene_name, config, w, h in cases:
        scene, pt = make_pathtracer(grt, scene_name, w, h, -1, **config)
        frame = oracle.Frame(oracle.SceneView(pt))
        queues = golden[name + "_queues"]
        for s in range(queues.shape[0]):
            oc = frame.render_sample(s)
            for k, queue in enumerate(("trace", "shadow", "diffuse")):
                got, want = list(getattr(oc, queue)[:8]), queues[s, k].tolist()
                assert all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(got, want)), (name, s, queue, got, want)
        want = golden[name + "_image"]
        got = frame.final[:, :w, :3]
        assert np.abs(got - want).sum() / want.sum() < 3e-4, name
        pt.close(); scene.close()
    grt.config_reset()


@pytest.mark.parametrize("toggles", [{}, {"enable_multiple_importance_sampling": 0}, {"enable_next_event_estimation": 0}, {"enable_mipmapping": 0}],
                         ids=["default", "no-mis", "no-nee", "no-mipmaps"])
@pytest.mark.reference_layout
def test_reference_kernels_on_a_scene_with_everything(grt, oracle, tmp_path, toggles):
    """One scene through the reference's kernels and the oracle with every feature at once: a textured rough-plastic
    floor (uv repeat, mip maps), two emitters of different power of which one is a rotated, scaled file mesh
    (light_mesh_transform_indices), a rough dielectric holding a back-scattering medium, a named conductor, and a dim sky."""
    from test_loaders import _png_bytes
    from scenes import write_scene_with_everything
    write_scene_with_everything(tmp_path, _png_bytes)
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "s.xml")); scene.set_sky_scale(0.3)
    grt.config_set(**toggles)
    pt = grt.Pathtracer(scene, 72, 48, device=-1); pt.update()
    totals = _compare_with_reference_kernels(oracle, pt, 72, 3, 1e-4, 2e-3, luts=_synthetic_luts())
    assert totals["plastic"] > 3000 and totals["dielectric"] > 800 and totals["conductor"] > 200
    assert (totals["shadow"] > 3000) == bool(toggles.get("enable_next_event_estimation", 1))
    pt.close(); scene.close(); grt.config_reset()
