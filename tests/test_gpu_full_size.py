"""Parity at the sizes that are benchmarked: BASELINE.json's configurations at 1920x1080 against the CPU oracle.

The other GPU tests compare small frames; these run what bench.py and tools/config_suite.py time -- the same
scene builders, frame size, bounce count, sample batching and submissions in flight -- and compare the result with
the oracle rendering the same samples one at a time (a 1080p oracle sample is a few seconds on the GPU box's cores).
Collected first (file name order), so that a regression of the benchmarked path is the first thing `pytest -x` shows.

Tolerances: queue sizes per bounce within 0.2 % + 2 rays (paths whose roulette / pdf comparison sits within an ulp of
its threshold, where glibc's and the device's sinf / cosf / logf differ); frames within REL_L1_TOL relative L1 and at
most OUTLIER_FRACTION_TOL of the pixels off by more than 1 %; SVGF frames within 1e-3 / 1 % (the edge-stopping
weights amplify last-bit depth differences)."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import make_pathtracer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

W, H = 1920, 1080
REL_L1_TOL = 1e-4
OUTLIER_FRACTION_TOL = 2e-3
QUEUES = ("trace", "shadow", "diffuse", "plastic", "dielectric", "conductor")


def queues_of(counters, nb):
    return {name: np.array(list(getattr(counters, name)[:nb]), np.int64) for name in QUEUES}


def assert_queues_agree(got, want, label, rel=0.002, slack=2):
    for name in QUEUES:
        assert got[name][0] == want[name][0], (label, name, got[name][0], want[name][0])
        assert (np.abs(got[name] - want[name]) <= slack + rel * want[name]).all(), (label, name, got[name].tolist(), want[name].tolist())


REL_L2_TOL = 2e-3   # per-pixel L2 (see pixel_l2): the bound BASELINE.md section 4 / north_star ask for, stated in DESIGN.md section 2


def pixel_l2(got, want):
    """Per-pixel L2 distance of two RGB frames, as ONE number: the root of the mean (over pixels) squared Euclidean RGB distance,
    relative to the root-mean-square pixel of the expected frame. Unlike the relative L1 next to it, it is dominated by the FEW
    pixels that differ a lot (a path whose roulette / acceptance decision flipped carries a whole different sample)."""
    d2 = ((got.astype(np.float64) - want) ** 2).sum(axis=2)
    return float(np.sqrt(d2.mean()) / np.sqrt((want.astype(np.float64) ** 2).sum(axis=2).mean()))


def record(label, **numbers):
    """The numbers behind the assertions, kept for DESIGN.md (gpurun_out/ travels back from the GPU box)."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_numbers.txt"), "a") as f:
            f.write("%-44s %s\n" % (label, "  ".join("%s %.3g" % kv for kv in numbers.items())))
    except OSError:
        pass


def pixel_breakdown(got, want, label, worst=12):
    """Which pixels carry the per-pixel L2 of a comparison, written to gpurun_out/parity_pixel_breakdown.txt: how many pixels differ by more
    than 1 % / 10 % / 100 % of their expected brightness, what share of the squared distance the worst 10 / 100 / 1000 pixels hold, and the
    worst ones themselves (position, expected and rendered RGB). A path whose roulette / acceptance / tie decision flipped shows up as ONE
    pixel with a whole different sample; rounding noise shows up as many pixels with a tiny share each."""
    g, w = got.astype(np.float64), want.astype(np.float64)
    d2 = ((g - w) ** 2).sum(axis=2)
    rel = np.sqrt(d2) / (np.sqrt((w ** 2).sum(axis=2)) + 1e-3)
    order = np.argsort(d2, axis=None)[::-1]
    total = float(d2.sum())
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_pixel_breakdown.txt"), "a") as f:
            f.write("%s: %d x %d pixels, per-pixel L2 %.3g\n" % (label, got.shape[1], got.shape[0], pixel_l2(got, want)))
            f.write("   pixels off by more than 1 %% / 10 %% / 100 %% of their expected brightness: %d / %d / %d\n" % ((rel > 0.01).sum(), (rel > 0.1).sum(), (rel > 1.0).sum()))
            f.write("   share of the squared distance held by the worst 1 / 10 / 100 / 1000 pixels: %s\n" % " / ".join("%.3f" % (float(d2.flat[order[:n]].sum()) / max(total, 1e-300)) for n in (1, 10, 100, 1000)))
            f.write("   per-pixel L2 without the worst 10 / 100 pixels: %s\n" % " / ".join("%.3g" % (np.sqrt((total - float(d2.flat[order[:n]].sum())) / d2.size) / np.sqrt((w ** 2).sum(axis=2).mean())) for n in (10, 100)))
            for k in order[:worst]:
                y, x = divmod(int(k), got.shape[1])
                f.write("   (%4d, %4d) expected %s rendered %s\n" % (x, y, np.array2string(w[y, x], precision=4), np.array2string(g[y, x], precision=4)))
    except OSError:
        pass


def assert_frames_agree(got, want, label, rel_tol=REL_L1_TOL, outlier_tol=OUTLIER_FRACTION_TOL, outlier_step=0.01, l2_tol=REL_L2_TOL):
    assert np.isfinite(got).all(), label
    rel = np.abs(got - want).sum() / want.sum()
    outliers = (np.abs(got - want).max(axis=2) > outlier_step * (want.max(axis=2) + 1e-3)).mean()
    l2 = pixel_l2(got, want)
    record(label, rel_l1=rel, outlier_fraction=outliers, pixel_l2=l2, worst_pixel=float(np.abs(got - want).max()))
    assert rel < rel_tol and outliers < outlier_tol and l2 < l2_tol, (label, rel, outliers, l2)


def bench_plan(steps, spp):
    """[(first sample, count)] as bench.py's submissions(steps) makes them with its default --batch."""
    out, k = [], 0
    while k < steps:
        first = k % spp
        count = min(spp - first, steps - k)
        k += count
        out.append((first, count))
    return out


def test_benchmarked_sponza_frame_matches_the_oracle(grt, oracle):
    """BASELINE config 2 exactly as bench.py submits it: Sponza with the plastic variant, 1920x1080, 10 bounces,
    the 4 samples of a frame as ONE submission (rt_render_samples(0, 4), virtual pixel indices, 8 M-ray launches),
    three such frames in flight. The oracle renders samples 0..3 one after the other."""
    import bench
    scene = bench.build_scene(grt)
    pt = grt.Pathtracer(scene, W, H, device=0); pt.update()
    nb = pt.device_config().num_bounces
    assert nb == bench.NUM_BOUNCES == 10
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    grt.set_samples_in_flight(pt.ctx, 3)
    for _ in range(3):   # three 4-spp frames back to back: each restarts the accumulation at sample 0
        assert lib.rt_render_samples(pt.ctx, 0, bench.SPP) == 0, lib.rt_last_error(pt.ctx)
    got_queues = queues_of(pt.counters(), nb)     # the last submission: its 4 samples summed per bounce
    got = pt.read_framebuffer()[:, :W, :3].copy()

    # The oracle renders the REFERENCE'S LAYOUT of the same scene -- one CWBVH per mesh under the TLAS, 384 instance entries,
    # object-space rays for the two transformed instances -- staged by a second integrator without a device (merge_static 0).
    # The device frame comes from the DEFAULT layout (all 384 instances flattened into one world-space tree with spatial
    # splits, the engine without TLAS code, decoded nodes): what is benchmarked meets what the reference defines in one
    # comparison, exact ties between coplanar triangles and world-space copies included.
    assert pt.static_geometry_whole_scene and pt.static_geometry_members == 384
    grt.config_set(merge_static=0)
    staged = grt.Pathtracer(scene, W, H, device=-1); staged.update()
    assert staged.static_geometry_members == 0
    frame = oracle.Frame(oracle.SceneView(staged))
    want_queues = {name: np.zeros(nb, np.int64) for name in QUEUES}
    for s in range(bench.SPP):
        oc = queues_of(frame.render_sample(s), nb)
        for name in QUEUES:
            want_queues[name] += oc[name]
    assert want_queues["trace"][0] == bench.SPP * W * H and want_queues["plastic"].sum() > 0
    assert_queues_agree(got_queues, want_queues, "bench frame")
    assert_frames_agree(got, frame.final[:, :W, :3], "bench frame (default layout) vs oracle (reference layout)")
    pixel_breakdown(got, frame.final[:, :W, :3], "bench frame (default layout) vs oracle (reference layout)")

    # ... and EXACTLY the submission pattern bench.py times with the driver's arguments (--steps 20): five 4-sample frames declared as ONE
    # burst (rt_set_frame_pipelining + rt_set_stream_batch: they enter the merged wavefront together, every traversal launch carries one
    # bounce of all five, 41 M primary rays in the first), the default kernel (kernel_trace_stream_bvh8_flat). Every frame restarts the
    # accumulation at sample 0, so the image read at the end is the fifth frame's 4 spp: the same paths as above, bit for bit.
    plan = bench_plan(20, bench.SPP)
    assert len(plan) == 5
    grt.set_frame_pipelining(pt.ctx, True)
    grt.set_stream_batch(pt.ctx, sum(count for _, count in plan) * W * H)
    for first, count in plan:
        assert lib.rt_render_samples(pt.ctx, first, count) == 0, lib.rt_last_error(pt.ctx)
    lib.rt_synchronize.argtypes = [ctypes.c_void_p]
    assert lib.rt_synchronize(pt.ctx) == 0
    burst_queues = queues_of(pt.counters(), nb)
    burst = pt.read_framebuffer()[:, :W, :3].copy()
    assert_queues_agree(burst_queues, want_queues, "bench burst")
    assert_frames_agree(burst, frame.final[:, :W, :3], "bench burst of 5 frames (default layout) vs oracle (reference layout)")
    assert np.array_equal(burst, got) and all((burst_queues[name] == got_queues[name]).all() for name in QUEUES)   # the schedule does not touch a path
    staged.close(); pt.close(); scene.close()
    grt.config_reset()


def test_sponza_svgf_taa_with_a_moving_camera_at_full_size(grt, oracle):
    """BASELINE config 3: Sponza 1920x1080 with SVGF (6 a-trous iterations) + TAA, one sample per filtered frame,
    five frames while the camera translates and turns (temporal reprojection from the previous frame's g-buffers,
    disocclusions at the columns, history lengths). Frames are submitted with 3 in flight, as config_suite times them;
    the oracle filters the same five frames."""
    import bench
    scene = bench.build_scene(grt)
    grt.config_set(enable_svgf=1, enable_taa=1, num_atrous_iterations=6)
    pt = grt.Pathtracer(scene, W, H, device=0); pt.update()
    grt.set_samples_in_flight(pt.ctx, 3)
    view = oracle.SceneView(pt)
    frame = oracle.Frame(view)
    for f in range(5):
        if f:
            position, rotation, _ = scene.get_camera()
            turned = np.array([rotation[0], rotation[1], rotation[2], rotation[3] - 0.004]); turned /= np.linalg.norm(turned)
            scene.set_camera((position[0] + 0.05, position[1] + 0.02, position[2] + 0.04), tuple(float(v) for v in turned))
            pt.update()
            view.scene.camera = oracle.SceneView(pt).scene.camera
        vp = pt.view_projection()
        for i in range(16):
            view.scene.view_projection[i] = vp[0][i]; view.scene.view_projection_prev[i] = vp[1][i]
        pt.render()
        frame.render_sample(pt.sample_index)
        got, want = pt.read_framebuffer()[:, :W, :3], frame.final[:, :W, :3]
        assert_frames_agree(got, want, "svgf frame %d" % f, rel_tol=1e-3, outlier_tol=1e-2, outlier_step=0.02)
    history = frame.buffers["hl"].reshape(H, -1)[:, :W]
    assert 3.0 < history.mean() <= 4.0 and (history == 0).any()   # most pixels reprojected four times, some disoccluded
    pt.close(); scene.close()


def test_config_4_and_5_stand_ins_match_the_oracle_at_full_size(grt, oracle, tmp_path):
    """The scenes tools/config_suite.py times for BASELINE configs 4 and 5 (SURVEY.md 8d stand-ins), at 1920x1080 and at the sample
    counts BASELINE.json states -- 4 spp and 16 spp: 441 rotated / scaled instances of a 102 400-triangle mesh (TLAS / BLAS with
    non-identity transforms, diffuse + plastic), and the rough-dielectric + medium + conductor scene."""
    import config_suite
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    # config 4
    grt.config_reset()
    scene = grt.Scene(config_suite.instancing_scene(str(tmp_path / "instancing")))
    grt.config_set(num_bounces=10)
    pt = grt.Pathtracer(scene, W, H, device=0); pt.update()
    assert lib.rt_render_samples(pt.ctx, 0, 4) == 0, lib.rt_last_error(pt.ctx)   # config 4: 4 spp, one submission
    got_queues = queues_of(pt.counters(), 10)
    got = pt.read_framebuffer()[:, :W, :3].copy()
    frame = oracle.Frame(oracle.SceneView(pt))
    want_queues = {name: np.zeros(10, np.int64) for name in QUEUES}
    for s in range(4):
        oc = queues_of(frame.render_sample(s), 10)
        for name in QUEUES:
            want_queues[name] += oc[name]
    assert want_queues["plastic"].sum() > 0 and want_queues["diffuse"].sum() > 0
    assert_queues_agree(got_queues, want_queues, "config 4")
    assert_frames_agree(got, frame.final[:, :W, :3], "config 4 (4 spp)")
    pixel_breakdown(got, frame.final[:, :W, :3], "config 4 (4 spp)")
    pt.close(); scene.close()
    # config 5
    grt.config_reset()
    scene = grt.Scene(config_suite.glass_scene(str(tmp_path / "glass")))
    pt = grt.Pathtracer(scene, W, H, device=0); pt.update()
    nb = pt.device_config().num_bounces
    luts = grt.read_luts(pt.ctx)
    got_queues = {name: np.zeros(nb, np.int64) for name in QUEUES}
    for first in range(0, 16, 4):   # config 5: 16 spp as four 4-sample submissions into the merged wavefront (counters are per submission)
        assert lib.rt_render_samples(pt.ctx, first, 4) == 0, lib.rt_last_error(pt.ctx)
        c = queues_of(pt.counters(), nb)
        for name in QUEUES:
            got_queues[name] += c[name]
    got = pt.read_framebuffer()[:, :W, :3].copy()
    frame = oracle.Frame(oracle.SceneView(pt, luts=luts))
    want_queues = {name: np.zeros(nb, np.int64) for name in QUEUES}
    for s in range(16):
        oc = queues_of(frame.render_sample(s), nb)
        for name in QUEUES:
            want_queues[name] += oc[name]
    assert want_queues["dielectric"].sum() > 0 and want_queues["conductor"].sum() > 0
    assert_queues_agree(got_queues, want_queues, "config 5", rel=0.004, slack=3)
    assert_frames_agree(got, frame.final[:, :W, :3], "config 5 (16 spp)", rel_tol=5e-4, outlier_tol=5e-3, outlier_step=0.02)
    pt.close(); scene.close()
    grt.config_reset()
