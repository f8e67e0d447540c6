"""GPU parity for the remaining rows of the scope table: dielectric + conductor BSDFs with the
Kulla-Conty LUTs, homogeneous media (BASELINE config #5 stand-in), and SVGF + TAA (config #3)."""
import numpy as np
import pytest

from conftest import make_pathtracer

pytestmark = pytest.mark.gpu

GLASS_SCENE = """<?xml version="1.0"?>
<scene version="0.5.0">
  <integrator type="path"><integer name="maxDepth" value="8"/></integrator>
  <sensor type="perspective"><float name="fov" value="40"/>
    <transform name="toWorld"><lookat origin="0, 1, 4.5" target="0, 0.9, 0" up="0, 1, 0"/></transform></sensor>
  <bsdf type="diffuse" id="white"><rgb name="reflectance" value="0.7, 0.7, 0.7"/></bsdf>
  <bsdf type="diffuse" id="red"><rgb name="reflectance" value="0.6, 0.1, 0.1"/></bsdf>
  <bsdf type="roughconductor" id="gold"><rgb name="eta" value="1.45, 0.43, 0.21"/><rgb name="k" value="1.95, 2.46, 3.27"/><float name="alpha" value="0.3"/></bsdf>
  <shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="-90"/><scale value="3"/></transform><ref id="white"/></shape>
  <shape type="rectangle"><transform name="toWorld"><scale value="3"/><translate z="-2" y="2"/></transform><ref id="red"/></shape>
  <shape type="cube"><transform name="toWorld"><scale value="0.4"/><translate x="-1.1" y="0.4" z="0.2"/></transform><ref id="gold"/></shape>
  <shape type="sphere"><float name="radius" value="0.6"/><transform name="toWorld"><translate x="0.4" y="0.6" z="0.3"/></transform>
    <bsdf type="roughdielectric"><string name="intIOR" value="water"/><float name="alpha" value="0.1"/></bsdf>
    <medium type="homogeneous" name="interior"><rgb name="sigmaA" value="0.3, 0.1, 0.05"/><rgb name="sigmaS" value="0.8, 0.9, 1.0"/>
      <phase type="hg"><float name="g" value="0.2"/></phase></medium></shape>
  <shape type="sphere"><float name="radius" value="0.3"/><transform name="toWorld"><translate x="1.3" y="0.3" z="0.9"/></transform>
    <bsdf type="dielectric"><float name="intIOR" value="1.5"/></bsdf></shape>
  <shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><scale value="0.7"/><translate y="2.8"/></transform>
    <emitter type="area"><rgb name="radiance" value="14, 13, 11"/></emitter></shape>
</scene>"""


def compare(grt, oracle, pt, frame, w, h, rel_tol, outlier_tol):
    pt.render()
    c = pt.counters()
    oc = frame.render_sample(pt.sample_index)
    nb = pt.device_config().num_bounces
    for name in ("trace", "shadow", "diffuse", "dielectric", "conductor"):
        got_q, want_q = list(getattr(c, name)[:nb]), list(getattr(oc, name)[:nb])
        assert all(abs(a - b) <= 3 + 0.004 * b for a, b in zip(got_q, want_q)), (name, got_q, want_q)
    got, want = pt.read_framebuffer()[:, :w, :3], frame.final[:, :w, :3]
    assert np.isfinite(got).all()
    rel = np.abs(got - want).sum() / want.sum()
    outliers = (np.abs(got - want).max(axis=2) > 0.02 * (want.max(axis=2) + 1e-3)).mean()
    assert rel < rel_tol and outliers < outlier_tol, (rel, outliers)
    return c


def test_kulla_conty_luts_match_the_oracle(grt, oracle):
    scene, pt = make_pathtracer(grt, "cornellbox", 64, 64, 0)
    luts = grt.read_luts(pt.ctx)   # kernel_integrate_* / kernel_average_* ran on the device
    view = oracle.SceneView(pt)
    # a spread of LUT cells, same 100 000 samples each: only transcendental ulps may differ
    for entering, lut in ((True, luts[0]), (False, luts[1])):
        for first in (0, 1000, 2345, 4090):
            want = view.integrate_dielectric_cells(entering, first, 6)
            assert np.allclose(lut[first:first + 6], want, atol=2e-4), (entering, first)
    want = view.integrate_conductor_cells(500, 8)
    assert np.allclose(luts[4][500:508], want, atol=2e-4)
    assert np.allclose(luts[2], oracle.average_dielectric(luts[0]), atol=1e-6)
    assert np.allclose(luts[3], oracle.average_dielectric(luts[1]), atol=1e-6)
    assert np.allclose(luts[5], oracle.average_conductor(luts[4]), atol=1e-6)
    # physical sanity: directional albedo in [0,1], smooth conductors lose little energy
    assert 0.0 <= luts[4].min() and luts[4].max() <= 1.0 + 1e-4 and luts[4].reshape(32, 32)[-1, 0] > 0.9
    pt.close(); scene.close()


def test_glass_conductor_medium_scene_matches_oracle(grt, oracle, tmp_path):
    (tmp_path / "glass.xml").write_text(GLASS_SCENE)
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "glass.xml"))
    pt = grt.Pathtracer(scene, 192, 128, device=0); pt.update()
    luts = grt.read_luts(pt.ctx)
    view = oracle.SceneView(pt, luts=luts)
    frame = oracle.Frame(view)
    for f in range(3):
        if f:
            pt.update()
        c = compare(grt, oracle, pt, frame, 192, 128, 5e-4, 5e-3)
    assert sum(c.dielectric[:8]) > 0 and sum(c.conductor[:8]) > 0
    pt.close(); scene.close()


@pytest.mark.parametrize("taa", [1, 0])
def test_svgf_taa_pipeline_matches_oracle(grt, oracle, taa):
    """Four frames with a static camera: temporal history, spatial variance (first 4 frames),
    6 a-trous iterations, finalize, TAA. Images and history lengths must agree."""
    scene, pt = make_pathtracer(grt, "cornellbox", 128, 96, 0, num_bounces=4, enable_svgf=1, enable_taa=taa)
    view = oracle.SceneView(pt)
    # the SVGF matrices of the frame (uploaded by Pathtracer::update)
    import ctypes
    frame = oracle.Frame(view)
    for f in range(5):
        if f:
            pt.update()
        vp = pt.view_projection()
        for i in range(16):
            view.scene.view_projection[i] = vp[0][i]
            view.scene.view_projection_prev[i] = vp[1][i]
        pt.render()
        frame.render_sample(pt.sample_index)
        got, want = pt.read_framebuffer()[:, :128, :3], frame.final[:, :128, :3]
        assert np.isfinite(got).all()
        rel = np.abs(got - want).sum() / want.sum()
        outliers = (np.abs(got - want).max(axis=2) > 0.02 * (want.max(axis=2) + 1e-3)).mean()
        assert rel < 1e-3 and outliers < 1e-2, (f, rel, outliers)
    # denoised image is smoother than the raw one-sample radiance
    assert got.std() > 0
    pt.close(); scene.close()


def test_svgf_frames_pipeline_without_changing_the_image(grt):
    """SVGF frames in flight: six frames submitted back to back give bit-identical filtered images under the slot
    scheduler with 1 and with 3 frames in flight (per-slot g-buffers, the filter stage ordered by events), in the merged
    wavefront (per-sample-slot g-buffers, the filter stage of a frame when it has passed its last bounce), and when the
    scheduler changes in the middle of the sequence (the g-buffers of the last frame are handed over) -- also where rays
    miss all geometry (those pixels keep the g-buffer of the last frame that hit)."""
    images = {}
    for label, schedule in (("slots, 1 in flight", ["slots"] * 6), ("slots, 3 in flight", ["slots"] * 6), ("merged", ["merged"] * 6),
                            ("merged then slots", ["merged"] * 3 + ["slots"] * 3), ("slots then merged", ["slots"] * 2 + ["merged"] * 4)):
        scene, pt = make_pathtracer(grt, "cornellbox", 200, 150, 0, num_bounces=4, enable_svgf=1, enable_taa=1)
        grt.set_samples_in_flight(pt.ctx, 1 if "1 in flight" in label else 3)
        for f, scheduler in enumerate(schedule):
            if f:
                pt.update()
            if f == 0 or scheduler != schedule[f - 1]:
                grt.set_scheduler(pt.ctx, scheduler)   # (completes what is in flight)
            pt.render()
        images[label] = pt.read_framebuffer().copy()
        pt.close(); scene.close()
    first = images["slots, 1 in flight"]
    assert np.isfinite(first).all() and first[..., :3].max() > 0.0
    for label, image in images.items():
        assert np.array_equal(image, first), label


@pytest.mark.parametrize("size", [(200, 150), (333, 77)])
def test_svgf_lds_tiles_do_not_change_a_frame(grt, size):
    """rt_set_svgf_tiles: the a-trous passes with a workgroup's taps staged in LDS (rows `step` apart, the default) against the
    passes that load every tap from the images -- six frames with history, spatial variance, six iterations (steps 1 .. 32,
    i.e. every instantiation of the tiled kernel) and TAA, bit-identical, also at a size that is no multiple of anything
    (partial tiles in x, rows beyond the image in the last block of every residue class)."""
    images = {}
    for tiles in (False, True):
        scene, pt = make_pathtracer(grt, "cornellbox", size[0], size[1], 0, num_bounces=4, enable_svgf=1, enable_taa=1, svgf_lds_tiles=int(tiles))
        for f in range(6):
            if f:
                pt.update()
            pt.render()
        images[tiles] = pt.read_framebuffer().copy()
        pt.close(); scene.close()
    assert np.isfinite(images[True]).all() and images[True][..., :3].max() > 0.0
    assert np.array_equal(images[True], images[False])


@pytest.mark.parametrize("bsdf,lo,hi", [
    ('<bsdf type="roughconductor"><rgb name="eta" value="0.2, 0.2, 0.2"/><rgb name="k" value="8, 8, 8"/><float name="alpha" value="0.4"/></bsdf>', 0.93, 1.01),
    ('<bsdf type="roughdielectric"><float name="intIOR" value="1.5"/><float name="alpha" value="0.3"/></bsdf>', 0.95, 1.03)])
def test_white_furnace_conductor_and_dielectric(grt, tmp_path, bsdf, lo, hi):
    """Energy conservation of the Kulla-Conty compensated microfacet BSDFs (BSDF.h:192-525 with the
    LUTs integrated on the device): a nearly lossless rough conductor and a non-absorbing rough
    dielectric under the constant white sky return (almost) the sky. Rendered on the GPU -- its parity
    with the oracle is established by the tests above."""
    from test_oracle import FURNACE_XML
    (tmp_path / "f.xml").write_text(FURNACE_XML % {"bsdf": bsdf})
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "f.xml"))
    grt.config_set(enable_russian_roulette=0)
    pt = grt.Pathtracer(scene, 96, 96, device=0); pt.update()
    pt.render_samples(1)
    pt.update(); pt.render_samples(16); pt.update(); pt.render_samples(16)
    img = pt.read_framebuffer()[:, :96, :3]
    assert lo <= img.mean() <= hi, img.mean()
    assert np.isfinite(img).all()
    pt.close(); scene.close()


def test_svgf_frames_under_the_tile_split_equal_the_single_context_frames(grt):
    """BASELINE config 3 on N GPUs: every rank path-traces its tiles, the per-frame AOVs and g-buffers are
    exchanged, every rank filters the whole frame. Two contexts on this one GPU play two ranks (the all-gather is a
    concatenation of their packed tiles); five frames with a moving camera are bit-identical, on both ranks, to one
    context rendering whole frames -- temporal histories, disocclusions and TAA included."""
    import ctypes
    import importlib
    import torch
    parallel = importlib.import_module("gpu_raytracer_amd.parallel")
    W, H, world = 256, 144, 2
    lib = grt.device_lib()
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]

    def camera_of(frame):
        return (0.05 * frame, 1.0 + 0.02 * frame, 6.8 - 0.03 * frame), (0.0, 0.004 * frame, 0.0, 1.0)

    scene, pt = make_pathtracer(grt, "cornellbox", W, H, 0, num_bounces=4, enable_svgf=1, enable_taa=1)
    want = []
    for f in range(5):
        if f:
            scene.set_camera(*camera_of(f)); pt.update()
        pt.render()
        want.append(pt.read_framebuffer().copy())
    pt.close(); scene.close()

    ranks = []
    for rank in range(world):
        scene, pt = make_pathtracer(grt, "cornellbox", W, H, 0, num_bounces=4, enable_svgf=1, enable_taa=1)
        split = parallel.SvgfTileSplit(rank, world, W, H)
        assert lib.rt_set_pixel_tiles(pt.ctx, split.tile_pixels, rank, world) == 0
        ranks.append((scene, pt, split))
    packed_by_rank = {}
    for f in range(5):
        # each "rank" renders and packs; the stand-in collective hands every rank the concatenation once all have packed
        for scene, pt, split in ranks:
            if f:
                scene.set_camera(*camera_of(f)); pt.update()
        sample_index = ranks[0][1].sample_index
        # pass 1: render + pack on every rank, pass 2: scatter the gathered tiles and filter
        for rank, (scene, pt, split) in enumerate(ranks):
            lib.rt_render_sample_unfiltered.argtypes = [ctypes.c_void_p, ctypes.c_int]
            lib.rt_pack_svgf_inputs.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4
            packed = torch.zeros((split.local_pixels, split.FLOATS_PER_PIXEL), device="cuda")
            torch.cuda.synchronize()
            assert lib.rt_render_sample_unfiltered(pt.ctx, sample_index) == 0, lib.rt_last_error(pt.ctx)
            assert lib.rt_pack_svgf_inputs(pt.ctx, packed.data_ptr(), split.tile_pixels, rank, world, split.tiles_per_rank) == 0
            assert lib.rt_synchronize(pt.ctx) == 0
            packed_by_rank[rank] = packed
        gathered = torch.cat([packed_by_rank[r] for r in range(world)])
        torch.cuda.synchronize()
        for rank, (scene, pt, split) in enumerate(ranks):
            lib.rt_unpack_svgf_inputs.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3
            lib.rt_filter_frame.argtypes = [ctypes.c_void_p, ctypes.c_int]
            assert lib.rt_unpack_svgf_inputs(pt.ctx, gathered.data_ptr(), split.tile_pixels, world, split.tiles_per_rank) == 0
            assert lib.rt_filter_frame(pt.ctx, sample_index) == 0
            assert np.array_equal(pt.read_framebuffer(), want[f]), (f, rank)
    for scene, pt, split in ranks:
        pt.close(); scene.close()
    # and through the helper the frame loop of a rank uses (one rank: the gather is a copy)
    scene, pt = make_pathtracer(grt, "cornellbox", W, H, 0, num_bounces=4, enable_svgf=1, enable_taa=1)
    split = parallel.SvgfTileSplit(0, 1, W, H)
    assert lib.rt_set_pixel_tiles(pt.ctx, split.tile_pixels, 0, 1) == 0
    for f in range(5):
        if f:
            scene.set_camera(*camera_of(f)); pt.update()
        split.render_frame(grt, pt.ctx, pt.sample_index)
        assert np.array_equal(pt.read_framebuffer(), want[f]), f
    pt.close(); scene.close()
