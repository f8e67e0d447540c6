"""The HIP path on the MI355X against the REFERENCE'S OWN KERNELS, material by material and feature by feature, with
the restated oracle as a second witness and not as the go-between: `Src/CUDA/Pathtracer.cu` compiled verbatim for the
host (oracle/ref/ref_cuda_harness.cpp, prebuilt into oracle/_ref; it travels to the GPU box with the repo) runs the same
staged arrays the device received, one CUDA thread at a time, and must give the queue sizes of every material and bounce
and the frames the device gives. tests/test_gpu_parity.py holds the first such test (Cornell box, diffuse); here:
Sponza with its textures (block-compressed on the device, decoded per texel fetch) and the rough-plastic variant, the
rough dielectric + medium / smooth dielectric / conductor scene, a scene with every feature at once, and -- for the first
time on the device at all -- a thin-lens camera under an HDR environment map with rotated and scaled instances
(CUDA/Camera.h:20-62, CUDA/Sky.h:7-16).

Tolerances: queue sizes within 0.2 % + 2 rays per bounce (ulp-level differences of sinf / cosf / logf between glibc and
the device library flip a handful of Russian-roulette and acceptance decisions per million), bounce 0 exact; images within
the relative L1 written at each call (the reference's kernels and the HIP kernels differ in where multiply-adds are fused
outside traversal, nothing else)."""
import ctypes

import numpy as np
import pytest

from conftest import make_pathtracer

pytestmark = pytest.mark.gpu

QUEUES = ("trace", "shadow", "diffuse", "plastic", "dielectric", "conductor")


def reference_frame(oracle, view):
    if oracle.ref_lib() is None or not hasattr(oracle.ref_lib(), "ref_cuda_frame_create"):
        pytest.skip("oracle/_ref was built without the reference's device code")
    return oracle.ReferenceFrame(view)


def render_and_compare(grt, oracle, pt, w, frames, rel_tol, outlier_tol, luts=None, with_oracle=True):
    """`frames` samples on the device, each against the reference's kernels (and the oracle) fed the same arrays."""
    view = oracle.SceneView(pt, luts=luts)
    theirs = reference_frame(oracle, view)
    ours = oracle.Frame(view) if with_oracle else None
    nb = pt.device_config().num_bounces
    totals = {}
    for f in range(frames):
        if f:
            pt.update()
        pt.render()
        c = pt.counters()
        rc = theirs.render_sample(pt.sample_index)
        oc = ours.render_sample(pt.sample_index) if ours else None
        for queue in QUEUES:
            got, want = list(getattr(c, queue)[:nb]), [int(v) for v in rc[queue][:nb]]
            assert got[0] == want[0] and all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(got, want)), (f, queue, got, want)
            if oc is not None:
                mid = list(getattr(oc, queue)[:nb])
                assert all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(got, mid)), (f, queue, got, mid)
            totals[queue] = totals.get(queue, 0) + sum(got)
        got = pt.read_framebuffer()[:, :w, :3]
        for name, frame in (("reference kernels", theirs), ("oracle", ours)):
            if frame is None:
                continue
            want = frame.final[:, :w, :3]
            assert np.isfinite(got).all() and np.isfinite(want).all()
            rel = np.abs(got - want).sum() / want.sum()
            outliers = (np.abs(got - want).max(axis=2) > 0.01 * (want.max(axis=2) + 1e-3)).mean()
            assert rel < rel_tol and outliers < outlier_tol, (name, f, rel, outliers)
    theirs.close()
    return totals


@pytest.mark.reference_layout
@pytest.mark.parametrize("plastic", [False, True], ids=["diffuse", "odd-materials-plastic"])
def test_sponza_frames_equal_the_references_kernels(grt, oracle, plastic):
    """384 instances through the TLAS, 19 mip-mapped BC1 textures (ray-cone LOD, anisotropic lookups at bounce 0), NEE +
    MIS + Russian roulette; second parameter: the benchmark's variant with every odd material rough plastic."""
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("sponza"))
    if plastic:
        for i in range(1, scene.material_count, 2):
            if scene.material_type(i) == grt.MATERIAL_DIFFUSE:
                scene.set_material(i, grt.MATERIAL_PLASTIC, None, 0.3)
    grt.config_set(num_bounces=5)
    w, h = 320, 180
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    totals = render_and_compare(grt, oracle, pt, w, 2, 3e-4, 2e-3)
    assert totals["diffuse"] > 30000 and totals["shadow"] > 50000 and (totals["plastic"] > 30000) == plastic
    pt.close(); scene.close(); grt.config_reset()


@pytest.mark.reference_layout
def test_glass_medium_and_conductor_frames_equal_the_references_kernels(grt, oracle, tmp_path):
    """Rough dielectric holding a scattering medium, a smooth dielectric, a rough conductor (BSDF.h:192-525, the medium
    branch of kernel_sort, Kulla-Conty energy compensation). The reference's kernels read the tables the DEVICE integrated
    (rt_read_luts): 100 000 samples per cell take the CPU minutes, and the table kernels have their own test."""
    from test_gpu_materials_svgf import GLASS_SCENE
    (tmp_path / "glass.xml").write_text(GLASS_SCENE)
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "glass.xml"))
    w, h = 192, 128
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    pt.render()                                           # (the tables are integrated on first use)
    luts = grt.read_luts(pt.ctx)
    pt.close()
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    totals = render_and_compare(grt, oracle, pt, w, 3, 1e-4, 2e-3, luts=luts)
    assert totals["dielectric"] > 4000 and totals["conductor"] > 1000
    pt.close(); scene.close(); grt.config_reset()


@pytest.mark.reference_layout
def test_scene_with_everything_equals_the_references_kernels(grt, oracle, tmp_path):
    """A textured rough-plastic floor with uv repeat, two emitters of different power (one a rotated, scaled file mesh:
    light_mesh_transform_indices), a rough dielectric with a back-scattering medium inside, a named conductor, a dim sky."""
    from test_loaders import _png_bytes
    from scenes import write_scene_with_everything
    grt.config_reset()
    scene = grt.Scene(write_scene_with_everything(tmp_path, _png_bytes)); scene.set_sky_scale(0.3)
    w, h = 216, 144
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    pt.render(); luts = grt.read_luts(pt.ctx); pt.close()
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    totals = render_and_compare(grt, oracle, pt, w, 3, 2e-4, 3e-3, luts=luts)
    assert totals["plastic"] > 20000 and totals["dielectric"] > 5000 and totals["conductor"] > 1500 and totals["shadow"] > 20000
    pt.close(); scene.close(); grt.config_reset()


@pytest.mark.reference_layout
def test_thin_lens_camera_hdr_sky_and_instances_on_the_device(grt, oracle, tmp_path):
    """kernel_generate with a thin-lens camera (aperture samples, focal plane), sample_sky on an HDR environment map at
    every miss, instanced file meshes with rotation + uniform scale: the device's primary rays against the oracle's
    (origins on the lens to one ulp of the camera position, directions to 5e-7), then frames against the reference's kernels and the oracle."""
    from scenes import write_thin_lens_hdr_scene
    xml, sky = write_thin_lens_hdr_scene(tmp_path)
    grt.config_reset()
    scene = grt.Scene(xml, sky=sky)
    w, h = 256, 160
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    assert pt.camera().aperture_radius > 0.1 and pt.sky()[1:3] == (32, 16)
    view = oracle.SceneView(pt)
    o, d, px = grt.generate_rays(pt.ctx, 0, 0, w * h)
    oo, od, opx = view.generate(0, 0, w * h)
    assert np.array_equal(px, opx)
    assert np.unique(np.round(o, 4), axis=1).shape[1] > 1000               # rays start all over the lens ...
    # sample_disk goes through sinf / cosf (a few ulp between glibc and the device library); an origin is camera position + lens
    # offset, so one ulp of the largest coordinate (7.0 -> 4.8e-7) is the resolution of the comparison
    assert np.allclose(o, oo, atol=1e-6, rtol=0) and np.allclose(d, od, atol=5e-7, rtol=0)
    assert (o == oo).mean() > 0.9
    totals = render_and_compare(grt, oracle, pt, w, 3, 1e-4, 2e-3)
    assert totals["plastic"] > 2000 and totals["shadow"] == 0              # lit by the sky alone
    pt.close(); scene.close(); grt.config_reset()


def _one_hop(grt, oracle, scene, w, h, frames, rel_tol, outlier_tol, l2_tol, label, luts=None):
    """The device renders the DEFAULT layout of `scene` (static instances flattened into one world-space tree: aliases, re-indexed
    light tables, decoded nodes, the engine without TLAS code when nothing is left outside); the reference's own kernels render
    the REFERENCE'S layout of the same scene, staged by a second integrator that has no device (merge_static 0). One comparison,
    no restated oracle and no second device run in between."""
    from test_gpu_full_size import pixel_l2, pixel_breakdown, record
    grt.config_set(merge_static=1)
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    assert pt.static_geometry_members >= 2
    if luts == "device":
        pt.render(); luts = grt.read_luts(pt.ctx); pt.close()
        pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    grt.config_set(merge_static=0)
    staged = grt.Pathtracer(scene, w, h, device=-1); staged.update()
    assert staged.static_geometry_members == 0
    theirs = reference_frame(oracle, oracle.SceneView(staged, luts=luts))
    nb = pt.device_config().num_bounces
    totals = {}
    for f in range(frames):
        if f:
            pt.update()
        pt.render()
        c = pt.counters()
        rc = theirs.render_sample(pt.sample_index)
        for queue in QUEUES:
            got, want = list(getattr(c, queue)[:nb]), [int(v) for v in rc[queue][:nb]]
            assert got[0] == want[0] and all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(got, want)), (label, f, queue, got, want)
            totals[queue] = totals.get(queue, 0) + sum(got)
        got, want = pt.read_framebuffer()[:, :w, :3], theirs.final[:, :w, :3]
        assert np.isfinite(got).all() and np.isfinite(want).all()
        rel = np.abs(got - want).sum() / want.sum()
        outliers = (np.abs(got - want).max(axis=2) > 0.01 * (want.max(axis=2) + 1e-3)).mean()
        l2 = pixel_l2(got, want)
        record("%s frame %d (one hop)" % (label, f), rel_l1=rel, outlier_fraction=outliers, pixel_l2=l2)
        pixel_breakdown(got, want, "%s frame %d (one hop: device, default layout, vs Pathtracer.cu on the CPU, reference layout)" % (label, f))
        assert rel < rel_tol and outliers < outlier_tol and l2 < l2_tol, (label, f, rel, outliers, l2)
    theirs.close(); staged.close(); pt.close()
    return totals


def test_benchmark_scene_in_the_default_layout_meets_the_references_kernels_in_one_hop(grt, oracle):
    """The benchmarked configuration -- Sponza, odd materials rough plastic, BC1 textures, NEE + MIS + RR -- at 640x360, rendered
    by the device in the layout bench.py times (all 384 instances flattened, kernel_trace_stream_bvh8_flat_decoded) against
    Src/CUDA/Pathtracer.cu walking the reference's own TLAS + 383 BLAS."""
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("sponza"))
    for i in range(1, scene.material_count, 2):
        if scene.material_type(i) == grt.MATERIAL_DIFFUSE:
            scene.set_material(i, grt.MATERIAL_PLASTIC, None, 0.3)
    grt.config_set(num_bounces=5)
    totals = _one_hop(grt, oracle, scene, 640, 360, 2, 3e-4, 2e-3, 3e-3, "sponza plastic 640x360")
    assert totals["diffuse"] > 120000 and totals["plastic"] > 120000 and totals["shadow"] > 200000
    scene.close(); grt.config_reset()


def test_scene_with_everything_in_the_default_layout_meets_the_references_kernels_in_one_hop(grt, oracle, tmp_path):
    from test_loaders import _png_bytes
    from scenes import write_scene_with_everything
    grt.config_reset()
    scene = grt.Scene(write_scene_with_everything(tmp_path, _png_bytes)); scene.set_sky_scale(0.3)
    totals = _one_hop(grt, oracle, scene, 216, 144, 3, 2e-4, 3e-3, 3e-3, "scene with everything", luts="device")
    assert totals["plastic"] > 20000 and totals["dielectric"] > 5000 and totals["conductor"] > 1500 and totals["shadow"] > 20000
    scene.close(); grt.config_reset()
