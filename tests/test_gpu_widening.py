"""GPU tests of the rows SURVEY.md 8(f) marks "next" (command line + exporters, texture block compression).
They live in their own file, collected after test_gpu_parity.py, so that with `pytest -x` a failure in a widening
row can never hide the hot-path parity tests."""
import os
import sys

import ctypes
import numpy as np
import pytest

from conftest import make_pathtracer

pytestmark = pytest.mark.gpu


def half_ties_away(values):
    """float32 -> half -> float32 the way tinyexr's float_to_half_full (and host/Exporters.cpp) round: the first
    dropped mantissa bit alone decides, i.e. ties go away from zero (numpy's astype(float16) rounds them to even)."""
    bits = np.ascontiguousarray(values, dtype=np.float32).view(np.uint32).astype(np.int64)
    sign = (bits >> 16) & 0x8000
    biased = (bits >> 23) & 0xff
    mantissa = bits & 0x7fffff
    exponent = biased - 127 + 15
    normal = ((exponent << 10) | (mantissa >> 13)) + ((mantissa >> 12) & 1)
    shift = np.clip(14 - exponent, 0, 25)
    full = mantissa | 0x800000
    subnormal = (full >> shift) + ((full >> np.clip(shift - 1, 0, 25)) & 1)
    half = np.where(exponent >= 31, 0x7c00, np.where(exponent <= 0, np.where(14 - exponent > 24, 0, subnormal), normal))
    half = np.where(biased == 0, 0, half)
    half = np.where(biased == 0xff, 0x7c00 | np.where(mantissa != 0, 0x200, 0), half)
    return (sign | half).astype(np.uint16).view(np.float16).astype(np.float32)


def test_command_line_render_and_screenshots_match_the_library(grt, tmp_path):
    """host/pathtracer (Args.cpp + the headless part of Main.cpp:75-150) renders what the library
    renders for the same options, and Integrator::save_image writes the frame the exporters' way."""
    import subprocess
    from test_loaders import CLI, _parse_exr
    scene_file = grt.scene_path("cornellbox")
    out = tmp_path / "cli.exr"
    # -W / -H / -b are given, but what the scene file says (<film> size, maxDepth) is applied later and
    # wins, as in the reference (MitsubaLoader.cpp:611-613, Main.cpp:109)
    r = subprocess.run([CLI, "-s", scene_file, "-W", "96", "-H", "64", "-N", "5", "-b", "2", "--bvh", "bvh8", "-o", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Wrote" in r.stdout
    _, cli = _parse_exr(out)

    grt.config_reset()
    scene = grt.Scene(scene_file)
    w, h = int(grt.config_get("initial_width")), int(grt.config_get("initial_height"))
    assert (w, h) == (1024, 1024) and cli["R"].shape == (h, w) and grt.config_get("num_bounces") > 2
    pt = grt.Pathtracer(scene, w, h, device=0)
    pt.update()
    while True:
        pt.render()
        if pt.sample_index >= 5:
            break
        pt.update()
    assert pt.sample_index == 5
    pt.save_image(tmp_path / "lib.exr"); pt.save_image(tmp_path / "lib.ppm")
    _, lib = _parse_exr(tmp_path / "lib.exr")
    frame = pt.read_framebuffer()[:, :w, :3][::-1]
    for k, name in enumerate("RGB"):
        assert lib[name].shape == (h, w) and np.array_equal(lib[name], half_ties_away(frame[:, :, k]))
        assert np.allclose(cli[name], lib[name], rtol=2e-3, atol=1e-4), name
    raw = open(tmp_path / "lib.ppm", "rb").read()
    header = b"P6\n %d\n %d\n 255\n" % (w, h)
    assert raw.startswith(header) and len(raw) == len(header) + w * h * 3 and np.frombuffer(raw[len(header):], np.uint8).mean() > 20
    pt.close(); scene.close()


def test_block_compressed_textures_render_like_the_oracle(grt, oracle):
    """enable_block_compression: Sponza's power-of-two maps are BC1-quantised on the host and carry the
    reference's block-count LOD size (rt_texture_desc::lod_width / lod_height), which shifts the bias of the
    bounce > 0 texture lookups by -2. Bounce-0 albedo and a 3-bounce frame match the oracle, which is given
    the same descriptors; and the frame differs from the uncompressed one (the switch does something)."""
    frames = {}
    for compress in (1, 0):
        scene, pt = make_pathtracer(grt, "sponza", 320, 180, 0, num_bounces=3, enable_block_compression=compress)
        if compress:
            textures = pt.textures()
            sizes = [pt.texture_lod_size(i) for i in range(len(textures))]
            assert sum(1 for s in sizes if s != (0, 0)) == 19                                  # every real map (they are all powers of two)
            assert all(s == ((0, 0) if t[1] == 1 else (t[1] // 4, t[2] // 4)) for s, t in zip(sizes, textures))   # e.g. 1024^2 texels -> 256^2 blocks
            pt.aov_enable(grt.AOV_ALBEDO); pt.update()
            view = oracle.SceneView(pt); frame = oracle.Frame(view)
            pt.render(); frame.render_sample(pt.sample_index)
            got, want = pt.read_aov(grt.AOV_ALBEDO)[:, :320, :3], frame.accumulator(grt.AOV_ALBEDO)[:, :320, :3]
            assert (np.abs(got - want).max(axis=2) > 2e-3).mean() < 1e-3
            got, want = pt.read_framebuffer()[:, :320, :3], frame.final[:, :320, :3]
            assert np.abs(got - want).sum() / want.sum() < 1e-3
        else:
            pt.render()
        frames[compress] = pt.read_framebuffer()[:, :320, :3].copy()
        pt.close(); scene.close()
    assert not np.array_equal(frames[0], frames[1])


def test_textures_decoded_at_upload_render_what_the_per_fetch_decode_renders(grt):
    """rt_set_texture_expansion (config expand_block_compressed_textures, on by default): rt_upload_textures decodes every BC1 block
    once into 16 texels and the material kernels' `_texels` instantiation fetches those; off, the device keeps the 8-byte blocks and
    decodes one per texel fetch (the reference leaves that to the texture unit: TextureLoader.cpp:208-262). One decode routine,
    so the albedo AOV (anisotropic probes at bounce 0) and a 4-bounce frame (trilinear lookups after it) are the same to the bit,
    under both schedulers; the expanded chain is 8 x the bytes."""
    results = {}
    for expand in (1, 0):
        for scheduler in ("merged", "slots"):
            scene, pt = make_pathtracer(grt, "sponza", 320, 180, 0, num_bounces=4, enable_block_compression=1, expand_block_compressed_textures=expand)
            grt.set_scheduler(pt.ctx, scheduler)
            pt.aov_enable(grt.AOV_ALBEDO); pt.update()
            pt.render(); pt.render()
            lib = grt.device_lib(); lib.rt_texture_bytes.restype = ctypes.c_size_t; lib.rt_texture_bytes.argtypes = [ctypes.c_void_p]
            results[expand, scheduler] = (pt.read_aov(grt.AOV_ALBEDO).copy(), pt.read_framebuffer().copy(), lib.rt_texture_bytes(pt.ctx))
            pt.close(); scene.close()
    for scheduler in ("merged", "slots"):
        on, off = results[1, scheduler], results[0, scheduler]
        assert np.array_equal(on[0].view(np.uint32), off[0].view(np.uint32)) and np.array_equal(on[1].view(np.uint32), off[1].view(np.uint32))
        assert off[2] > 1 << 20 and 7.9 * off[2] < on[2] <= 8 * off[2]   # (the 1x1 stand-ins of untextured slots are not compressed)
    assert np.abs(results[1, "merged"][0]).sum() > 0


def test_frame_split_in_one_process_without_python_collectives(grt):
    """host/FrameSplit.h over the C ABI's own frame exchange (rt_comm_init_all, rt_all_gather_framebuffers): two and three
    contexts of ONE GPU -- RCCL refuses a device twice in a communicator, so they exchange by stream-ordered peer copies; on
    distinct GPUs the same calls go through ncclAllGather -- render one frame as row tiles dealt round-robin. After every
    render() EVERY rank's final image is the whole frame, bit-identical to a single context's: accumulated samples,
    4-sample submissions, and SVGF + TAA frames (inputs of the filter gathered, every rank filters)."""
    for config, steps in ((dict(num_bounces=4), ("render", "render", "samples3", "render")), (dict(num_bounces=3, enable_svgf=1, enable_taa=1), ("render",) * 4)):
        images = {}
        for world in (1, 2, 3):
            scene, pt = make_pathtracer(grt, "cornellbox", 200, 152, -1, **config)
            pt.close()
            split = grt.FrameSplit(scene, 200, 152, [0] * world)
            assert split.submitting_threads == (world if world > 1 else 0)   # one submitting thread per rank
            for step in steps:
                split.update()
                if step == "render":
                    split.render()
                else:
                    split.render_samples(int(step[-1]))
            images[world] = [split.rank(r).read_framebuffer().copy() for r in range(world)]
            split.close(); scene.close()
        assert np.isfinite(images[1][0]).all() and images[1][0][..., :3].max() > 0
        for world in (2, 3):
            for r in range(world):
                assert np.array_equal(images[world][r], images[1][0]), (config, world, r)
    grt.config_reset()
