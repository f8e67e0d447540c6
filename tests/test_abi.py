"""The C-ABI shared library loads and exports every symbol include/*.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gpu_raytracer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    names = declared_functions()
    for required in ("rt_create", "rt_destroy", "rt_upload_geometry", "rt_upload_tlas", "rt_upload_instances", "rt_upload_materials",
                     "rt_upload_textures", "rt_upload_lights", "rt_upload_rng", "rt_set_sky", "rt_resize", "rt_set_camera",
                     "rt_set_svgf_matrices", "rt_set_config", "rt_set_pixel_range", "rt_render_sample", "rt_read_aov", "rt_get_counters"):
        assert required in names


def test_library_exports_every_declared_symbol(grt):
    lib = ctypes.CDLL(grt.DEVICE_LIB_PATH)
    missing = [name for name in declared_functions() if not hasattr(lib, name)]
    assert missing == []


def test_struct_sizes_match_the_header(grt):
    # rt_gpu_config: 16 x 4 B, rt_camera: 15 floats, rt_counters: 6*128 ints + 7 floats
    assert ctypes.sizeof(grt.GPUConfig) == 64
    assert ctypes.sizeof(grt.Camera) == 60
    assert ctypes.sizeof(grt.Counters) == 6 * 128 * 4 + 7 * 4


def test_create_without_gpu_fails_loudly(grt):
    import torch
    if torch.cuda.is_available():
        return  # covered by the gpu tests
    ctx = ctypes.c_void_p()
    status = grt.device_lib().rt_create(0, ctypes.byref(ctx))
    assert status != 0 and not ctx.value
    assert b"HIP device" in grt.device_lib().rt_last_error(None)


def test_host_pathtracer_refuses_to_render_without_device(grt):
    import pytest
    grt.config_reset()
    scene = grt.Scene(grt.scene_path("cornellbox"))
    pt = grt.Pathtracer(scene, 64, 64, device=-1)
    pt.update()
    with pytest.raises(RuntimeError, match="device"):
        pt.render()
    pt.close()
    scene.close()


def test_the_flat_engines_addressing_gate_sits_where_its_multiply_ends(grt):
    """The flattened scene's engine forms a node's byte offset with a 24-bit multiply (exact below 2^24 nodes) and a triangle's with a 32-bit one (below 4 GiB of
    48-byte records); rt_geometry_fits_flat_engine is the gate both rt_upload_geometry and rt_build_geometry apply (round 5 checked 4 GiB only: advisor finding)."""
    lib = grt.device_lib()
    lib.rt_geometry_fits_flat_engine.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    lib.rt_geometry_fits_flat_engine.restype = ctypes.c_int
    assert lib.rt_geometry_fits_flat_engine(1, 1) == 1
    assert lib.rt_geometry_fits_flat_engine((1 << 24) - 1, 1) == 1
    assert lib.rt_geometry_fits_flat_engine(1 << 24, 1) == 0                       # 2^24 * 80 B = 1.34 GB: far below 4 GiB, and past the multiply
    assert lib.rt_geometry_fits_flat_engine(1000, (1 << 32) // 48) == 1            # 89 478 485 triangles: 4 294 967 280 bytes
    assert lib.rt_geometry_fits_flat_engine(1000, (1 << 32) // 48 + 1) == 0
    assert lib.rt_geometry_fits_flat_engine(0, 0) == 1
