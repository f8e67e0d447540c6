"""ctypes front-end of the MI355X-native path tracer.

Two in-tree shared libraries are loaded from this directory:

* ``csrc/libgrt_device.so`` -- HIP kernels + the C ABI of ``include/gpu_raytracer_amd.h``
* ``host/libgrt_host.so``   -- C++ host classes (Scene, Mitsuba/OBJ loaders, BVH builders,
  Integrator/Pathtracer mirroring the reference's API) plus a flat C shim

Nothing here computes anything: it only marshals numpy arrays into those libraries.  The
libraries must have been built (``python __graft_entry__.py`` or ``make -C gpu-raytracer_amd``);
there is no Python or CPU fallback for the device path.
"""
import ctypes
import fcntl
import os
import tarfile
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
ASSET_DIR = os.path.join(REPO_ROOT, "assets")
DEVICE_LIB_PATH = os.environ.get("GRT_DEVICE_LIB") or os.path.join(_HERE, "csrc", "libgrt_device.so")  # override: kernel-variant experiments
HOST_LIB_PATH = os.path.join(_HERE, "host", "libgrt_host.so")

RT_MAX_BOUNCES = 128
RT_BATCH_SIZE = 1080 * 720
AOV_RADIANCE, AOV_RADIANCE_DIRECT, AOV_RADIANCE_INDIRECT, AOV_ALBEDO, AOV_NORMAL, AOV_POSITION, AOV_COUNT = range(7)
MATERIAL_LIGHT, MATERIAL_DIFFUSE, MATERIAL_PLASTIC, MATERIAL_DIELECTRIC, MATERIAL_CONDUCTOR = range(5)
FILTER_BOX, FILTER_TENT, FILTER_GAUSSIAN = range(3)


class GPUConfig(ctypes.Structure):  # rt_gpu_config
    _fields_ = [("reconstruction_filter", c_int32), ("aov_mask", c_uint32), ("num_bounces", c_int32),
                ("enable_mipmapping", c_int32), ("enable_next_event_estimation", c_int32),
                ("enable_multiple_importance_sampling", c_int32), ("enable_russian_roulette", c_int32),
                ("enable_svgf", c_int32), ("enable_spatial_variance", c_int32), ("enable_taa", c_int32),
                ("alpha_colour", c_float), ("alpha_moment", c_float), ("num_atrous_iterations", c_int32),
                ("sigma_z", c_float), ("sigma_n", c_float), ("sigma_l", c_float)]


class Camera(ctypes.Structure):  # rt_camera
    _fields_ = [("position", c_float * 3), ("bottom_left_corner", c_float * 3), ("x_axis", c_float * 3),
                ("y_axis", c_float * 3), ("pixel_spread_angle", c_float), ("aperture_radius", c_float),
                ("focal_distance", c_float)]


class Counters(ctypes.Structure):  # rt_counters
    _fields_ = [("trace", c_int32 * RT_MAX_BOUNCES), ("shadow", c_int32 * RT_MAX_BOUNCES),
                ("diffuse", c_int32 * RT_MAX_BOUNCES), ("plastic", c_int32 * RT_MAX_BOUNCES),
                ("dielectric", c_int32 * RT_MAX_BOUNCES), ("conductor", c_int32 * RT_MAX_BOUNCES),
                ("ms_generate", c_float), ("ms_trace", c_float), ("ms_sort", c_float), ("ms_shade", c_float),
                ("ms_shadow", c_float), ("ms_post", c_float), ("ms_total", c_float)]


class DeviceLibraryMissing(RuntimeError):
    pass


_device = None
_host = None


def device_lib():
    """The C-ABI library. Raises loudly when it has not been built: there is no fallback."""
    global _device
    if _device is None:
        if not os.path.exists(DEVICE_LIB_PATH):
            raise DeviceLibraryMissing("%s is missing -- run `python __graft_entry__.py` (build()) first" % DEVICE_LIB_PATH)
        # HIP reads this at the first HIP call of the process; the per-submission scheduler runs up to 18 streams.
        # The library itself leaves the environment alone; a process that initialised HIP earlier (torch imported and used
        # first) keeps HIP's default of 4 queues and the context says so when it matters.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
        lib = ctypes.CDLL(DEVICE_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        if hasattr(lib, "rt_abi_version") and lib.rt_abi_version() != 8:
            raise DeviceLibraryMissing("%s has ABI version %d, this front end was written for 8 -- rebuild (python __graft_entry__.py)" % (DEVICE_LIB_PATH, lib.rt_abi_version()))
        lib.rt_last_error.restype = c_char_p
        lib.rt_last_error.argtypes = [c_void_p]
        lib.rt_version.restype = c_char_p
        lib.rt_create.argtypes = [c_int, POINTER(c_void_p)]
        lib.rt_destroy.argtypes = [c_void_p]
        fp, u32p, u8p = POINTER(c_float), POINTER(c_uint32), POINTER(c_uint8)
        lib.rt_trace_rays.argtypes = [c_void_p] + [c_void_p] * 6 + [c_size_t, c_void_p, c_int, POINTER(c_float)]
        lib.rt_trace_shadow_rays.argtypes = [c_void_p] + [c_void_p] * 7 + [c_size_t, c_void_p, c_int, POINTER(c_float)]
        lib.rt_generate_rays.argtypes = [c_void_p, c_int, c_int, c_int] + [c_void_p] * 7
        lib.rt_random_samples.argtypes = [c_void_p, c_int, c_void_p, c_size_t, c_uint32, c_uint32, c_void_p]
        lib.rt_measure_stream_bandwidth.argtypes = [c_void_p, c_size_t, c_int, POINTER(c_float)]
        lib.rt_set_profiling.argtypes = [c_void_p, c_int]
        lib.rt_set_samples_in_flight.argtypes = [c_void_p, c_int]
        lib.rt_set_batch_size.argtypes = [c_void_p, c_int]
        lib.rt_get_counters.argtypes = [c_void_p, POINTER(Counters)]
        lib.rt_render_sample.argtypes = [c_void_p, c_int]
        lib.rt_render_samples.argtypes = [c_void_p, c_int, c_int]
        lib.rt_synchronize.argtypes = [c_void_p]
        lib.rt_set_pixel_range.argtypes = [c_void_p, c_int, c_int]
        lib.rt_read_framebuffer.argtypes = [c_void_p, c_void_p]
        lib.rt_read_aov.argtypes = [c_void_p, c_int, c_void_p, c_int]
        lib.rt_framebuffer_device_ptr.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]
        lib.rt_screen_pitch.argtypes = [c_void_p]
        lib.rt_read_luts.argtypes = [c_void_p] + [c_void_p] * 6
        lib.rt_set_config.argtypes = [c_void_p, POINTER(GPUConfig)]
        _device = lib
    return _device


def host_lib():
    global _host
    if _host is None:
        device_lib()  # libgrt_host.so links against it
        if not os.path.exists(HOST_LIB_PATH):
            raise DeviceLibraryMissing("%s is missing -- run `python __graft_entry__.py` (build()) first" % HOST_LIB_PATH)
        os.environ.setdefault("GRT_ASSET_DIR", ASSET_DIR)
        lib = ctypes.CDLL(HOST_LIB_PATH)
        lib.grt_last_error.restype = c_char_p
        lib.grt_config_set.argtypes = [c_char_p, c_double]
        lib.grt_config_get.argtypes = [c_char_p]
        lib.grt_config_get.restype = c_double
        lib.grt_scene_load.restype = c_void_p
        lib.grt_scene_load.argtypes = [c_char_p, c_char_p]
        lib.grt_scene_free.argtypes = [c_void_p]
        for name in ("grt_scene_mesh_count", "grt_scene_material_count", "grt_scene_texture_count", "grt_scene_mesh_data_count", "grt_scene_wait_until_loaded"):
            getattr(lib, name).argtypes = [c_void_p]
        lib.grt_scene_bvh_build_ms.argtypes = [c_void_p]
        lib.grt_scene_bvh_build_ms.restype = c_double
        lib.grt_scene_set_sky_scale.argtypes = [c_void_p, c_float]
        lib.grt_scene_set_camera.argtypes = [c_void_p, POINTER(c_float), POINTER(c_float), c_float]
        lib.grt_scene_get_camera.argtypes = [c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_float)]
        lib.grt_scene_set_material.argtypes = [c_void_p, c_int, c_int, POINTER(c_float), c_float]
        lib.grt_scene_material_type.argtypes = [c_void_p, c_int]
        lib.grt_mesh_data_array.restype = c_void_p
        lib.grt_mesh_data_array.argtypes = [c_void_p, c_int, c_char_p, POINTER(c_size_t)]
        lib.grt_scene_set_mesh_transform.argtypes = [c_void_p, c_int, POINTER(c_float), POINTER(c_float), c_float]
        lib.grt_scene_get_mesh_transform.argtypes = [c_void_p, c_int, POINTER(c_float), POINTER(c_float), POINTER(c_float)]
        lib.grt_pathtracer_create.restype = c_void_p
        lib.grt_pathtracer_create.argtypes = [c_void_p, c_int, c_int, c_int]
        lib.grt_ao_create.restype = c_void_p
        lib.grt_ao_create.argtypes = [c_void_p, c_int, c_int, c_int]
        lib.grt_ao_set_radius.argtypes = [c_void_p, c_float]
        lib.grt_pathtracer_free.argtypes = [c_void_p]
        lib.grt_pathtracer_update.argtypes = [c_void_p, c_float]
        lib.grt_pathtracer_render.argtypes = [c_void_p]
        lib.grt_pathtracer_set_pixel_query.argtypes = [c_void_p, c_int, c_int]
        lib.grt_pathtracer_get_pixel_query.argtypes = [c_void_p] + [POINTER(c_int)] * 4
        lib.grt_pathtracer_render_samples.argtypes = [c_void_p, c_int]
        lib.grt_pathtracer_resize.argtypes = [c_void_p, c_int, c_int]
        lib.grt_pathtracer_set_pixel_range.argtypes = [c_void_p, c_int, c_int]
        lib.grt_pathtracer_sample_index.argtypes = [c_void_p]
        lib.grt_pathtracer_screen_pitch.argtypes = [c_void_p]
        lib.grt_pathtracer_invalidate.argtypes = [c_void_p, c_char_p]
        lib.grt_pathtracer_aov_enable.argtypes = [c_void_p, c_int, c_int]
        lib.grt_pathtracer_context.restype = c_void_p
        lib.grt_pathtracer_context.argtypes = [c_void_p]
        lib.grt_pathtracer_device_blas_build_ms.restype = c_float
        lib.grt_pathtracer_device_blas_build_ms.argtypes = [c_void_p]
        lib.grt_pathtracer_static_geometry_members.restype = c_int
        lib.grt_pathtracer_static_geometry_members.argtypes = [c_void_p]
        lib.grt_pathtracer_static_geometry_whole_scene.restype = c_int
        lib.grt_pathtracer_static_geometry_whole_scene.argtypes = [c_void_p]
        lib.grt_pathtracer_skip_behind_hit.restype = c_int
        lib.grt_pathtracer_skip_behind_hit.argtypes = [c_void_p]
        lib.grt_pathtracer_static_geometry_root.restype = c_int
        lib.grt_pathtracer_static_geometry_root.argtypes = [c_void_p]
        lib.grt_pathtracer_static_geometry_top_nodes.restype = c_int
        lib.grt_pathtracer_static_geometry_top_nodes.argtypes = [c_void_p]
        lib.grt_pathtracer_set_flatten_asynchronously.restype = None
        lib.grt_pathtracer_set_flatten_asynchronously.argtypes = [c_void_p, c_int]
        lib.grt_pathtracer_set_reseat_asynchronously.restype = None
        lib.grt_pathtracer_set_reseat_asynchronously.argtypes = [c_void_p, c_int]
        lib.grt_pathtracer_reseats_completed.restype = c_int
        lib.grt_pathtracer_reseats_completed.argtypes = [c_void_p]
        lib.grt_pathtracer_reseat_pending.restype = c_int
        lib.grt_pathtracer_reseat_pending.argtypes = [c_void_p]
        lib.grt_pathtracer_last_reseat_seconds.restype = ctypes.c_double
        lib.grt_pathtracer_last_reseat_seconds.argtypes = [c_void_p]
        lib.grt_pathtracer_reflattens_completed.restype = c_int
        lib.grt_pathtracer_reflattens_completed.argtypes = [c_void_p]
        lib.grt_pathtracer_reflatten_in_progress.restype = c_int
        lib.grt_pathtracer_reflatten_in_progress.argtypes = [c_void_p]
        lib.grt_pathtracer_static_geometry_bytes.restype = ctypes.c_double
        lib.grt_pathtracer_static_geometry_bytes.argtypes = [c_void_p]
        lib.grt_pathtracer_static_geometry_build_seconds.restype = ctypes.c_double
        lib.grt_pathtracer_static_geometry_build_seconds.argtypes = [c_void_p]
        lib.grt_pathtracer_lights_total_weight.restype = c_float
        lib.grt_pathtracer_lights_total_weight.argtypes = [c_void_p]
        lib.grt_pathtracer_read_aov.argtypes = [c_void_p, c_int, c_int, c_void_p]
        lib.grt_pathtracer_read_framebuffer.argtypes = [c_void_p, c_void_p]
        lib.grt_pathtracer_save_image.argtypes = [c_void_p, c_char_p]
        lib.grt_export_image.argtypes = [c_char_p, c_int, c_int, c_int, c_void_p]
        lib.grt_pathtracer_array.restype = c_void_p
        lib.grt_pathtracer_array.argtypes = [c_void_p, c_char_p, POINTER(c_size_t)]
        lib.grt_pathtracer_sky_size.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_float)]
        lib.grt_pathtracer_texture.argtypes = [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_int)]
        lib.grt_pathtracer_device_config.argtypes = [c_void_p, POINTER(GPUConfig)]
        lib.grt_pathtracer_counters.argtypes = [c_void_p, POINTER(Counters)]
        lib.grt_frame_split_create.restype = c_void_p
        lib.grt_frame_split_create.argtypes = [c_void_p, c_int, c_int, POINTER(c_int), c_int]
        lib.grt_frame_split_free.argtypes = [c_void_p]
        lib.grt_frame_split_update.argtypes = [c_void_p, c_float]
        lib.grt_frame_split_render.argtypes = [c_void_p]
        lib.grt_frame_split_render_samples.argtypes = [c_void_p, c_int]
        lib.grt_frame_split_submitting_threads.restype = c_int
        lib.grt_frame_split_submitting_threads.argtypes = [c_void_p]
        lib.grt_frame_split_rank.restype = c_void_p
        lib.grt_frame_split_rank.argtypes = [c_void_p, c_int]
        lib.grt_build_blas.restype = c_void_p
        lib.grt_build_blas.argtypes = [c_void_p, c_int]
        lib.grt_build_static_bvh.restype = c_void_p
        lib.grt_build_static_bvh.argtypes = [c_void_p, c_int, c_int]
        lib.grt_build_device_bvh.restype = c_void_p
        lib.grt_build_device_bvh.argtypes = [c_void_p, c_int, c_int]
        lib.grt_built_array.restype = c_void_p
        lib.grt_built_array.argtypes = [c_void_p, c_char_p, POINTER(c_size_t)]
        lib.grt_built_free.argtypes = [c_void_p]
        lib.grt_built_learn_slot_order.restype = c_int
        lib.grt_built_learn_slot_order.argtypes = [c_void_p, c_int, c_int]
        _host = lib
    return _host


def _host_check(status):
    if status != 0:
        raise RuntimeError(host_lib().grt_last_error().decode(errors="replace"))


def _view(ptr, nbytes, dtype):
    if not ptr or nbytes == 0:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


# ---- scenes ---------------------------------------------------------------------------------------

def scene_path(name):
    """Path of a bundled scene's xml; archives under assets/scenes are unpacked on first use.

    'cornellbox' and 'sponza' are the geometry of the reference's Data/cornellbox and
    Data/Sponza (Crytek Sponza), packed because /root/reference does not exist on the GPU box.
    """
    cache = os.path.join(ASSET_DIR, "_cache")
    if name == "sponza_reference_maps":   # Sponza with the reference's own texture files (install_reference_sponza_textures)
        if not reference_sponza_textures_installed():
            raise FileNotFoundError("the reference's Sponza textures are not installed (build() copies them where /root/reference is mounted)")
        scene_path("sponza")
        return os.path.join(cache, "Sponza", "scene_reference_maps.xml")
    table = {"cornellbox": ("cornellbox.tar.xz", "cornellbox/scene.xml"), "sponza": ("sponza_geometry.tar.xz", "Sponza/scene.xml")}
    if name not in table:
        raise KeyError("unknown bundled scene %r" % name)
    archive, xml = table[name]
    target = os.path.join(cache, xml)
    os.makedirs(cache, exist_ok=True)
    with open(os.path.join(cache, ".lock"), "w") as lock:   # the ranks of a multi-GPU job start together
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(target):
                with tarfile.open(os.path.join(ASSET_DIR, "scenes", archive)) as tar:
                    tar.extractall(cache)
            if name == "sponza":
                _unpack_sponza_textures(cache)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return target


REFERENCE_SPONZA_TEXTURES = "/root/reference/Data/Sponza/textures"


def install_reference_sponza_textures():
    """Called by build() (which runs where /root/reference is mounted): the 19 diffuse maps Data/Sponza/scene.xml finds upstream
    go, as they are, into assets/_cache/Sponza/textures_reference/ next to a copy of the scene file that names them
    (scene_path("sponza_reference_maps")). assets/_cache is git-ignored (nothing of the reference enters the history) but
    travels to the GPU box with the snapshot, so the benchmark there renders the real texture set instead of the quarter-size
    maps replicated 4x4 that travel inside the repository. Returns the number of maps in place (0: no reference mount)."""
    import re
    import shutil
    xml = scene_path("sponza")
    directory = os.path.dirname(xml)
    target = os.path.join(directory, "textures_reference")
    done = os.path.join(target, ".complete")
    if os.path.exists(done):
        return len([n for n in os.listdir(target) if n.endswith(".tga")])
    if not os.path.isdir(REFERENCE_SPONZA_TEXTURES):
        return 0
    text = open(xml).read()
    names = sorted(set(re.findall(r"textures[\\/]+([A-Za-z0-9_]+\.tga)", text)))
    os.makedirs(target, exist_ok=True)
    count = 0
    for name in names:
        source = os.path.join(REFERENCE_SPONZA_TEXTURES, name)
        if os.path.exists(source):       # (5 of the 24 are missing upstream as well: the loader's fallback texel, as in the reference)
            shutil.copyfile(source, os.path.join(target, name + ".part")); os.replace(os.path.join(target, name + ".part"), os.path.join(target, name))
            count += 1
    with open(os.path.join(directory, "scene_reference_maps.xml"), "w") as f:
        f.write(re.sub(r"textures[\\/]+([A-Za-z0-9_]+\.tga)", r"textures_reference/\1", text))
    open(done, "w").close()
    return count


def reference_sponza_textures_installed():
    return os.path.exists(os.path.join(ASSET_DIR, "_cache", "Sponza", "textures_reference", ".complete"))


def _unpack_sponza_textures(cache):
    """The 19 diffuse maps of Data/Sponza/textures travel at a quarter of their side length
    (tools/pack_sponza_textures.py); every texel is replicated 4x4 here so that the renderer loads
    textures of the reference's dimensions (mip chain depth, memory footprint, LOD selection)."""
    done = os.path.join(cache, "Sponza", "textures", ".complete")
    archive = os.path.join(ASSET_DIR, "scenes", "sponza_textures_256.tar.xz")
    if os.path.exists(done) or not os.path.exists(archive):
        return
    os.makedirs(os.path.dirname(done), exist_ok=True)
    with tarfile.open(archive) as tar:
        for member in tar.getmembers():
            data = tar.extractfile(member).read()
            w, h, bpp, desc = np.frombuffer(data, np.uint16, 2, 12).tolist() + [data[16], data[17]]
            channels = bpp // 8
            px = np.frombuffer(data, np.uint8, w * h * channels, 18).reshape(h, w, channels)
            big = np.repeat(np.repeat(px, 4, axis=0), 4, axis=1)
            header = bytearray(data[:18])
            header[12:16] = np.array([w * 4, h * 4], np.uint16).tobytes()
            tmp = os.path.join(os.path.dirname(done), os.path.basename(member.name))
            with open(tmp + ".part", "wb") as f:
                f.write(bytes(header) + big.tobytes())
            os.replace(tmp + ".part", tmp)
    open(done, "w").close()


def config_reset():
    host_lib().grt_config_reset()


BVH_TYPES = {"sbvh": 1, "sah": 2, "bvh": 2, "bvh4": 4, "bvh8": 8}   # the reference's --bvh names (Args.cpp:71-84)


def config_set(**kwargs):
    lib = host_lib()
    for key, value in kwargs.items():
        if key == "bvh_type" and isinstance(value, str):
            value = BVH_TYPES[value.lower()]
        if lib.grt_config_set(key.encode(), float(value)) != 0:
            raise KeyError(lib.grt_last_error().decode(errors="replace"))


def load_texture(filename, block_compression=False):
    """Decodes an image file the way the scene loader does (TGA / PPM / PNG / BMP: sRGB -> linear RGBA8 +
    box-filtered mips; DDS: stored DXT levels as they are). Returns a list of (height, width, 4) uint8 levels.
    block_compression: False = the decoded levels; True = what survives BC1 (power-of-two textures only, the
    scene loader's default); None = whatever `enable_block_compression` says."""
    lib = host_lib()
    if block_compression is not None:
        previous = config_get("enable_block_compression")
        config_set(enable_block_compression=int(bool(block_compression)))
        try:
            return load_texture(filename, None)
        finally:
            config_set(enable_block_compression=previous)
    lib.grt_texture_load.restype = c_void_p
    lib.grt_texture_load.argtypes = [c_char_p]
    lib.grt_texture_data.restype = c_void_p
    lib.grt_texture_data.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]
    lib.grt_texture_free.argtypes = [c_void_p]
    handle = lib.grt_texture_load(str(filename).encode())
    if not handle:
        raise RuntimeError(lib.grt_last_error().decode(errors="replace"))
    w, h, levels, nbytes = c_int(), c_int(), c_int(), c_size_t()
    ptr = lib.grt_texture_data(handle, byref(w), byref(h), byref(levels), byref(nbytes))
    data = _view(ptr, nbytes.value, np.uint8).copy()
    lib.grt_texture_free(handle)
    out, offset = [], 0
    for l in range(levels.value):
        lw, lh = max(w.value >> l, 1), max(h.value >> l, 1)
        out.append(data[offset:offset + lw * lh * 4].reshape(lh, lw, 4))
        offset += lw * lh * 4
    return out


def export_image(filename, rgb):
    """Writes a (height, width, 3) float32 image (row 0 at the bottom, as the integrator holds frames)
    through the host's PPM / EXR exporters."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    lib = host_lib()
    _host_check(lib.grt_export_image(str(filename).encode(), w, w, h, rgb.ctypes.data))


def config_get(key):
    return host_lib().grt_config_get(key.encode())


class Scene:
    """reference: Src/Renderer/Scene.h -- loads one .xml/.obj scene file."""

    def __init__(self, filename, sky=None):
        lib = host_lib()
        self.handle = lib.grt_scene_load(os.fsencode(filename), os.fsencode(sky) if sky else b"")
        if not self.handle:
            raise RuntimeError("scene load failed: " + lib.grt_last_error().decode(errors="replace"))

    def close(self):
        if self.handle:
            host_lib().grt_scene_free(self.handle)
            self.handle = None

    def wait_until_loaded(self):
        _host_check(host_lib().grt_scene_wait_until_loaded(self.handle))

    @property
    def mesh_count(self):
        return host_lib().grt_scene_mesh_count(self.handle)

    @property
    def material_count(self):
        return host_lib().grt_scene_material_count(self.handle)

    @property
    def mesh_data_count(self):
        return host_lib().grt_scene_mesh_data_count(self.handle)

    @property
    def bvh_build_ms(self):
        return host_lib().grt_scene_bvh_build_ms(self.handle)

    def mesh_data_array(self, index, name, dtype):
        n = c_size_t()
        ptr = host_lib().grt_mesh_data_array(self.handle, index, name.encode(), byref(n))
        return _view(ptr, n.value, dtype)

    def describe(self):
        """One line per object the loaders produced (floats as bit patterns)."""
        lib = host_lib()
        lib.grt_scene_describe.restype = ctypes.c_size_t
        lib.grt_scene_describe.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        n = lib.grt_scene_describe(self.handle, None, 0)
        buf = ctypes.create_string_buffer(n)
        lib.grt_scene_describe(self.handle, buf, n)
        return buf.value.decode(errors="replace")

    def texture(self, index):
        """-> dict(width, height, lod_width, lod_height, mip_offsets (texels), texels (RGBA8, all levels))"""
        lib = host_lib()
        lib.grt_scene_texture_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.grt_scene_texture_data.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        info = (ctypes.c_int * 6)()
        if lib.grt_scene_texture_info(self.handle, index, info) != 0:
            raise RuntimeError(lib.grt_last_error().decode(errors="replace"))
        texels = np.zeros((info[5], 4), np.uint8); offsets = np.zeros(info[2], np.int32)
        lib.grt_scene_texture_data(self.handle, index, texels.ctypes.data, offsets.ctypes.data)
        return dict(width=info[0], height=info[1], lod_width=info[3], lod_height=info[4], mip_offsets=offsets, texels=texels)

    def sky(self):
        lib = host_lib()
        lib.grt_scene_sky.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        out = np.zeros((lib.grt_scene_sky(self.handle, None), 4), np.float32)
        lib.grt_scene_sky(self.handle, out.ctypes.data)
        return out

    def set_camera(self, position, rotation, fov=-1.0):
        pos = (c_float * 3)(*position)
        rot = (c_float * 4)(*rotation)
        host_lib().grt_scene_set_camera(self.handle, pos, rot, float(fov))

    def get_camera(self):
        pos, rot, fov = (c_float * 3)(), (c_float * 4)(), c_float()
        host_lib().grt_scene_get_camera(self.handle, pos, rot, byref(fov))
        return list(pos), list(rot), fov.value

    def set_sky_scale(self, scale):
        host_lib().grt_scene_set_sky_scale(self.handle, float(scale))

    def mesh_transform(self, index):
        """(position[3], rotation quaternion xyzw[4], scale) of mesh `index` (Mesh.h: position / rotation / scale)."""
        pos, rot, scale = (c_float * 3)(), (c_float * 4)(), c_float()
        _host_check(host_lib().grt_scene_get_mesh_transform(self.handle, index, pos, rot, byref(scale)))
        return list(pos), list(rot), scale.value

    def set_mesh_transform(self, index, position, rotation_xyzw, scale):
        """Edit a mesh's transform as the reference's UI does; the next update() with invalidate("scene")
        (or enable_scene_update) rebuilds the TLAS."""
        _host_check(host_lib().grt_scene_set_mesh_transform(self.handle, index, (c_float * 3)(*position), (c_float * 4)(*rotation_xyzw), float(scale)))

    def set_material(self, index, mtype, diffuse=None, linear_roughness=0.5):
        d = (c_float * 3)(*diffuse) if diffuse is not None else None
        _host_check(host_lib().grt_scene_set_material(self.handle, index, mtype, d, float(linear_roughness)))

    def material_type(self, index):
        return host_lib().grt_scene_material_type(self.handle, index)


_ARRAY_DTYPES = {
    "triangles": np.float32, "bvh8_nodes": np.uint8, "bvh2_nodes": np.uint8, "bvh4_nodes": np.uint8, "reverse_indices": np.int32,
    "mesh_bvh_root_indices": np.int32, "mesh_material_ids": np.int32, "mesh_transforms": np.float32,
    "mesh_transforms_inv": np.float32, "mesh_transforms_prev": np.float32, "material_types": np.uint8,
    "materials": np.float32, "media": np.float32, "tlas_indices": np.int32, "tlas_nodes": np.uint8,
    "tlas_raw_nodes": np.uint8, "pmj_samples": np.float32, "blue_noise": np.uint8,
    "light_triangle_indices": np.int32, "light_triangle_cumulative_probability": np.float32,
    "light_mesh_cumulative_probability": np.float32, "light_mesh_triangle_span": np.int32,
    "light_mesh_transform_indices": np.int32, "sky": np.float32, "camera": np.uint8, "svgf_matrices": np.float32,
    "scene_order_roots": np.int32, "scene_order_materials": np.int32, "scene_order_transforms": np.float32,
    "scene_order_transforms_inv": np.float32, "scene_order_transforms_prev": np.float32, "scene_order_boxes": np.float32,
    "alias_mesh_ids": np.int32, "alias_triangle_ids": np.int32,
}


class Pathtracer:
    """reference: Src/Renderer/Integrators/Pathtracer.h -- update()/render() protocol.

    device < 0 creates a host-only integrator that bakes the device data formats but cannot
    render (used by the CPU tests and the oracle).
    """

    _create = "grt_pathtracer_create"

    def __init__(self, scene, width, height, device=0):
        lib = host_lib()
        self.scene = scene
        self.handle = getattr(lib, self._create)(scene.handle, width, height, device)
        if not self.handle:
            raise RuntimeError("%s creation failed: %s" % (type(self).__name__, lib.grt_last_error().decode(errors="replace")))
        self.width, self.height = width, height

    def close(self):
        if self.handle:
            host_lib().grt_pathtracer_free(self.handle)
            self.handle = None

    @property
    def ctx(self):
        return host_lib().grt_pathtracer_context(self.handle)

    @property
    def sample_index(self):
        return host_lib().grt_pathtracer_sample_index(self.handle)

    @property
    def pitch(self):
        return host_lib().grt_pathtracer_screen_pitch(self.handle)

    @property
    def device_blas_build_ms(self):
        """config device_blas = 1: what the BLAS build took on the device (0 when the host built the trees)."""
        return float(host_lib().grt_pathtracer_device_blas_build_ms(self.handle))

    @property
    def static_geometry_members(self):
        """config merge_static = 1: instances flattened into the one static bottom-level tree (0: none, or dissolved)."""
        return int(host_lib().grt_pathtracer_static_geometry_members(self.handle))

    @property
    def static_geometry_whole_scene(self):
        """Every instance is in the flattened tree: there is no TLAS, rays start inside the tree (rt_set_static_geometry)."""
        return bool(host_lib().grt_pathtracer_static_geometry_whole_scene(self.handle))

    @property
    def skip_behind_hit(self):
        """Closest-hit rays drop stacked groups of children behind the hit they hold (config skip_behind_hit AND a one-tree scene: rt_set_skip_behind_hit)."""
        return bool(host_lib().grt_pathtracer_skip_behind_hit(self.handle))

    @property
    def static_geometry_top_levels(self):
        """(root node of the flattened tree, nodes from it that make up its top three levels): the tree is numbered breadth-first, what
        every ray walks comes first."""
        return int(host_lib().grt_pathtracer_static_geometry_root(self.handle)), int(host_lib().grt_pathtracer_static_geometry_top_nodes(self.handle))

    def set_flatten_asynchronously(self, enable):
        """True (default): when a flattened instance starts to move the new tree is built on a worker thread while frames are rendered
        in the reference's layout; False: rebuilt inside update() (a stall of the build time)."""
        host_lib().grt_pathtracer_set_flatten_asynchronously(self.handle, 1 if enable else 0)

    def set_reseat_asynchronously(self, enable):
        """True (default): the flattened tree is seated again (config static_reseat_distance; device-built trees: for the first time) on a worker thread and its
        nodes are swapped in between two frames; False: inside update()."""
        host_lib().grt_pathtracer_set_reseat_asynchronously(self.handle, 1 if enable else 0)

    @property
    def reseats_completed(self):
        return int(host_lib().grt_pathtracer_reseats_completed(self.handle))

    @property
    def reseat_pending(self):
        return bool(host_lib().grt_pathtracer_reseat_pending(self.handle))

    @property
    def last_reseat_seconds(self):
        return float(host_lib().grt_pathtracer_last_reseat_seconds(self.handle))

    @property
    def reflattens_completed(self):
        return int(host_lib().grt_pathtracer_reflattens_completed(self.handle))

    @property
    def reflatten_in_progress(self):
        """0: none; 1: a worker thread is building the tree; 2: it is done, the next update() installs it."""
        return int(host_lib().grt_pathtracer_reflatten_in_progress(self.handle))

    @property
    def static_geometry_bytes(self):
        """Device bytes the flattened tree adds: its triangle copies (shading + traversal records + names) and its nodes."""
        return int(host_lib().grt_pathtracer_static_geometry_bytes(self.handle))

    @property
    def static_geometry_build_seconds(self):
        return float(host_lib().grt_pathtracer_static_geometry_build_seconds(self.handle))

    @property
    def lights_total_weight(self):
        return host_lib().grt_pathtracer_lights_total_weight(self.handle)

    def update(self, delta=0.0):
        _host_check(host_lib().grt_pathtracer_update(self.handle, float(delta)))

    def render(self):
        _host_check(host_lib().grt_pathtracer_render(self.handle))

    def render_samples(self, count):
        """`count` samples per pixel as one wavefront; same image as `count` x (update(); render())."""
        _host_check(host_lib().grt_pathtracer_render_samples(self.handle, int(count)))

    def set_pixel_query(self, x, y):
        """Integrator::set_pixel_query (window coordinates, y top-down): which mesh / triangle is under
        this pixel? Armed for the next render(); the update() after it fetches the answer."""
        host_lib().grt_pathtracer_set_pixel_query(self.handle, int(x), int(y))

    @property
    def pixel_query(self):
        """(pixel_index, scene mesh index, triangle id, status) -- status 0 inactive, 1 pending, 2 output ready."""
        v = [c_int() for _ in range(4)]
        host_lib().grt_pathtracer_get_pixel_query(self.handle, *[byref(i) for i in v])
        return tuple(i.value for i in v)

    def invalidate(self, what):
        host_lib().grt_pathtracer_invalidate(self.handle, what.encode())

    def aov_enable(self, aov, enable=True):
        host_lib().grt_pathtracer_aov_enable(self.handle, aov, 1 if enable else 0)

    def set_pixel_range(self, offset, count):
        _host_check(host_lib().grt_pathtracer_set_pixel_range(self.handle, offset, count))

    def array(self, name):
        n = c_size_t()
        ptr = host_lib().grt_pathtracer_array(self.handle, name.encode(), byref(n))
        return _view(ptr, n.value, _ARRAY_DTYPES[name])

    def view_projection(self):
        """(view_projection, view_projection_prev) as uploaded for SVGF, 16 row-major floats each."""
        m = self.array("svgf_matrices")
        return m[:16].tolist(), m[16:].tolist()

    def camera(self):
        cam = Camera()
        raw = self.array("camera")
        ctypes.memmove(byref(cam), raw.ctypes.data, ctypes.sizeof(cam))
        return cam

    def device_config(self):
        cfg = GPUConfig()
        host_lib().grt_pathtracer_device_config(self.handle, byref(cfg))
        return cfg

    def sky(self):
        w, h, s = c_int(), c_int(), c_float()
        host_lib().grt_pathtracer_sky_size(self.handle, byref(w), byref(h), byref(s))
        return self.array("sky"), w.value, h.value, s.value

    def textures(self):
        out = []
        i = 0
        while True:
            texels, w, h, levels = c_void_p(), c_int(), c_int(), c_int()
            if host_lib().grt_pathtracer_texture(self.handle, i, byref(texels), byref(w), byref(h), byref(levels)) != 0:
                break
            count = 0
            for l in range(levels.value):
                count += max(w.value >> l, 1) * max(h.value >> l, 1)
            out.append((_view(texels.value, count * 4, np.uint8), w.value, h.value, levels.value))
            i += 1
        return out

    def texture_lod_size(self, index):
        """(lod_width, lod_height) of texture `index`: what enters its LOD bias; (0, 0) = its own size."""
        w, h = c_int(), c_int()
        lib = host_lib()
        lib.grt_pathtracer_texture_lod_size.argtypes = [c_void_p, c_int, POINTER(c_int), POINTER(c_int)]
        if lib.grt_pathtracer_texture_lod_size(self.handle, index, byref(w), byref(h)) != 0:
            raise IndexError(index)
        return w.value, h.value

    def read_framebuffer(self):
        image = np.zeros((self.height, self.pitch, 4), np.float32)
        _host_check(host_lib().grt_pathtracer_read_framebuffer(self.handle, image.ctypes.data))
        return image

    def save_image(self, filename):
        """Screenshot like the reference's `-o`: .ppm (ACES + gamma, 8 bit) or .exr (raw radiance, half),
        plus albedo.exr / normal.exr / position.exr next to it for the enabled AOVs (Main.cpp:199-246)."""
        _host_check(host_lib().grt_pathtracer_save_image(self.handle, str(filename).encode()))

    def read_aov(self, aov, accumulated=True):
        image = np.zeros((self.height, self.pitch, 4), np.float32)
        _host_check(host_lib().grt_pathtracer_read_aov(self.handle, aov, 1 if accumulated else 0, image.ctypes.data))
        return image

    def counters(self):
        c = Counters()
        _host_check(host_lib().grt_pathtracer_counters(self.handle, byref(c)))
        return c


# ---- kernel-level entry points of the C ABI -----------------------------------------------------------

class FrameSplit:
    """host/FrameSplit.h: one frame over several GPUs from one process -- row tiles dealt round-robin to one Pathtracer per
    entry of `devices` (an ordinal may repeat: contexts sharing a GPU exchange by peer copies), ONE grouped all-gather over
    RCCL per render(); no torch.distributed. rank(r) is that rank's Pathtracer (owned by the split)."""

    def __init__(self, scene, width, height, devices):
        lib = host_lib()
        self.scene, self.width, self.height, self.world = scene, width, height, len(devices)
        ordinals = (c_int * len(devices))(*devices)
        self.handle = lib.grt_frame_split_create(scene.handle, width, height, ordinals, len(devices))
        if not self.handle:
            raise RuntimeError("FrameSplit creation failed: %s" % lib.grt_last_error().decode(errors="replace"))

    def close(self):
        if self.handle:
            host_lib().grt_frame_split_free(self.handle)
            self.handle = None

    def rank(self, r):
        view = Pathtracer.__new__(Pathtracer)
        view.scene, view.width, view.height = self.scene, self.width, self.height
        view.handle = host_lib().grt_frame_split_rank(self.handle, r)
        view.close = lambda: None      # owned by the split
        return view

    @property
    def submitting_threads(self):
        """Host threads that enqueue the ranks' launches: one per rank (0 for a single rank: the caller's thread)."""
        return int(host_lib().grt_frame_split_submitting_threads(self.handle))

    def update(self, delta=0.0):
        _host_check(host_lib().grt_frame_split_update(self.handle, float(delta)))

    def render(self):
        _host_check(host_lib().grt_frame_split_render(self.handle))

    def render_samples(self, count):
        _host_check(host_lib().grt_frame_split_render_samples(self.handle, int(count)))


class AO(Pathtracer):
    """reference: Src/Renderer/Integrators/AO.h -- the ambient-occlusion integrator. Shares the
    update()/render()/read_* protocol and the staging arrays with Pathtracer; `radius` is AO::ao_radius."""
    _create = "grt_ao_create"

    def __init__(self, scene, width, height, device=0, radius=1.0):
        super().__init__(scene, width, height, device)
        self.radius = radius

    @property
    def radius(self):
        return self._radius

    @radius.setter
    def radius(self, value):
        _host_check(host_lib().grt_ao_set_radius(self.handle, float(value)))
        self._radius = float(value)


def _dev_check(ctx, status):
    if status != 0:
        raise RuntimeError("device layer: " + device_lib().rt_last_error(ctx).decode())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def trace_rays(ctx, origin, direction, repeat=1):
    """rt_trace_rays: origin/direction are (3, N) float32 SoA. Returns (hits uint32[N,4], mean kernel ms)."""
    o, d = _f32(origin), _f32(direction)
    n = o.shape[1]
    hits = np.zeros((n, 4), np.uint32)
    ms = c_float()
    _dev_check(ctx, device_lib().rt_trace_rays(ctx, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, d[0].ctypes.data, d[1].ctypes.data, d[2].ctypes.data, n, hits.ctypes.data, repeat, byref(ms)))
    return hits, ms.value


def trace_shadow_rays(ctx, origin, direction, max_distance, repeat=1):
    o, d, m = _f32(origin), _f32(direction), _f32(max_distance)
    n = o.shape[1]
    occluded = np.zeros(n, np.uint8)
    ms = c_float()
    _dev_check(ctx, device_lib().rt_trace_shadow_rays(ctx, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, d[0].ctypes.data, d[1].ctypes.data, d[2].ctypes.data, m.ctypes.data, n, occluded.ctypes.data, repeat, byref(ms)))
    return occluded, ms.value


def generate_rays(ctx, sample_index, pixel_offset, pixel_count):
    o = np.zeros((3, pixel_count), np.float32)
    d = np.zeros((3, pixel_count), np.float32)
    px = np.zeros(pixel_count, np.uint32)
    _dev_check(ctx, device_lib().rt_generate_rays(ctx, sample_index, pixel_offset, pixel_count, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, d[0].ctypes.data, d[1].ctypes.data, d[2].ctypes.data, px.ctypes.data))
    return o, d, px


def random_samples(ctx, dimension, pixel_indices, bounce, sample_index):
    px = np.ascontiguousarray(pixel_indices, dtype=np.uint32)
    out = np.zeros((px.size, 2), np.float32)
    _dev_check(ctx, device_lib().rt_random_samples(ctx, dimension, px.ctypes.data, px.size, bounce, sample_index, out.ctypes.data))
    return out


def measure_stream_bandwidth(ctx, nbytes=1 << 30, repeat=5):
    gbps = c_float()
    _dev_check(ctx, device_lib().rt_measure_stream_bandwidth(ctx, nbytes, repeat, byref(gbps)))
    return gbps.value


def read_luts(ctx):
    shapes = [4096, 4096, 256, 256, 1024, 32]
    arrays = [np.zeros(s, np.float32) for s in shapes]
    _dev_check(ctx, device_lib().rt_read_luts(ctx, *[a.ctypes.data for a in arrays]))
    return arrays


def set_trace_statistics(ctx, enable):
    lib = device_lib()
    lib.rt_set_trace_statistics.argtypes = [c_void_p, c_int]
    _dev_check(ctx, lib.rt_set_trace_statistics(ctx, 1 if enable else 0))


def get_trace_statistics(ctx):
    """Returns {'closest': {...}, 'shadow': {...}} with nodes / triangles / instances / rays and the
    algorithmic bytes they imply (24 B ray + 16 B hit | 4 B max_distance, 80 B per node, 48 B per
    triangle, 52 B per transformed instance entry, 4 B per identity entry)."""
    lib = device_lib()
    lib.rt_get_trace_statistics.argtypes = [c_void_p, c_void_p]
    raw = np.zeros(10, np.uint64)
    _dev_check(ctx, lib.rt_get_trace_statistics(ctx, raw.ctypes.data))
    out = {}
    for k, name in enumerate(("closest", "shadow")):
        nodes, tris, ix, ii, rays = [int(v) for v in raw[5 * k:5 * k + 5]]
        per_ray = 24 + (16 if k == 0 else 4)
        out[name] = dict(nodes=nodes, triangles=tris, instances_transformed=ix, instances_identity=ii, rays=rays,
                         algorithmic_bytes=per_ray * rays + 80 * nodes + 48 * tris + 52 * ix + 4 * ii)
    return out


def set_samples_in_flight(ctx, count):
    """Samples per pixel rendered concurrently (1..4); results do not depend on it."""
    _dev_check(ctx, device_lib().rt_set_samples_in_flight(ctx, int(count)))


def set_profiling(ctx, enable):
    """False/0 off; True/1 per-stage events (serialised); 2 events around the trace launches only."""
    _dev_check(ctx, device_lib().rt_set_profiling(ctx, int(enable)))


SCHEDULER_MERGED, SCHEDULER_SLOTS = 0, 1


def set_scheduler(ctx, scheduler):
    """'merged' (default): consecutive submissions feed one wavefront; 'slots': one launch chain per submission,
    several in flight on their own streams (rt_set_scheduler)."""
    if isinstance(scheduler, str):
        scheduler = {"merged": SCHEDULER_MERGED, "slots": SCHEDULER_SLOTS}[scheduler]
    lib = device_lib()
    lib.rt_set_scheduler.argtypes = [c_void_p, c_int]
    _dev_check(ctx, lib.rt_set_scheduler(ctx, int(scheduler)))




def set_svgf_tiles(ctx, enable):
    """True (default): the a-trous passes of the SVGF filter stage a workgroup's taps in LDS; False: every tap is a global
    load (rt_set_svgf_tiles). Images are bit-identical."""
    lib = device_lib()
    lib.rt_set_svgf_tiles.argtypes = [c_void_p, c_int]
    _dev_check(ctx, lib.rt_set_svgf_tiles(ctx, 1 if enable else 0))


def set_frame_pipelining(ctx, enable):
    lib = device_lib()
    lib.rt_set_frame_pipelining.argtypes = [c_void_p, c_int]
    _dev_check(ctx, lib.rt_set_frame_pipelining(ctx, 1 if enable else 0))


def set_stream_batch(ctx, paths):
    """Paths the submissions of one iteration of the merged wavefront may bring (frame pipelining on); 0: the default of
    1920 x 1080 x 4. A burst of frames declared this way enters the wavefront together (rt_set_stream_batch)."""
    lib = device_lib()
    lib.rt_set_stream_batch.argtypes = [c_void_p, ctypes.c_longlong]
    _dev_check(ctx, lib.rt_set_stream_batch(ctx, int(paths)))


def advance(ctx):
    """One iteration of the merged wavefront without new samples (no-op when nothing is in flight)."""
    lib = device_lib()
    lib.rt_advance.argtypes = [c_void_p]
    _dev_check(ctx, lib.rt_advance(ctx))


def submissions_completed(ctx):
    lib = device_lib()
    lib.rt_submissions_completed.argtypes = [c_void_p, c_void_p]
    n = ctypes.c_uint64(0)
    _dev_check(ctx, lib.rt_submissions_completed(ctx, byref(n)))
    return int(n.value)


# kinds of rt_get_launch_timings (include/gpu_raytracer_amd.h, RT_TIMING_*)
TIMING_KINDS = {"trace": 0, "shadow": 1, "sort": 2, "generate": 3, "accumulate": 4, "material_diffuse": 5, "material_plastic": 6,
                "material_dielectric": 7, "material_conductor": 8, "svgf_reproject": 9, "svgf_variance": 10, "svgf_atrous": 11,
                "svgf_finalize": 12, "taa": 13, "taa_finalize": 14}


def launch_timings(ctx, kind=0):
    """Durations (ms) of the launches of one kind timed by set_profiling(ctx, 2 or 3) since the last call for that kind."""
    kind = TIMING_KINDS.get(kind, kind)
    lib = device_lib()
    lib.rt_get_launch_timings.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p]
    n = c_int(0)
    _dev_check(ctx, lib.rt_get_launch_timings(ctx, int(kind), None, 0, byref(n)))
    out = np.zeros(max(n.value, 1), np.float32)
    _dev_check(ctx, lib.rt_get_launch_timings(ctx, int(kind), out.ctypes.data, n.value, byref(n)))
    return out[:n.value]


def trace_statistics_history(ctx):
    """(rows, 10) uint64: the trace statistics after each iteration of the merged wavefront (cumulative)."""
    lib = device_lib()
    lib.rt_get_trace_statistics_history.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
    n = c_int(0)
    _dev_check(ctx, lib.rt_get_trace_statistics_history(ctx, None, 0, byref(n)))
    out = np.zeros((max(n.value, 1), 10), np.uint64)
    _dev_check(ctx, lib.rt_get_trace_statistics_history(ctx, out.ctypes.data, n.value, byref(n)))
    return out[:n.value]


def algorithmic_bytes(row10):
    """SURVEY.md 8d bytes of ten trace-statistics counters {closest: nodes, triangles, transformed, identity, rays; shadow: same}."""
    r = [int(v) for v in row10]
    closest = 40 * r[4] + 80 * r[0] + 48 * r[1] + 52 * r[2] + 4 * r[3]
    shadow = 28 * r[9] + 80 * r[5] + 48 * r[6] + 52 * r[7] + 4 * r[8]
    return closest, shadow
