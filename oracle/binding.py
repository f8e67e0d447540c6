"""ctypes binding of liboracle.so / _ref/libref_bvh.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It turns the host staging arrays of a gpu_raytracer_amd.Pathtracer (exactly what the device
is given) into an ``oracle_scene`` and calls the CPU restatement on it.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libref_bvh.so")

RT_MAX_BOUNCES = 128
RT_AOV_COUNT = 6
RT_AOV_RADIANCE, RT_AOV_RADIANCE_DIRECT, RT_AOV_RADIANCE_INDIRECT, RT_AOV_ALBEDO, RT_AOV_NORMAL, RT_AOV_POSITION = range(6)


class GPUConfig(Structure):
    _fields_ = [("reconstruction_filter", c_int32), ("aov_mask", c_uint32), ("num_bounces", c_int32),
                ("enable_mipmapping", c_int32), ("enable_next_event_estimation", c_int32),
                ("enable_multiple_importance_sampling", c_int32), ("enable_russian_roulette", c_int32),
                ("enable_svgf", c_int32), ("enable_spatial_variance", c_int32), ("enable_taa", c_int32),
                ("alpha_colour", c_float), ("alpha_moment", c_float), ("num_atrous_iterations", c_int32),
                ("sigma_z", c_float), ("sigma_n", c_float), ("sigma_l", c_float)]


class Camera(Structure):
    _fields_ = [("position", c_float * 3), ("bottom_left_corner", c_float * 3), ("x_axis", c_float * 3),
                ("y_axis", c_float * 3), ("pixel_spread_angle", c_float), ("aperture_radius", c_float),
                ("focal_distance", c_float)]


class OracleTexture(Structure):
    _fields_ = [("texels", c_void_p), ("width", c_int32), ("height", c_int32), ("mip_levels", c_int32), ("lod_width", c_int32), ("lod_height", c_int32)]


class OracleScene(Structure):
    _fields_ = [
        ("triangles", c_void_p), ("triangle_count", c_int32),
        ("bvh8_nodes", c_void_p), ("bvh2_nodes", c_void_p), ("bvh4_nodes", c_void_p), ("bvh_type", c_int32),
        ("mesh_bvh_root_indices", c_void_p), ("mesh_material_ids", c_void_p),
        ("mesh_transforms", c_void_p), ("mesh_transforms_inv", c_void_p), ("mesh_transforms_prev", c_void_p), ("mesh_count", c_int32),
        ("material_types", c_void_p), ("materials", c_void_p), ("material_count", c_int32),
        ("media", c_void_p), ("medium_count", c_int32),
        ("textures", c_void_p), ("texture_count", c_int32),
        ("light_triangle_indices", c_void_p), ("light_triangle_cumulative_probability", c_void_p), ("light_triangle_count", c_int32),
        ("light_mesh_cumulative_probability", c_void_p), ("light_mesh_triangle_span", c_void_p), ("light_mesh_transform_indices", c_void_p),
        ("light_mesh_count", c_int32), ("lights_total_weight", c_float),
        ("pmj_samples", c_void_p), ("blue_noise", c_void_p),
        ("lut_dielectric_directional_albedo_enter", c_void_p), ("lut_dielectric_directional_albedo_leave", c_void_p),
        ("lut_dielectric_albedo_enter", c_void_p), ("lut_dielectric_albedo_leave", c_void_p),
        ("lut_conductor_directional_albedo", c_void_p), ("lut_conductor_albedo", c_void_p),
        ("sky", c_void_p), ("sky_width", c_int32), ("sky_height", c_int32), ("sky_scale", c_float),
        ("camera", Camera), ("config", GPUConfig),
        ("view_projection", c_float * 16), ("view_projection_prev", c_float * 16),
        ("screen_width", c_int32), ("screen_height", c_int32), ("screen_pitch", c_int32),
        ("alias_mesh_ids", c_void_p), ("alias_triangle_ids", c_void_p),
        ("static_whole_scene", c_int32), ("skip_behind_hit", c_int32),
    ]


class TraceStats(Structure):
    _fields_ = [("nodes", c_uint64), ("triangles", c_uint64), ("instances_transformed", c_uint64), ("instances_identity", c_uint64), ("rays", c_uint64)]

    def algorithmic_bytes(self, shadow=False):
        """24 B ray + 16 B hit (4 B max_distance for shadow rays) + 80 B/node
        + 48 B/triangle + 52 B per transformed instance entry + 4 B per identity entry."""
        per_ray = 24 + (4 if shadow else 16)
        return per_ray * self.rays + 80 * self.nodes + 48 * self.triangles + 52 * self.instances_transformed + 4 * self.instances_identity


class OracleFrame(Structure):
    _fields_ = [("framebuffer", c_void_p * RT_AOV_COUNT), ("accumulator", c_void_p * RT_AOV_COUNT), ("final_image", c_void_p),
                ("gbuffer_normal_and_depth", c_void_p), ("gbuffer_mesh_id_and_triangle_id", c_void_p), ("gbuffer_screen_position_prev", c_void_p),
                ("frame_buffer_moment", c_void_p), ("history_length", c_void_p),
                ("history_direct", c_void_p), ("history_indirect", c_void_p), ("history_moment", c_void_p), ("history_normal_and_depth", c_void_p),
                ("taa_frame_prev", c_void_p), ("taa_frame_curr", c_void_p), ("scratch_direct", c_void_p), ("scratch_indirect", c_void_p)]


class OracleCounters(Structure):
    _fields_ = [("trace", c_int32 * RT_MAX_BOUNCES), ("shadow", c_int32 * RT_MAX_BOUNCES),
                ("diffuse", c_int32 * RT_MAX_BOUNCES), ("plastic", c_int32 * RT_MAX_BOUNCES),
                ("dielectric", c_int32 * RT_MAX_BOUNCES), ("conductor", c_int32 * RT_MAX_BOUNCES),
                ("trace_stats", TraceStats), ("shadow_stats", TraceStats)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB_PATH):
            raise RuntimeError("%s missing: run `make -C oracle` (build() does)" % ORACLE_LIB_PATH)
        l = ctypes.CDLL(ORACLE_LIB_PATH)
        l.oracle_trace.argtypes = [POINTER(OracleScene)] + [c_void_p] * 6 + [c_size_t, c_void_p, POINTER(TraceStats), c_int]
        l.oracle_trace_shadow.argtypes = [POINTER(OracleScene)] + [c_void_p] * 7 + [c_size_t, c_void_p, POINTER(TraceStats), c_int]
        l.oracle_generate.argtypes = [POINTER(OracleScene), c_int, c_int, c_int] + [c_void_p] * 7
        l.oracle_random.argtypes = [POINTER(OracleScene), c_int, c_void_p, c_size_t, c_uint32, c_uint32, c_void_p]
        l.oracle_render_sample.argtypes = [POINTER(OracleScene), POINTER(OracleFrame), c_int, c_int, c_int, POINTER(OracleCounters), c_int]
        l.oracle_render_sample_unfiltered.argtypes = [POINTER(OracleScene), POINTER(OracleFrame), c_int, c_int, c_int, POINTER(OracleCounters), c_int]
        l.oracle_filter_frame.argtypes = [POINTER(OracleScene), POINTER(OracleFrame), c_int]
        l.oracle_render_ao_sample.argtypes = [POINTER(OracleScene), POINTER(OracleFrame), c_int, c_float, c_int, c_int, POINTER(OracleCounters), c_int]
        l.oracle_integrate_dielectric_cells.argtypes = [POINTER(OracleScene), c_int, c_int, c_int, c_int, c_void_p, c_int]
        l.oracle_integrate_conductor_cells.argtypes = [POINTER(OracleScene), c_int, c_int, c_int, c_void_p, c_int]
        l.oracle_average_dielectric.argtypes = [c_void_p, c_void_p]
        l.oracle_average_conductor.argtypes = [c_void_p, c_void_p]
        _lib = l
    return _lib


def ref_lib():
    """The reference's own BVH builder (oracle/_ref), or None where it has not been built."""
    global _ref
    if _ref is None and os.path.exists(REF_LIB_PATH):
        r = ctypes.CDLL(REF_LIB_PATH)
        r.ref_bvh_build_triangles.restype = c_void_p
        r.ref_bvh_build_triangles.argtypes = [c_void_p, c_int]
        r.ref_bvh_build_meshes.restype = c_void_p
        r.ref_bvh_build_meshes.argtypes = [c_void_p, c_int]
        for f in ("ref_bvh2_node_count", "ref_bvh2_index_count", "ref_bvh8_node_count", "ref_bvh8_index_count", "ref_bvh_free"):
            getattr(r, f).argtypes = [c_void_p]
        for f in ("ref_bvh2_copy_nodes", "ref_bvh2_copy_indices", "ref_bvh8_copy_nodes", "ref_bvh8_copy_indices"):
            getattr(r, f).argtypes = [c_void_p, c_void_p]
        if hasattr(r, "ref_bvh4_node_count"):
            r.ref_bvh4_node_count.argtypes = [c_void_p]
            r.ref_bvh4_copy_nodes.argtypes = [c_void_p, c_void_p]
        if hasattr(r, "ref_mipmap_downsample"):
            r.ref_mipmap_downsample.argtypes = [c_int] * 5 + [c_void_p, c_void_p]
        if hasattr(r, "ref_stbi_load_rgba"):
            r.ref_stbi_load_rgba.argtypes = [ctypes.c_char_p, POINTER(c_int), POINTER(c_int), c_void_p, ctypes.c_size_t]
            r.ref_stb_compress_bc1_block.argtypes = [c_void_p, c_void_p]
        if hasattr(r, "ref_bvh_build_binary_variant"):
            r.ref_bvh_build_binary_variant.restype = c_void_p
            r.ref_bvh_build_binary_variant.argtypes = [c_void_p, c_int, c_int, c_int, ctypes.c_float]
        for f in ("ref_bvh_ms_bvh2", "ref_bvh_ms_bvh8"):
            getattr(r, f).argtypes = [c_void_p]
            getattr(r, f).restype = ctypes.c_double
        _ref = r
    return _ref


def ref_build(tris24):
    """Reference BVH2 + BVH8 over (n,24) float32 triangles -> dict of byte/int arrays + timings."""
    r = ref_lib()
    t = np.ascontiguousarray(tris24, dtype=np.float32)
    n = t.size // 24
    h = r.ref_bvh_build_triangles(t.ctypes.data, n)
    out = {}
    n2, n8 = r.ref_bvh2_node_count(h), r.ref_bvh8_node_count(h)
    out["bvh2_nodes"] = np.zeros(n2 * 32, np.uint8); r.ref_bvh2_copy_nodes(h, out["bvh2_nodes"].ctypes.data)
    out["bvh8_nodes"] = np.zeros(n8 * 80, np.uint8); r.ref_bvh8_copy_nodes(h, out["bvh8_nodes"].ctypes.data)
    out["bvh2_indices"] = np.zeros(r.ref_bvh2_index_count(h), np.int32); r.ref_bvh2_copy_indices(h, out["bvh2_indices"].ctypes.data)
    out["bvh8_indices"] = np.zeros(r.ref_bvh8_index_count(h), np.int32); r.ref_bvh8_copy_indices(h, out["bvh8_indices"].ctypes.data)
    if hasattr(r, "ref_bvh4_node_count"):
        out["bvh4_nodes"] = np.zeros(r.ref_bvh4_node_count(h) * 128, np.uint8); r.ref_bvh4_copy_nodes(h, out["bvh4_nodes"].ctypes.data)
    out["ms_bvh2"], out["ms_bvh8"] = r.ref_bvh_ms_bvh2(h), r.ref_bvh_ms_bvh8(h)
    r.ref_bvh_free(h)
    return out


def ref_build_many(meshes, threads):
    """The reference's load-time BVH schedule: one job per mesh (SAH BVH2 + BVH8 conversion, the reference's own code) on
    `threads` workers, as AssetManager's thread pool runs it (Assets/AssetManager.cpp:57).
    meshes: list of (n, 24) float32 triangle arrays. Returns (wall ms, BVH2 nodes, BVH8 nodes), or None where oracle/_ref
    was built without it."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_bvh_build_many"):
        return None
    arrays = [np.ascontiguousarray(m, dtype=np.float32) for m in meshes]
    pointers = (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
    counts = (ctypes.c_int * len(arrays))(*[a.size // 24 for a in arrays])
    totals = (ctypes.c_longlong * 2)()
    r.ref_bvh_build_many.restype = ctypes.c_double
    r.ref_bvh_build_many.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    wall = r.ref_bvh_build_many(pointers, counts, len(arrays), int(threads), totals)
    return float(wall), int(totals[0]), int(totals[1])


def tlas_build(transforms, local_boxes):
    """CPU restatement of the device TLAS build (oracle_tlas.cpp): (n, 12) instance matrices and (n, 6) object-space boxes
    in scene order -> (nodes as (count, 80) uint8, order[position] = scene index)."""
    t = np.ascontiguousarray(transforms, np.float32).reshape(-1, 12); b = np.ascontiguousarray(local_boxes, np.float32).reshape(-1, 6)
    n = t.shape[0]
    nodes = np.zeros((2 * n, 80), np.uint8); order = np.zeros(n, np.int32)
    l = lib()
    l.oracle_tlas_build.restype = ctypes.c_int
    l.oracle_tlas_build.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    count = l.oracle_tlas_build(t.ctypes.data, b.ctypes.data, n, nodes.ctypes.data, order.ctypes.data)
    return nodes[:count], order


def effective_parallelism(threads, seconds_per_run=0.25):
    """threads x (time of a fixed spin loop on one thread) / (time of `threads` such loops at once): how many cores the
    process really gets (BASELINE.md 3 asks for it next to the core count)."""
    l = lib()
    l.oracle_effective_parallelism.restype = ctypes.c_double
    l.oracle_effective_parallelism.argtypes = [ctypes.c_int, ctypes.c_double]
    return float(l.oracle_effective_parallelism(int(threads), float(seconds_per_run)))


def ref_mipmap_downsample(filter_type, src, w_dst, h_dst):
    """One mip step by the reference's own Mipmap::downsample; src (h, w, 4) float32 -> (h_dst, w_dst, 4)."""
    src = np.ascontiguousarray(src, np.float32)
    dst = np.zeros((h_dst, w_dst, 4), np.float32)
    ref_lib().ref_mipmap_downsample(filter_type, src.shape[1], src.shape[0], w_dst, h_dst, src.ctypes.data, dst.ctypes.data)
    return dst


class ReferenceFrame:
    """The reference's own device code (Src/CUDA/Pathtracer.cu, compiled verbatim for the host CPU into oracle/_ref,
    see oracle/ref/ref_cuda_harness.cpp) rendering the scene of a SceneView: same protocol as Frame."""

    def __init__(self, view):
        if view.scene.alias_mesh_ids:
            raise ValueError("the reference's kernels know nothing of flattened static geometry: build this scene with merge_static = 0")
        r = ref_lib()
        if r is None or not hasattr(r, "ref_cuda_frame_create"):
            raise RuntimeError("oracle/_ref was built without the CUDA-on-CPU harness")
        r.ref_cuda_frame_create.restype = c_void_p
        r.ref_cuda_frame_create.argtypes = [c_void_p]
        r.ref_cuda_frame_free.argtypes = [c_void_p]
        r.ref_cuda_render_sample.argtypes = [c_void_p, c_int, c_void_p]
        r.ref_cuda_read_frame.argtypes = [c_void_p, c_void_p]
        r.ref_cuda_read_aov.argtypes = [c_void_p, c_int, c_void_p]
        self.view = view                       # keeps the staged arrays alive
        self.handle = r.ref_cuda_frame_create(ctypes.addressof(view.scene))
        s = view.scene
        self.shape = (s.screen_height, s.screen_pitch, 4)

    def render_sample(self, sample_index):
        """One sample of the whole frame; returns {queue: per-bounce sizes} like the device counters."""
        counters = np.zeros((6, 128), np.int32)
        ref_lib().ref_cuda_render_sample(self.handle, sample_index, counters.ctypes.data)
        return dict(zip(("trace", "diffuse", "plastic", "dielectric", "conductor", "shadow"), counters))

    @property
    def final(self):
        out = np.zeros(self.shape, np.float32)
        ref_lib().ref_cuda_read_frame(self.handle, out.ctypes.data)
        return out

    def history_length(self):
        out = np.zeros(self.shape[:2], np.int32)
        ref_lib().ref_cuda_read_history_length.argtypes = [c_void_p, c_void_p]
        ref_lib().ref_cuda_read_history_length(self.handle, out.ctypes.data)
        return out

    def accumulator(self, aov):
        out = np.zeros(self.shape, np.float32)
        if not ref_lib().ref_cuda_read_aov(self.handle, aov, out.ctypes.data):
            raise KeyError("AOV %d is not enabled" % aov)
        return out

    def integrate_dielectric_cells(self, entering, first_cell, cell_count):
        out = np.zeros(cell_count, np.float32)
        f = ref_lib().ref_cuda_integrate_dielectric_cells; f.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p]
        f(self.handle, 1 if entering else 0, first_cell, cell_count, out.ctypes.data)
        return out

    def integrate_conductor_cells(self, first_cell, cell_count):
        out = np.zeros(cell_count, np.float32)
        f = ref_lib().ref_cuda_integrate_conductor_cells; f.argtypes = [c_void_p, c_int, c_int, c_void_p]
        f(self.handle, first_cell, cell_count, out.ctypes.data)
        return out

    @staticmethod
    def average_dielectric(directional):
        d = np.ascontiguousarray(directional, np.float32); out = np.zeros(256, np.float32)
        f = ref_lib().ref_cuda_average_dielectric; f.argtypes = [c_void_p, c_void_p]; f(d.ctypes.data, out.ctypes.data)
        return out

    @staticmethod
    def average_conductor(directional):
        d = np.ascontiguousarray(directional, np.float32); out = np.zeros(32, np.float32)
        f = ref_lib().ref_cuda_average_conductor; f.argtypes = [c_void_p, c_void_p]; f(d.ctypes.data, out.ctypes.data)
        return out

    def close(self):
        if self.handle:
            ref_lib().ref_cuda_frame_free(self.handle); self.handle = None


_ref_ao = None


def ref_ao_lib():
    """The reference's AO.cu compiled for the host (oracle/_ref/libref_ao.so), or None where it has not been built."""
    global _ref_ao
    path = os.path.join(os.path.dirname(REF_LIB_PATH), "libref_ao.so")
    if _ref_ao is None and os.path.exists(path):
        r = ctypes.CDLL(path)
        r.ref_ao_frame_create.restype = c_void_p
        r.ref_ao_frame_create.argtypes = [c_void_p]
        r.ref_ao_frame_free.argtypes = [c_void_p]
        r.ref_ao_render_sample.argtypes = [c_void_p, c_int, c_float, c_void_p]
        r.ref_ao_read_frame.argtypes = [c_void_p, c_void_p]
        _ref_ao = r
    return _ref_ao


class ReferenceAOFrame:
    """The reference's ambient-occlusion kernels (Src/CUDA/AO.cu, verbatim, on the host CPU) rendering a SceneView."""

    def __init__(self, view):
        if view.scene.alias_mesh_ids:
            raise ValueError("the reference's kernels know nothing of flattened static geometry: build this scene with merge_static = 0")
        if ref_ao_lib() is None:
            raise RuntimeError("oracle/_ref/libref_ao.so has not been built")
        self.view = view
        self.handle = ref_ao_lib().ref_ao_frame_create(ctypes.addressof(view.scene))
        self.shape = (view.scene.screen_height, view.scene.screen_pitch, 4)

    def render_ao_sample(self, sample_index, ao_radius=1.0):
        counters = np.zeros(2, np.int32)
        ref_ao_lib().ref_ao_render_sample(self.handle, sample_index, ao_radius, counters.ctypes.data)
        return int(counters[0]), int(counters[1])

    @property
    def final(self):
        out = np.zeros(self.shape, np.float32)
        ref_ao_lib().ref_ao_read_frame(self.handle, out.ctypes.data)
        return out

    def close(self):
        if self.handle:
            ref_ao_lib().ref_ao_frame_free(self.handle); self.handle = None


def ref_geometry_shape(shape, transform16, p0=(0, 0, 0), p1=(0, 0, 1), radius=1.0, detail=-1):
    """Triangles (n, 24) of a primitive shape by the reference's own Geometry.cpp."""
    r = ref_lib()
    r.ref_geometry_shape.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_int]
    t = np.ascontiguousarray(transform16, np.float32); a = np.asarray(p0, np.float32); b = np.asarray(p1, np.float32)
    n = r.ref_geometry_shape(shape, t.ctypes.data, a.ctypes.data, b.ctypes.data, radius, detail, None, 0)
    out = np.zeros((n, 24), np.float32)
    r.ref_geometry_shape(shape, t.ctypes.data, a.ctypes.data, b.ctypes.data, radius, detail, out.ctypes.data, n)
    return out


def ref_stbi_loadf(filename):
    """(h, w, 3) float32 as the reference's Sky::load gets it from stbi_loadf, or None."""
    r = ref_lib()
    r.ref_stbi_loadf_rgb.argtypes = [ctypes.c_char_p, POINTER(c_int), POINTER(c_int), c_void_p, ctypes.c_size_t]
    w, h = c_int(), c_int()
    n = r.ref_stbi_loadf_rgb(str(filename).encode(), ctypes.byref(w), ctypes.byref(h), None, 0)
    if n == 0:
        return None
    out = np.zeros((h.value, w.value, 3), np.float32)
    r.ref_stbi_loadf_rgb(str(filename).encode(), ctypes.byref(w), ctypes.byref(h), out.ctypes.data, out.size)
    return out


def ref_stbi_load(filename):
    """The file decoded by the reference's own stb_image (RGBA8, (h, w, 4), row 0 = top), or None."""
    r = ref_lib()
    w, h = c_int(), c_int()
    if not r.ref_stbi_load_rgba(str(filename).encode(), ctypes.byref(w), ctypes.byref(h), None, 0):
        return None
    out = np.zeros((h.value, w.value, 4), np.uint8)
    assert r.ref_stbi_load_rgba(str(filename).encode(), ctypes.byref(w), ctypes.byref(h), out.ctypes.data, out.nbytes)
    return out


def ref_build_binary_variant(tris24, sbvh, collapse, sbvh_alpha=10e-5):
    """The reference's device-side binary tree for bvh_type = BVH / SBVH (optionally leaf-collapsed
    like a file-loaded mesh) and the BVH4 converted from it."""
    r = ref_lib()
    t = np.ascontiguousarray(tris24, dtype=np.float32)
    h = r.ref_bvh_build_binary_variant(t.ctypes.data, t.size // 24, int(sbvh), int(collapse), sbvh_alpha)
    out = {}
    out["bvh2_nodes"] = np.zeros(r.ref_bvh2_node_count(h) * 32, np.uint8); r.ref_bvh2_copy_nodes(h, out["bvh2_nodes"].ctypes.data)
    out["bvh2_indices"] = np.zeros(r.ref_bvh2_index_count(h), np.int32); r.ref_bvh2_copy_indices(h, out["bvh2_indices"].ctypes.data)
    out["bvh4_nodes"] = np.zeros(r.ref_bvh4_node_count(h) * 128, np.uint8); r.ref_bvh4_copy_nodes(h, out["bvh4_nodes"].ctypes.data)
    out["ms_bvh2"] = r.ref_bvh_ms_bvh2(h)
    r.ref_bvh_free(h)
    return out


_ref_scene = None


def ref_scene_lib():
    """oracle/_ref/libref_scene.so: the reference's scene-loading side compiled verbatim (None where it was not built)."""
    global _ref_scene
    if _ref_scene is None:
        path = os.path.join(_HERE, "_ref", "libref_scene.so")
        if not os.path.exists(path):
            return None
        r = ctypes.CDLL(path)
        r.ref_scene_load.restype = ctypes.c_void_p
        r.ref_scene_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_int] * 4
        r.ref_scene_free.argtypes = [ctypes.c_void_p]
        r.ref_scene_description.restype = ctypes.c_char_p
        r.ref_scene_description.argtypes = [ctypes.c_void_p]
        r.ref_scene_triangles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        r.ref_scene_texture_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        r.ref_scene_texture_data.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        r.ref_scene_mesh_transform.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        r.ref_scene_sky.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        r.ref_load_mesh_file.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_float, ctypes.c_void_p]
        _ref_scene = r
    return _ref_scene


class ReferenceScene:
    """Scene::Scene + AssetManager::wait_until_loaded of the reference for one scene file. It writes `.bvh` caches next
    to the meshes it loads: point it at a scratch copy."""

    def __init__(self, filename, sky="", bvh_type=3, enable_block_compression=True, mipmap_filter=0, enable_mipmapping=True):
        self.lib = ref_scene_lib()
        self.h = self.lib.ref_scene_load(filename.encode(), sky.encode(), bvh_type, int(enable_block_compression), mipmap_filter, int(enable_mipmapping))
        self.description = self.lib.ref_scene_description(self.h).decode(errors="replace")

    def close(self):
        if self.h:
            self.lib.ref_scene_free(self.h); self.h = None

    def count(self, kind):
        return sum(1 for line in self.description.splitlines() if line.startswith(kind + " "))

    def triangles(self, mesh_data):
        n = self.lib.ref_scene_triangles(self.h, mesh_data, None)
        out = np.zeros((n, 24), np.float32)
        self.lib.ref_scene_triangles(self.h, mesh_data, out.ctypes.data)
        return out

    def texture(self, index):
        """-> dict(format 0..3 = BC1, BC2, BC3, RGBA; channels, width, height, mip_offsets (bytes), data)"""
        info = (ctypes.c_int * 6)()
        self.lib.ref_scene_texture_info(self.h, index, info)
        data = np.zeros(info[5], np.uint8); offsets = np.zeros(info[4], np.int32)
        self.lib.ref_scene_texture_data(self.h, index, data.ctypes.data, offsets.ctypes.data)
        return dict(format=info[0], channels=info[1], width=info[2], height=info[3], mip_offsets=offsets, data=data)

    def mesh_transform(self, mesh):
        out = np.zeros((3, 16), np.float32)
        self.lib.ref_scene_mesh_transform(self.h, mesh, out.ctypes.data)
        return out

    def sky(self):
        n = self.lib.ref_scene_sky(self.h, None)
        out = np.zeros((n, 4), np.float32)
        self.lib.ref_scene_sky(self.h, out.ctypes.data)
        return out


def ref_export_image(filename, image, pitch=None):
    """PPMExporter::save (.ppm, display-space values) / EXRExporter::save (.exr) of the reference; image: (h, w, 3) float32, row 0 at the bottom."""
    r = ref_scene_lib()
    h, w, _ = image.shape
    pitch = pitch or w
    buf = np.zeros((h, pitch, 3), np.float32); buf[:, :w] = image
    r.ref_export_image.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    r.ref_export_image(0 if str(filename).endswith(".ppm") else 1, str(filename).encode(), pitch, w, h, buf.ctypes.data)


def ref_args_parse(arguments):
    """Args::parse of the reference on fresh configurations -> one line in the form of `pathtracer --print-config`."""
    r = ref_scene_lib()
    argv = (ctypes.c_char_p * (len(arguments) + 1))(b"pathtracer", *[a.encode() for a in arguments])
    out = ctypes.create_string_buffer(4096)
    r.ref_args_parse.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_char_p, ctypes.c_int]
    r.ref_args_parse(len(arguments) + 1, argv, out, 4096)
    return out.value.decode(errors="replace")


def ref_load_mesh_file(kind, filename, arg=0.0):
    """The reference's mesh loaders on their own. kind: "obj", "ply", "serialized" (arg = shape index), "hair" (arg = radius)."""
    r = ref_scene_lib()
    k = ("obj", "ply", "serialized", "hair").index(kind)
    n = r.ref_load_mesh_file(k, filename.encode(), float(arg), None)
    out = np.zeros((n, 24), np.float32)
    r.ref_load_mesh_file(k, filename.encode(), float(arg), out.ctypes.data)
    return out


def ref_build_optimized(tris24, sbvh, max_batches):
    """BVH::create_from_triangles with enable_bvh_optimization, limited to `max_batches` (+1) batches, and the BVH8
    (SAH only) / BVH4 converted from the optimised tree."""
    r = ref_lib()
    t = np.ascontiguousarray(tris24, dtype=np.float32)
    r.ref_bvh_build_optimized.restype = ctypes.c_void_p
    r.ref_bvh_build_optimized.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    h = r.ref_bvh_build_optimized(t.ctypes.data, t.size // 24, int(sbvh), int(max_batches))
    out = {}
    out["bvh2_nodes"] = np.zeros(r.ref_bvh2_node_count(h) * 32, np.uint8); r.ref_bvh2_copy_nodes(h, out["bvh2_nodes"].ctypes.data)
    out["bvh2_indices"] = np.zeros(r.ref_bvh2_index_count(h), np.int32); r.ref_bvh2_copy_indices(h, out["bvh2_indices"].ctypes.data)
    out["bvh4_nodes"] = np.zeros(r.ref_bvh4_node_count(h) * 128, np.uint8); r.ref_bvh4_copy_nodes(h, out["bvh4_nodes"].ctypes.data)
    if not sbvh:
        out["bvh8_nodes"] = np.zeros(r.ref_bvh8_node_count(h) * 80, np.uint8); r.ref_bvh8_copy_nodes(h, out["bvh8_nodes"].ctypes.data)
        out["bvh8_indices"] = np.zeros(r.ref_bvh8_index_count(h), np.int32); r.ref_bvh8_copy_indices(h, out["bvh8_indices"].ctypes.data)
    out["ms_bvh2"] = r.ref_bvh_ms_bvh2(h)
    r.ref_bvh_free(h)
    return out


class SceneView:
    """Keeps the numpy arrays alive that an OracleScene points into."""

    def __init__(self, pathtracer, bvh_type=8, luts=None):
        pt = pathtracer
        self.keep = {}
        s = OracleScene()

        def arr(name):
            a = pt.array(name)
            self.keep[name] = a
            return a.ctypes.data if a.size else None

        s.triangles = arr("triangles"); s.triangle_count = self.keep["triangles"].size // 24
        s.bvh8_nodes = arr("bvh8_nodes"); s.bvh2_nodes = arr("bvh2_nodes"); s.bvh4_nodes = arr("bvh4_nodes"); s.bvh_type = bvh_type
        s.mesh_bvh_root_indices = arr("mesh_bvh_root_indices"); s.mesh_material_ids = arr("mesh_material_ids")
        s.mesh_transforms = arr("mesh_transforms"); s.mesh_transforms_inv = arr("mesh_transforms_inv"); s.mesh_transforms_prev = arr("mesh_transforms_prev")
        s.mesh_count = self.keep["mesh_material_ids"].size
        s.alias_mesh_ids = arr("alias_mesh_ids"); s.alias_triangle_ids = arr("alias_triangle_ids")   # flattened static geometry, if any
        s.static_whole_scene = 1 if pt.static_geometry_whole_scene else 0
        s.skip_behind_hit = 1 if pt.skip_behind_hit else 0   # the walk the device takes (rt_set_skip_behind_hit)
        s.material_types = arr("material_types"); s.materials = arr("materials"); s.material_count = self.keep["material_types"].size
        s.media = arr("media"); s.medium_count = self.keep["media"].size // 8

        tex = pt.textures()
        self.keep["tex"] = tex
        table = (OracleTexture * max(len(tex), 1))()
        for i, (texels, w, h, levels) in enumerate(tex):
            table[i].texels = texels.ctypes.data; table[i].width = w; table[i].height = h; table[i].mip_levels = levels
            table[i].lod_width, table[i].lod_height = pt.texture_lod_size(i)
        self.keep["tex_table"] = table
        s.textures = ctypes.cast(table, c_void_p); s.texture_count = len(tex)

        s.light_triangle_indices = arr("light_triangle_indices"); s.light_triangle_cumulative_probability = arr("light_triangle_cumulative_probability")
        s.light_triangle_count = self.keep["light_triangle_indices"].size
        s.light_mesh_cumulative_probability = arr("light_mesh_cumulative_probability"); s.light_mesh_triangle_span = arr("light_mesh_triangle_span")
        s.light_mesh_transform_indices = arr("light_mesh_transform_indices"); s.light_mesh_count = self.keep["light_mesh_transform_indices"].size
        s.lights_total_weight = pt.lights_total_weight
        s.pmj_samples = arr("pmj_samples"); s.blue_noise = arr("blue_noise")

        if luts is not None:
            self.keep["luts"] = [np.ascontiguousarray(l, dtype=np.float32) for l in luts]
            (s.lut_dielectric_directional_albedo_enter, s.lut_dielectric_directional_albedo_leave, s.lut_dielectric_albedo_enter,
             s.lut_dielectric_albedo_leave, s.lut_conductor_directional_albedo, s.lut_conductor_albedo) = [l.ctypes.data for l in self.keep["luts"]]

        sky, w, h, scale = pt.sky()
        self.keep["sky"] = sky
        s.sky = sky.ctypes.data; s.sky_width = w; s.sky_height = h; s.sky_scale = scale

        cam = pt.camera()
        ctypes.memmove(byref(s.camera), byref(cam), ctypes.sizeof(cam))
        cfg = pt.device_config()
        ctypes.memmove(byref(s.config), byref(cfg), ctypes.sizeof(cfg))
        s.config.aov_mask |= 1
        s.screen_width = pt.width; s.screen_height = pt.height; s.screen_pitch = pt.pitch
        self.scene = s

    def trace(self, origin, direction, threads=0):
        o = np.ascontiguousarray(origin, np.float32); d = np.ascontiguousarray(direction, np.float32)
        n = o.shape[1]
        hits = np.zeros((n, 4), np.uint32)
        stats = TraceStats()
        lib().oracle_trace(byref(self.scene), o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, d[0].ctypes.data, d[1].ctypes.data, d[2].ctypes.data, n, hits.ctypes.data, byref(stats), threads)
        return hits, stats

    def trace_shadow(self, origin, direction, max_distance, threads=0):
        o = np.ascontiguousarray(origin, np.float32); d = np.ascontiguousarray(direction, np.float32); m = np.ascontiguousarray(max_distance, np.float32)
        n = o.shape[1]
        occ = np.zeros(n, np.uint8)
        stats = TraceStats()
        lib().oracle_trace_shadow(byref(self.scene), o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, d[0].ctypes.data, d[1].ctypes.data, d[2].ctypes.data, m.ctypes.data, n, occ.ctypes.data, byref(stats), threads)
        return occ, stats

    def generate(self, sample_index, pixel_offset, pixel_count):
        o = np.zeros((3, pixel_count), np.float32); d = np.zeros((3, pixel_count), np.float32); px = np.zeros(pixel_count, np.uint32)
        lib().oracle_generate(byref(self.scene), sample_index, pixel_offset, pixel_count, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data, d[0].ctypes.data, d[1].ctypes.data, d[2].ctypes.data, px.ctypes.data)
        return o, d, px

    def integrate_dielectric_cells(self, entering, first_cell, cell_count, num_samples=100000, threads=0):
        out = np.zeros(cell_count, np.float32)
        lib().oracle_integrate_dielectric_cells(byref(self.scene), 1 if entering else 0, num_samples, first_cell, cell_count, out.ctypes.data, threads)
        return out

    def integrate_conductor_cells(self, first_cell, cell_count, num_samples=100000, threads=0):
        out = np.zeros(cell_count, np.float32)
        lib().oracle_integrate_conductor_cells(byref(self.scene), num_samples, first_cell, cell_count, out.ctypes.data, threads)
        return out

    def random(self, dimension, pixel_indices, bounce, sample_index):
        px = np.ascontiguousarray(pixel_indices, np.uint32)
        out = np.zeros((px.size, 2), np.float32)
        lib().oracle_random(byref(self.scene), dimension, px.ctypes.data, px.size, bounce, sample_index, out.ctypes.data)
        return out


def average_dielectric(directional):
    d = np.ascontiguousarray(directional, np.float32); out = np.zeros(256, np.float32)
    lib().oracle_average_dielectric(d.ctypes.data, out.ctypes.data)
    return out


def average_conductor(directional):
    d = np.ascontiguousarray(directional, np.float32); out = np.zeros(32, np.float32)
    lib().oracle_average_conductor(d.ctypes.data, out.ctypes.data)
    return out


class Frame:
    """AOV + SVGF state for oracle_render_sample, zero-initialised (the device memsets too)."""

    def __init__(self, view):
        s = view.scene
        n = s.screen_pitch * s.screen_height
        self.view = view
        self.f = OracleFrame()
        self.buffers = {}
        mask = s.config.aov_mask
        if s.config.enable_svgf:
            mask |= 0b1110
        for i in range(RT_AOV_COUNT):
            if (mask >> i) & 1:
                fb = np.zeros((n, 4), np.float32); acc = np.zeros((n, 4), np.float32)
                self.buffers["fb%d" % i] = fb; self.buffers["acc%d" % i] = acc
                self.f.framebuffer[i] = fb.ctypes.data; self.f.accumulator[i] = acc.ctypes.data
        self.final = np.zeros((s.screen_height, s.screen_pitch, 4), np.float32)
        self.f.final_image = self.final.ctypes.data
        if s.config.enable_svgf:
            def mk(name, shape, dtype=np.float32):
                a = np.zeros(shape, dtype); self.buffers[name] = a; return a.ctypes.data
            self.f.gbuffer_normal_and_depth = mk("gnd", (n, 4)); self.f.gbuffer_mesh_id_and_triangle_id = mk("gid", (n, 2), np.int32)
            self.f.gbuffer_screen_position_prev = mk("gsp", (n, 2)); self.f.frame_buffer_moment = mk("mom", (n, 4))
            self.f.history_length = mk("hl", (n,), np.int32)
            self.f.history_direct = mk("hd", (n, 4)); self.f.history_indirect = mk("hi", (n, 4)); self.f.history_moment = mk("hm", (n, 4))
            self.f.history_normal_and_depth = mk("hnd", (n, 4)); self.f.taa_frame_prev = mk("tp", (n, 4)); self.f.taa_frame_curr = mk("tc", (n, 4))

    def render_sample(self, sample_index, pixel_offset=0, pixel_count=None, threads=0):
        s = self.view.scene
        if pixel_count is None:
            pixel_count = s.screen_width * s.screen_height - pixel_offset
        counters = OracleCounters()
        lib().oracle_render_sample(byref(s), byref(self.f), sample_index, pixel_offset, pixel_count, byref(counters), threads)
        return counters

    def render_sample_unfiltered(self, sample_index, pixel_offset=0, pixel_count=None, threads=0):
        """SVGF frame, first half: path-trace a pixel range into the per-frame AOVs and g-buffers; no filter, nothing cleared."""
        s = self.view.scene
        if pixel_count is None:
            pixel_count = s.screen_width * s.screen_height - pixel_offset
        counters = OracleCounters()
        lib().oracle_render_sample_unfiltered(byref(s), byref(self.f), sample_index, pixel_offset, pixel_count, byref(counters), threads)
        return counters

    def filter_frame(self, sample_index):
        """SVGF frame, second half: reproject / variance / a-trous / finalize / TAA over the whole frame, then clear the AOVs."""
        lib().oracle_filter_frame(byref(self.view.scene), byref(self.f), sample_index)

    def svgf_inputs(self):
        """The arrays an SVGF frame's filter reads that the path tracing of THIS frame wrote: name -> (pitch * height, C) array."""
        return {"direct": self.buffers["fb%d" % RT_AOV_RADIANCE_DIRECT], "indirect": self.buffers["fb%d" % RT_AOV_RADIANCE_INDIRECT], "albedo": self.buffers["fb%d" % RT_AOV_ALBEDO],
                "normal_and_depth": self.buffers["gnd"], "mesh_and_triangle": self.buffers["gid"], "screen_position_prev": self.buffers["gsp"]}

    def render_ao_sample(self, sample_index, ao_radius=1.0, pixel_offset=0, pixel_count=None, threads=0):
        """AO::render (Integrators/AO.cpp:148-200) for one sample."""
        s = self.view.scene
        if pixel_count is None:
            pixel_count = s.screen_width * s.screen_height - pixel_offset
        counters = OracleCounters()
        lib().oracle_render_ao_sample(byref(s), byref(self.f), sample_index, float(ao_radius), pixel_offset, pixel_count, byref(counters), threads)
        return counters

    def accumulator(self, aov):
        s = self.view.scene
        return self.buffers["acc%d" % aov].reshape(s.screen_height, s.screen_pitch, 4)
