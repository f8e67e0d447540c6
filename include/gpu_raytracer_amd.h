/*
 * gpu_raytracer_amd.h -- C ABI of the MI355X-native wavefront path tracer device layer.
 *
 * This is the drop-in boundary for the hot path of jan-van-bergen/GPU-Raytracer
 * (ray-generate -> BVH8/CWBVH trace -> sort -> shade/NEE -> shadow trace ->
 * accumulate | SVGF/TAA).  The reference has no FFI: its host class `Integrator`
 * talks to the device code by *name* -- 23 `extern "C" __global__` kernels found
 * with cuModuleGetFunction (Src/Renderer/Integrators/Pathtracer.cpp:79-101) and
 * ~55 `__device__` globals found with cuModuleGetGlobal (`get_global("...")`,
 * Integrator.cpp / Pathtracer.cpp).  Every entry point below replaces one group of
 * those by-name bindings; the comment on each one cites what it replaces.
 *
 * Conventions: plain pointers and sizes only (no C++/torch types); one opaque
 * context per GPU; every function returns RT_OK (0) or a negative rt_status and
 * records a message retrievable with rt_last_error(); host pointers unless the
 * name says `_device`; all work is enqueued on the context's HIP stream and is
 * complete when the call returns only where stated ("synchronous").
 */
#ifndef GPU_RAYTRACER_AMD_H
#define GPU_RAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants shared with the reference (Src/CUDA/Common.h) -------------------- */
#define RT_MAX_BOUNCES                  128   /* Common.h:75  MAX_BOUNCES            */
#define RT_BATCH_SIZE                   (1080 * 720) /* Common.h:71 BATCH_SIZE       */
#define RT_PMJ_NUM_SEQUENCES            64    /* Common.h:79                         */
#define RT_PMJ_NUM_SAMPLES_PER_SEQUENCE 4096  /* Common.h:80                         */
#define RT_BLUE_NOISE_NUM_TEXTURES      16    /* Common.h:82                         */
#define RT_BLUE_NOISE_TEXTURE_DIM       128   /* Common.h:83                         */
#define RT_MAX_ATROUS_ITERATIONS        10    /* Common.h:99                         */
#define RT_LUT_DIELECTRIC_DIM           16    /* Common.h:87-89 (ior, roughness, cos)*/
#define RT_LUT_CONDUCTOR_DIM            32    /* Common.h:94-95                      */
#define RT_INVALID                      (-1)

typedef enum rt_status {
	RT_OK                 =  0,
	RT_ERROR_INVALID_ARG  = -1,
	RT_ERROR_NO_DEVICE    = -2,   /* no HIP device / hipSetDevice failed               */
	RT_ERROR_HIP          = -3,   /* a HIP runtime call failed (message has the detail)*/
	RT_ERROR_NOT_READY    = -4,   /* render before the scene/tables were uploaded      */
	RT_ERROR_OUT_OF_RANGE = -5
} rt_status;

/* AOVType, Common.h:28-37 */
typedef enum rt_aov_type {
	RT_AOV_RADIANCE = 0, RT_AOV_RADIANCE_DIRECT, RT_AOV_RADIANCE_INDIRECT,
	RT_AOV_ALBEDO, RT_AOV_NORMAL, RT_AOV_POSITION, RT_AOV_COUNT
} rt_aov_type;

/* ReconstructionFilter, Common.h:21-25 */
typedef enum rt_filter { RT_FILTER_BOX = 0, RT_FILTER_TENT, RT_FILTER_GAUSSIAN } rt_filter;

/* MaterialType, CUDA/Material.h:13-19 (one byte per material on the device) */
typedef enum rt_material_type {
	RT_MATERIAL_LIGHT = 0, RT_MATERIAL_DIFFUSE, RT_MATERIAL_PLASTIC, RT_MATERIAL_DIELECTRIC, RT_MATERIAL_CONDUCTOR
} rt_material_type;

/* GPUConfig, Common.h:39-67 -- replaces the `config` device constant (Integrator.cpp:521). */
typedef struct rt_gpu_config {
	int32_t  reconstruction_filter;            /* rt_filter                               */
	uint32_t aov_mask;                         /* bit i = AOV i enabled                   */
	int32_t  num_bounces;
	int32_t  enable_mipmapping;
	int32_t  enable_next_event_estimation;
	int32_t  enable_multiple_importance_sampling;
	int32_t  enable_russian_roulette;
	int32_t  enable_svgf;
	int32_t  enable_spatial_variance;
	int32_t  enable_taa;
	float    alpha_colour;
	float    alpha_moment;
	int32_t  num_atrous_iterations;
	float    sigma_z;
	float    sigma_n;
	float    sigma_l;
} rt_gpu_config;

/* Camera, CUDA/Camera.h:10-18 -- replaces the `camera` constant (Integrator.cpp:454-481). */
typedef struct rt_camera {
	float position[3];
	float bottom_left_corner[3];
	float x_axis[3];
	float y_axis[3];
	float pixel_spread_angle;
	float aperture_radius;
	float focal_distance;
} rt_camera;

enum { RT_TEXTURE_RGBA8 = 0, RT_TEXTURE_BC1 = 1 };

/* One mip-mapped RGBA8 texture in LINEAR light, i.e. what the reference hands to
 * cuMipmappedArray before optional BC1 compression (Assets/TextureLoader.cpp:145-206).
 * CDNA compute parts have no texture units, so wrap addressing and bi/tri-linear
 * filtering are done in the shade kernel on these texels.                             */
typedef struct rt_texture_desc {
	const uint8_t * texels;      /* all mip levels back to back, level 0 first, 4 B/texel */
	int32_t         width, height; /* level 0 size; level l is max(w>>l,1) x max(h>>l,1)  */
	int32_t         mip_levels;
	/* Size that enters lod_bias = 0.5 * log2(w * h) (Integrator.cpp:95); 0 = width / height. The reference
	 * overwrites Texture::width/height with the BLOCK counts when it BC1-compresses a texture
	 * (TextureLoader.cpp:256-258), so compressed textures carry a bias that is 2 lower.                  */
	int32_t         lod_width, lod_height;
	/* RT_TEXTURE_RGBA8: `texels` as described above. RT_TEXTURE_BC1: `texels` points at the BC1 blocks the reference
	 * hands to its texture unit (TextureLoader.cpp:208-262): 8 bytes per 4x4 texels, ((w+3)/4) x ((h+3)/4) blocks per
	 * level in row-major order, levels back to back; the shade kernels decode a block per texel fetch (D3D rules:
	 * bit-replicated end points, thirds rounded to nearest, 3-colour + transparent mode when c0 <= c1).            */
	int32_t         format;
	int32_t         reserved;
} rt_texture_desc;

/* Per-stage counters of the last rt_render_sample, replaces the read-back of
 * `buffer_sizes` (Pathtracer.cpp:845-847) and the CUDAEventPool timings.              */
typedef struct rt_counters {
	int32_t trace     [RT_MAX_BOUNCES];        /* rays traced per bounce (closest hit)    */
	int32_t shadow    [RT_MAX_BOUNCES];        /* shadow rays traced per bounce           */
	int32_t diffuse   [RT_MAX_BOUNCES];
	int32_t plastic   [RT_MAX_BOUNCES];
	int32_t dielectric[RT_MAX_BOUNCES];
	int32_t conductor [RT_MAX_BOUNCES];
	float   ms_generate, ms_trace, ms_sort, ms_shade, ms_shadow, ms_post; /* HIP-event sums */
	float   ms_total;
} rt_counters;

typedef struct rt_context rt_context;

/* ---- lifetime: replaces CUDAContext::init / Integrator::cuda_init / cuda_free --------
 * (Device/CUDAContext.cpp:21, Pathtracer.cpp:9-41, Integrator.h:203-229)                */
int          rt_create (int device_ordinal, rt_context ** out_ctx);
void         rt_destroy(rt_context * ctx);
const char * rt_last_error(const rt_context * ctx);      /* ctx may be NULL: last global error */
const char * rt_version(void);
/* Layout version of the structs that cross this boundary. Bumped whenever one changes size or meaning, so that a client
 * built against an older header fails its start-up check instead of passing arrays with the wrong stride:
 *   1  round 1
 *   2  rt_texture_desc grew `format`, `lod_width`, `lod_height` (32 -> 40 bytes; rt_upload_textures rejects unknown formats)
 *   3  RT_TIMING_* kinds of rt_get_launch_timings, rt_comm_* / rt_all_gather_* entry points (additions only)
 *   4  rt_upload_triangle_aliases, rt_set_static_geometry (additions only)
 *   5  rt_set_texture_expansion, rt_texture_bytes (additions only; BC1 textures are decoded at upload unless asked otherwise)
 *   6  rt_set_svgf_tiles, rt_set_stream_batch (additions only; 5 and 6 also carried rt_set_node_format / rt_set_node_cache)
 *   7  rt_set_node_format (a 96-byte decoded copy of the node array) and rt_set_node_cache (the top of the flattened tree in LDS) REMOVED:
 *      both measured slower than the 80-byte walk on MI355X (profiles/r04_traversal_experiments.txt items 2 and 4) and were off by default;
 *      rt_set_build_boxes added
 *   8  rt_set_skip_behind_hit, rt_get_skip_behind_hit, rt_geometry_fits_flat_engine, rt_update_nodes (additions only)
 * Check `rt_abi_version() == RT_ABI_VERSION` once after loading the library.                                          */
#define RT_ABI_VERSION 8
int rt_abi_version(void);

/* ---- scene upload ------------------------------------------------------------------- */
/* Replaces globals `triangles` and `bvh8_nodes` (Integrator.cpp:153-154,268-269).
 * triangles: triangle_count x 96 B in the device layout CUDA/Raytracing/Triangle.h:4-11
 * (position_0, edge_1, edge_2, normal_0, n_edge_1, n_edge_2, uv_0, uv_edge_1, uv_edge_2),
 * already permuted by the BLAS indices.  bvh8_nodes: node_count x 80 B CWBVH nodes
 * (BVH/BVH.h:61-80); slots [0, 2*mesh_count) are reserved for the TLAS.                 */
int rt_upload_geometry(rt_context * ctx, const void * triangles, size_t triangle_count,
                       const void * bvh8_nodes, size_t node_count);
/* 1 when a one-tree scene of this size (rt_set_static_geometry(ctx, 1)) is walked by the flattened scene's engine, whose node offsets are a 24-bit multiply and
 * whose triangle offsets are 32 bits wide: fewer than 2^24 nodes and less than 4 GiB of 48-byte triangle records. Larger scenes are walked by the general
 * engine (64-bit addresses), the reference's way. A pure function: no context, no device.                                                             */
int rt_geometry_fits_flat_engine(size_t node_count, size_t triangle_count);
/* Static geometry flattened into ONE bottom-level tree (no counterpart in the reference, which traverses one BLAS per mesh
 * under the TLAS whether the meshes move or not -- Integrator.cpp:101-283,399-430): the caller adds COPIES of the triangles
 * of its static instances to the triangle array, builds (or lets rt_build_geometry build) one more CWBVH over the copies and
 * gives that tree one TLAS leaf. mesh_ids / triangle_ids have one entry per triangle of the uploaded array: mesh_ids[i] < 0
 * for an ordinary triangle; for a copy, a closest hit on it is reported as instance mesh_ids[i] (a row of the
 * rt_upload_instances tables, which may lie behind the rows the TLAS references) and triangle triangle_ids[i] (an ordinary
 * triangle of the array), so that everything behind traversal -- materials, transforms, light tables, the SVGF ids -- sees
 * the instances the scene names. Call after rt_upload_geometry / rt_build_geometry, which forget the aliases; NULL arrays
 * clear them. CWBVH traversal only.                                                                                  */
int rt_upload_triangle_aliases(rt_context * ctx, const int32_t * mesh_ids, const int32_t * triangle_ids);
/* ... and when NOTHING is left outside that tree (no instance moves), the TLAS has nothing to decide: with whole_scene = 1 the
 * caller uploads the ROOT NODE OF THE FLATTENED TREE as the one "TLAS" node (rt_upload_tlas, node slot 0; its child and
 * triangle indices are absolute, so the copy works from there) and row 0 of rt_upload_instances describes the tree (identity
 * transform); every ray then starts INSIDE the tree -- one node step and one instance entry less per ray. whole_scene = 0
 * (the state after every geometry upload): node 0 is a TLAS root, as in the reference (BVH8.h:161-165), and a flattened tree
 * is one of its leaves. Drains the context when the value changes. CWBVH traversal only.                              */
int rt_set_static_geometry(rt_context * ctx, int32_t whole_scene);
/* Replaces nodes [first_node, first_node + node_count) of the uploaded CWBVH node array (80 bytes each) in place, between frames: for a tree whose children were
 * given other octant slots beside the frame loop (the host re-seats the flattened tree when the camera has travelled: Integrator.cpp, reseat worker). The
 * caller keeps node count, boxes, leaves and every index as they were; the context is drained first (a ray's stack entries are only valid within one launch). */
int rt_update_nodes(rt_context * ctx, const void * nodes, size_t first_node, size_t node_count);
/* The walk of closest-hit rays through a one-tree scene (rt_set_static_geometry(ctx, 1)). The reference keeps the children of a node that a ray
 * enters but does not visit at once as ONE stack entry without a distance (BVH8.h:166-199) and therefore still walks into every one of them after a
 * hit in front of them has been found; it makes up for that with 32-lane warps and triangle postponing (BVH8.h:200,234-240). enable = 1 (the
 * default): a stack entry also carries a 16-bit lower bound of the distance at which the ray enters any child left in it (the mask word's unused
 * bits 8..23) and is dropped at its pop when that bound is not in front of the hit held -- a visit that could enter no child. Closest hits are the
 * reference walk's (ties in t between coplanar triangles aside: the walk's order decides those, as it does in the reference); the number of nodes a
 * ray fetches falls (Sponza: 13.1 -> 11.5 per ray). enable = 0: the reference's walk, node for node. Scenes that keep a TLAS always walk the
 * reference's way. Drains the context when the value changes. rt_get_skip_behind_hit: 1 while the walk is in effect (wish AND one-tree scene).  */
int rt_set_skip_behind_hit(rt_context * ctx, int32_t enable);
int rt_get_skip_behind_hit(const rt_context * ctx);
/* Replaces the per-frame TLAS memcpy into the front of `bvh8_nodes` (Integrator.cpp:404-409). The
 * TLAS, the instance tables (rt_upload_instances) and the light tables (rt_upload_lights) are
 * versioned on the device: the call copies the host data into pinned staging and returns (the
 * caller's buffers are free again), the device copy is asynchronous, samples already in flight keep
 * the version they were submitted with and later rt_render_sample calls see the new one. The GPU is
 * only drained when a table outgrows its ring.                                                   */
/* Builds the CWBVH of every mesh ON THE DEVICE instead of uploading host-built nodes (replaces rt_upload_geometry; reference:
 * SAHBuilder.cpp:13-104 + BVH8Converter.cpp:7-335 run per mesh on host threads at load time). `triangles` are the 96-byte
 * device triangles of all meshes back to back in any order, mesh m owning [mesh_first_triangle[m], mesh_first_triangle[m + 1]).
 * The device sorts each mesh's triangles along a Morton curve, builds 8-wide compressed nodes over the sorted order (<= 3
 * triangles per leaf) and stores the triangles in leaf order: out_triangle_positions[i] is where input triangle i went (the
 * caller remaps whatever names triangles by index: light tables), out_root_indices[m] the root node of mesh m (node slots
 * [0, reserved_tlas_nodes) stay free for the TLAS). A linear BVH: built in a fraction of the host build's time, dearer to
 * traverse than the SAH tree; closest hits are the same. rt_read_geometry returns what was built (host views, checker).   */
int rt_build_geometry(rt_context * ctx, const void * triangles, size_t triangle_count, const int32_t * mesh_first_triangle, size_t mesh_count,
                      size_t reserved_tlas_nodes, int32_t * out_root_indices, int32_t * out_triangle_positions, size_t * out_node_count, float * out_build_ms);
/* Spatial splits for the build above, decided by the caller (early split clipping): triangles [first_triangle, first_triangle + count) of the NEXT
 * rt_build_geometry are references -- copies of triangles that were cut into pieces, one copy per piece -- and `boxes` (6 floats each: min xyz,
 * max xyz) are the pieces' boxes. The tree is built over those boxes (intersected with the triangle's own), a leaf still tests the whole
 * triangle, so hits do not change; what changes is how many nodes a ray walks past a floor or wall triangle that spans half the scene
 * (the reference gets this from SBVHBuilder.cpp's spatial splits). Consumed by one build; the flattened static geometry uses it
 * (host/Integrator.cpp, cpu_config.device_presplit), whose copies carry the names of their originals (rt_upload_triangle_aliases).        */
int rt_set_build_boxes(rt_context * ctx, const float * boxes, size_t first_triangle, size_t count);
int rt_read_geometry(rt_context * ctx, void * out_triangles, void * out_bvh8_nodes);
int rt_upload_tlas(rt_context * ctx, const void * tlas_nodes, size_t tlas_node_count);
/* Replaces `bvh2_nodes` (Integrator.cpp:205-206): binary SAH BVH, 32 B nodes, for
 * rt_set_bvh_type(ctx, 2) (BASELINE config #1).  TLAS occupies the first slots likewise. */
int rt_upload_geometry_bvh2(rt_context * ctx, const void * triangles, size_t triangle_count,
                            const void * bvh2_nodes, size_t node_count);
int rt_upload_tlas_bvh2(rt_context * ctx, const void * tlas_nodes, size_t tlas_node_count);
/* Replaces `bvh4_nodes` (Integrator.cpp:216-251): 4-wide BVH, 128 B nodes (BVH/BVH.h:25-57), node 1
 * of the TLAS and of every BLAS is the entry point; for rt_set_bvh_type(ctx, 4).            */
int rt_upload_geometry_bvh4(rt_context * ctx, const void * triangles, size_t triangle_count,
                            const void * bvh4_nodes, size_t node_count);
int rt_upload_tlas_bvh4(rt_context * ctx, const void * tlas_nodes, size_t tlas_node_count);
/* Selects the trace kernels (kernel_trace_bvh2 / bvh4 / bvh8 of the reference, Pathtracer.cpp:115-135). */
int rt_set_bvh_type(rt_context * ctx, int bvh_width /* 2, 4 or 8 */);

/* Per-frame TLAS build ON THE DEVICE: replaces, for scenes whose instances move, the host work of Integrator::build_tlas
 * (Integrator.cpp:399-430: SAH build over the instance boxes, CWBVH conversion, re-ordering of the per-instance tables)
 * and rt_upload_tlas + rt_upload_instances. Everything is given in SCENE order (one entry per instance, any order the
 * host likes, the same every frame): root_indices / material_ids / transforms / transforms_inv / transforms_prev as for
 * rt_upload_instances, local_boxes = 6 floats per instance, the object-space min and max of its BLAS. One kernel launch
 * (asynchronous, a new version of the scene ring like the uploads) sorts the instances along a Morton curve, builds the
 * 8-wide compressed TLAS over them in node slots [0, 2 * mesh_count) and gathers the tables into TLAS order. CWBVH
 * kernels only; 1 <= mesh_count <= 4096. With a device-built TLAS rt_upload_lights takes SCENE indices in
 * light_mesh_transform_indices. rt_read_tlas copies the result back (tests, pixel-query translation):
 * order[position] = scene index, the nodes actually used (80 bytes each) and their count; any pointer may be NULL. */
int rt_build_tlas(rt_context * ctx, const int32_t * root_indices, const int32_t * material_ids,
                  const float * transforms, const float * transforms_inv, const float * transforms_prev,
                  const float * local_boxes, size_t mesh_count);
int rt_read_tlas(rt_context * ctx, int32_t * order, void * nodes, size_t node_capacity, int32_t * node_count);
/* Replaces mesh_bvh_root_indices / mesh_material_ids / mesh_transforms{,_inv,_prev}
 * (Integrator.cpp:412-429). Index = TLAS-order mesh id. Matrices are 12 floats, row-major
 * 3x4.  MSB of root_indices[i] = "identity transform" (Integrator.cpp:415).              */
int rt_upload_instances(rt_context * ctx, const int32_t * root_indices, const int32_t * material_ids,
                        const float * transforms, const float * transforms_inv, const float * transforms_prev,
                        size_t mesh_count);
/* Replaces material_types / materials (Pathtracer.cpp:545-589): types = 1 B each,
 * materials = 32 B each in the union layout of CUDA/Material.h:21-39.                    */
int rt_upload_materials(rt_context * ctx, const uint8_t * types, const void * materials, size_t count);
/* Replaces `media` (Pathtracer.cpp:681-697): 32 B each {sigma_a.xyz, g, sigma_s.xyz, pad}. */
int rt_upload_media(rt_context * ctx, const void * media, size_t count);
/* Replaces `textures` (Integrator.cpp:33-98). */
int rt_upload_textures(rt_context * ctx, const rt_texture_desc * descs, size_t count);
/* Where RT_TEXTURE_BC1 textures are decoded. enable = 1 (the default): once, by rt_upload_textures -- the device keeps 16 RGBA8
 * texels (64 bytes) per block, blocks and mip levels in the order of the compressed chain, and a filtered fetch reads texels.
 * enable = 0: the device keeps the 8-byte blocks and the shade kernels decode a block per texel fetch. The reference hands
 * the blocks to NVIDIA's texture unit (Assets/TextureLoader.cpp:208-262, CUDA/Material.h:60-75), which decodes for free; CDNA
 * compute has no such unit and the per-fetch decode was a third of a shade kernel's instructions, while 8 x the bytes of the
 * compressed chain is nothing next to 288 GB. Both settings produce the same texel values (one decode routine), hence the same
 * images. Takes effect at the next rt_upload_textures. rt_texture_bytes: device bytes the uploaded textures occupy.          */
int rt_set_texture_expansion(rt_context * ctx, int enable);
size_t rt_texture_bytes(rt_context * ctx);
/* Replaces light_* globals and lights_total_weight (Pathtracer.cpp:455-534).             */
int rt_upload_lights(rt_context * ctx,
                     const int32_t * light_triangle_indices, const float * light_triangle_cumulative_probability, size_t light_triangle_count,
                     const float * light_mesh_cumulative_probability, const int32_t * light_mesh_triangle_span /* 2 per mesh */,
                     const int32_t * light_mesh_transform_indices, size_t light_mesh_count,
                     float lights_total_weight);
/* Replaces pmj_samples / blue_noise_textures (Integrator.cpp:298-304):
 * pmj = 64*4096 float2, blue_noise = 16*128*128 uchar2.                                   */
int rt_upload_rng(rt_context * ctx, const float * pmj_samples, const uint8_t * blue_noise);
/* Replaces sky_texture / sky_scale (Integrator.cpp:285-296): equirect float4 image.       */
int rt_set_sky(rt_context * ctx, const float * rgba, int width, int height, float scale);

/* ---- per-frame state ------------------------------------------------------------------ */
/* Replaces resize_init/resize_free: screen_width/pitch/height, AOV buffers, SVGF buffers
 * (Pathtracer.cpp:255-301,316-357). pitch = round_up(width, 32).                          */
int rt_resize(rt_context * ctx, int width, int height);
int rt_set_camera(rt_context * ctx, const rt_camera * camera);
/* Replaces `svgf_data` (Pathtracer.cpp:707-717): two row-major 4x4 matrices.              */
int rt_set_svgf_matrices(rt_context * ctx, const float * view_projection, const float * view_projection_prev);
int rt_set_config(rt_context * ctx, const rt_gpu_config * config);
/* How the a-trous passes of the SVGF filter (CUDA/SVGF/SVGF.h:416-554: kernel_svgf_atrous, one thread per pixel, nine taps of three images
 * through the texture cache) fetch their taps. enable = 1 (the default): a workgroup stages the (direct, indirect, normal + depth) values
 * its pixels tap -- rows `step` apart, 64 adjacent columns plus `step` on either side -- in LDS once and taps from there; enable = 0: every
 * tap is a global load, as in the reference. The arithmetic of a pixel is one function for both: images are bit-identical. Passes with a
 * step above 32 (more than six iterations) always use the untiled form.                                                        */
int rt_set_svgf_tiles(rt_context * ctx, int enable);
/* Multi-GPU tile split: this context only renders pixels [offset, offset+count) of the
 * scan-order frame (the reference's kernel_generate(sample, pixel_offset, pixel_count),
 * Pathtracer.cu:122-131). Default = whole frame.                                          */
int rt_set_pixel_range(rt_context * ctx, int pixel_offset, int pixel_count);

/* Interleaved variant for load balance: this context owns tiles first_tile, first_tile +
 * tile_stride, ... of `tile_pixels` consecutive scan-order pixels each (whole rows).
 * rt_set_pixel_range switches back to one contiguous range.                                */
int rt_set_pixel_tiles(rt_context * ctx, int tile_pixels, int first_tile, int tile_stride);
/* Frame exchange for the tile split (both asynchronous on the context's stream, DEVICE pointers):
 * pack   copies this context's `tiles` tiles of the final image into dst (tiles*tile_pixels
 *        float4, zero padded) -- the send buffer of one all-gather over RCCL;
 * unpack scatters the all-gathered buffer [world][tiles_per_rank*tile_pixels] float4 back into
 *        the final image in scan order.                                                       */
int rt_pack_pixels(rt_context * ctx, void * dst_device, int tile_pixels, int first_tile, int tile_stride, int tiles);
int rt_unpack_pixels(rt_context * ctx, const void * src_device, int tile_pixels, int world, int tiles_per_rank);

/* The same exchange WITHOUT a Python / torch.distributed layer: the tile split of a C++ host (host/FrameSplit.h,
 * `pathtracer --devices 0,1,...`). One communicator per context, over RCCL (librccl.so is bound at run time, so the library
 * has no link-time dependency on it):
 *   several processes, one GPU each:  rank 0 calls rt_comm_unique_id and hands the 128 bytes to the others (any side channel:
 *                                     a file, a socket, MPI), every rank calls rt_comm_init_rank (= ncclCommInitRank);
 *   one process, several GPUs:        rt_comm_init_all over its contexts (= ncclCommInitAll). Contexts that SHARE a GPU --
 *                                     which RCCL refuses -- are joined by stream-ordered peer copies instead (tests).
 * A context renders the tiles rt_set_pixel_tiles(ctx, tile_pixels, rank, world) names; rt_all_gather_framebuffer packs them,
 * all-gathers (ncclAllGather on the context's stream) and scatters the result into every context's final image, asynchronously.
 * Contexts of one process are exchanged by ONE call (grouped: ncclGroupStart / End). rt_all_gather_svgf_inputs moves what the
 * SVGF filter stage reads of a frame instead (DIRECT, INDIRECT, ALBEDO, the g-buffers: 80 B per pixel; see rt_pack_svgf_inputs).
 * Environment GRT_COLLECTIVE_LIBRARY names another library with RCCL's entry points (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll, ncclCommDestroy,
 * ncclAllGather, ncclGroupStart / End, ncclGetErrorString) to bind instead of librccl.so: a site's own build -- or the loopback stand-in of the test suite
 * (tests/support/loopback_ccl.cpp), which lets two PROCESSES that share one GPU run this exchange with world = 2 (tests/test_gpu_rccl.py).               */
int rt_comm_unique_id(void * out_id_128_bytes);
int rt_comm_init_rank(rt_context * ctx, const void * unique_id_128_bytes, int rank, int world);
int rt_comm_init_all(rt_context ** contexts, int count);
int rt_comm_destroy(rt_context * ctx);
int rt_all_gather_framebuffer(rt_context * ctx);
int rt_all_gather_framebuffers(rt_context ** contexts, int count);
int rt_all_gather_svgf_inputs(rt_context ** contexts, int count);
/* Stream-ordered hand-over of device buffers between the context and a consumer's HIP stream
 * (e.g. the stream RCCL runs on), so that the host never has to block between frames:
 *   rt_stream_wait_for_context: `stream` (a hipStream_t) waits for everything the context has
 *       enqueued on its transfer stream so far (rt_pack_pixels, rt_unpack_pixels);
 *   rt_context_wait_for_stream: the context's transfer stream waits for everything enqueued on
 *       `stream` so far (e.g. the collective that still reads the buffer the next pack writes). */
/* SVGF / TAA frames under the tile split (config 3 on N GPUs). The filter stage needs neighbourhoods of +-(3 + 2^5) pixels
 * and reprojects from anywhere in the previous frame, so every rank filters the WHOLE frame, redundantly: a rank path-traces
 * its own tiles (rt_render_sample_unfiltered: like rt_render_sample, stopping before the filter stage), the ranks exchange
 * what the filter reads of this frame -- the per-frame DIRECT / INDIRECT / ALBEDO buffers and the three g-buffers, packed
 * as 5 float4 per pixel by rt_pack_svgf_inputs (same tile arguments as rt_pack_pixels), one all-gather,
 * rt_unpack_svgf_inputs -- and rt_filter_frame runs reproject / variance / a-trous / finalize / TAA on every rank
 * (Pathtracer.cpp:798-838). Bit-identical to one context rendering the whole frame (tests/test_gpu_materials_svgf.py). */
int rt_render_sample_unfiltered(rt_context * ctx, int sample_index);
int rt_pack_svgf_inputs(rt_context * ctx, void * dst_device, int tile_pixels, int first_tile, int tile_stride, int tiles);
int rt_unpack_svgf_inputs(rt_context * ctx, const void * src_device, int tile_pixels, int world, int tiles_per_rank);
int rt_filter_frame(rt_context * ctx, int sample_index);
int rt_stream_wait_for_context(rt_context * ctx, void * stream);
int rt_context_wait_for_stream(rt_context * ctx, void * stream);

/* Pixels per wavefront batch. The reference hard-codes BATCH_SIZE = 1080*720 to bound VRAM
 * (Common.h:69-71) and loops over batches; results do not depend on it. Default 0 = the whole
 * frame in one batch (MI355X has 288 GB of HBM; bigger launches hide traversal latency better). */
int rt_set_batch_size(rt_context * ctx, int batch_size);

/* Samples per pixel rendered concurrently (1..8, default 3). The reference submits the ~40
 * launches of one sample strictly one after the other on one stream (Pathtracer.cpp:738-855);
 * here consecutive rt_render_sample calls alternate between `count` sets of queues / streams and
 * only the accumulate step is ordered between them, so a sample's small deep-bounce launches
 * run in the shadow of its neighbour's. Results are identical for every count (SVGF frames
 * included: only their filter stage is ordered). Profiled / statistics passes run one at a time. Memory: one set of wavefront
 * queues + per-sample frame buffers per sample in flight (~1 GB each at 1920x1080).        */
int rt_set_samples_in_flight(rt_context * ctx, int count);

/* ---- render: replaces Pathtracer::render (Pathtracer.cpp:738-855) ---------------------- */
/* One sample for this context's pixel range: generate, (trace, sort, shade*, shadow) x
 * bounces, then accumulate or SVGF/TAA.  Asynchronous; rt_synchronize or a read waits.    */
int rt_render_sample(rt_context * ctx, int sample_index);
/* sample_count (1..16) consecutive samples per pixel, sample_index .. sample_index+count-1, as
 * ONE wavefront: every launch carries count x as many paths, so the deep bounces (and the small
 * pixel share of one rank of a multi-GPU job) fill the GPU, and the launch-latency floor of a
 * bounce is paid once per batch. Each path is keyed on (pixel, its own sample index) and the
 * per-sample frame buffers are folded into the accumulators in sample order, so the result is
 * bit-identical to count calls of rt_render_sample (tests/test_gpu_parity.py). Not for SVGF.
 * rt_get_counters reports the batch totals.                                               */
int rt_render_samples(rt_context * ctx, int sample_index, int sample_count);

/* How consecutive rt_render_sample(s) calls are scheduled on the device.
 * RT_SCHEDULER_MERGED (default): the calls feed ONE wavefront. Each call ("submission") adds its primary rays to the
 *   wavefront and advances it by one iteration (trace, sort, shade of every submission in flight: the new one is at
 *   bounce 0, the one before at bounce 1, ...), so every launch has the size of a whole sample and a submission is
 *   complete num_bounces - 1 calls later; reads and rt_synchronize run the remaining iterations. Same arithmetic per
 *   path, per-sample frame buffers folded in submission order: images are bit-identical to RT_SCHEDULER_SLOTS.
 *   (The reference runs the chain of one sample at a time, Pathtracer.cpp:738-855: its deep bounces are launches of a
 *   few thousand rays.) SVGF frames go through it as well: every sample slot has its own g-buffer set, a frame is
 *   filtered when it has passed its last bounce, frames in submission order. The binary / 4-wide BVH kernels,
 *   num_bounces = 0, explicit pixel batches and the SVGF tile split always use the slot scheduler.
 * RT_SCHEDULER_SLOTS: one launch chain per submission, up to rt_set_samples_in_flight chains concurrently on their own
 *   streams, each reading the scene version (TLAS, instances, lights) that was current when it was submitted: the
 *   choice for scenes that upload a new TLAS every frame.                                                           */
enum { RT_SCHEDULER_MERGED = 0, RT_SCHEDULER_SLOTS = 1 };
int rt_set_scheduler(rt_context * ctx, int scheduler);
/* Merged scheduler: one more iteration of the wavefront without new samples (no-op when nothing is in flight), and the
 * number of submissions whose accumulate step has been enqueued so far -- a frame loop that hands every completed
 * frame to a collective calls rt_advance / rt_render_samples and packs the frame when the count moves
 * (bench.py); rt_pack_pixels and the reads on their own complete everything first.                       */
int rt_advance(rt_context * ctx);
/* enable = 1: the application keeps several frames in flight.
 *  - rt_pack_pixels / rt_unpack_pixels are ordered after the submissions COMPLETED so far instead of first completing
 *    everything in flight, so that later frames keep filling the wavefront while an earlier one is exchanged;
 *  - small submissions share iterations: a submission generates its primary rays at once, but the iteration that traces
 *    them is only enqueued when 1920 x 1080 x 4 paths are waiting for it (or 8 submissions), so that the launches of one
 *    rank of an N-GPU tile split are as large as those of a whole frame. Whatever needs progress enqueues the iteration
 *    with the submissions that are there: rt_advance, a camera change, every call that completes the work in flight
 *    (reads, uploads, rt_synchronize). rt_submissions_completed only reports.                                          */
int rt_set_frame_pipelining(rt_context * ctx, int enable);
/* How many paths the submissions of ONE iteration may bring (frame pipelining on; default 1920 x 1080 x 4, 0 restores it; at most 8
 * submissions share an iteration whatever the number). An application that knows its burst -- a batch job of N samples that asks for
 * the image only at the end -- gives the burst's paths: its frames then enter the wavefront together and walk their bounces side by
 * side, N_bounces iterations in all, instead of one after the other through N_submissions + N_bounces - 1 iterations of which the
 * first and the last N_bounces - 1 are only partly filled (20 samples at 1080p as five 4-sample submissions: 10 iterations instead
 * of 14, 1.49 -> 1.43 ms per sample). Queues and sample frames grow with the number (412 bytes per path, one frame per sample and
 * bounce in flight). A long-running frame loop keeps the default: its steady state already fills every iteration.                 */
int rt_set_stream_batch(rt_context * ctx, long long paths);
int rt_submissions_completed(rt_context * ctx, uint64_t * out_count);
/* Replaces the `pixel_query` global (Integrator.h:266-277, Integrator.cpp:483-495, Pathtracer.cu:345-348):
 * the mesh (TLAS-order id) and triangle that the primary ray of pixel `pixel_index` = x + y * pitch hits
 * in the next sample rendered (-1 disarms it). rt_get_pixel_query waits and returns -1, -1 for a miss. */
int rt_set_pixel_query(rt_context * ctx, int pixel_index);
int rt_get_pixel_query(rt_context * ctx, int * mesh_id, int * triangle_id);
/* The reference's second integrator, AO::render (Integrators/AO.cpp:148-200, CUDA/AO.cu):
 * primary hit -> one cosine-weighted occlusion ray of length ao_radius -> RADIANCE = 1 where it
 * escapes, NORMAL / POSITION AOVs; needs geometry, instances, RNG tables, rt_resize, rt_set_camera
 * (materials, lights and sky are not used). One sample per call for the context's pixel range. */
int rt_render_ao_sample(rt_context * ctx, int sample_index, float ao_radius);
int rt_synchronize(rt_context * ctx);
/* Per-stage HIP-event timing (ms_generate..ms_post of rt_counters) costs ~2 events per
 * kernel launch, so it is opt-in; ms_total is always measured. Replaces the CUDAEventPool
 * instrumentation of the reference (Pathtracer.cpp:751-843, Device/CUDAEvent.h:31-53).
 * enable = 1: events around every stage; samples are rendered one at a time with the shadow rays
 *             on the main chain, so the stage times are those of each kernel running alone.
 * enable = 2: events around EVERY traversal launch, on the stream each is launched on; concurrency
 *             (merged wavefront, or side stream and samples in flight) stays as in production.
 *             rt_get_counters then returns in ms_trace / ms_shadow the SUM over those launches
 *             since the mode was enabled or counters were last read, rt_get_launch_timings each.
 * enable = 3: as 2, plus events around every OTHER launch of the merged wavefront (generate, sort, one per
 *             material queue, accumulate, each SVGF / TAA kernel): the per-stage rooflines of bench.py. */
int rt_set_profiling(rt_context * ctx, int enable);
/* What a timed launch was (rt_set_profiling 2 / 3). */
enum {
	RT_TIMING_TRACE = 0,        /* closest-hit launches; the fused traversal launch of the merged wavefront */
	RT_TIMING_SHADOW = 1,       /* separate shadow launches (slot scheduler) */
	RT_TIMING_SORT = 2, RT_TIMING_GENERATE = 3, RT_TIMING_ACCUMULATE = 4,
	RT_TIMING_MATERIAL_0 = 5,   /* + material slot: diffuse, plastic, dielectric, conductor */
	RT_TIMING_SVGF_REPROJECT = 9, RT_TIMING_SVGF_VARIANCE = 10, RT_TIMING_SVGF_ATROUS = 11, RT_TIMING_SVGF_FINALIZE = 12,
	RT_TIMING_TAA = 13, RT_TIMING_TAA_FINALIZE = 14,
	RT_TIMING_KINDS = 15
};
/* Durations in milliseconds of the launches of one kind since the mode was enabled (or since the last call for
 * that kind), in submission order; each event pair sits on the stream the launch runs on.                    */
int rt_get_launch_timings(rt_context * ctx, int kind, float * out_ms, int capacity, int * out_count);
/* Work statistics of the trace kernels: when enabled, rt_render_sample runs counting variants
 * of kernel_trace(_shadow)_bvh8 (slower: per-ray atomics) and rt_get_trace_statistics returns,
 * for the last sample, 10 x u64 {closest-hit: BVH8 nodes fetched, triangles tested, transformed
 * instance entries, identity instance entries, rays; then the same five for shadow rays}.
 * These are the N_node / N_tri / N_inst of the algorithmic-bytes roofline.     */
int rt_set_trace_statistics(rt_context * ctx, int enable);
/* Merged scheduler with statistics on: the ten counters after each iteration (cumulative), oldest first; per-launch
 * work = difference of consecutive rows.                                                                  */
int rt_get_trace_statistics_history(rt_context * ctx, uint64_t * out_rows10, int capacity_rows, int * out_rows);
int rt_get_trace_statistics(rt_context * ctx, uint64_t * out10);
/* Counters of the most recent completed rt_render_sample (synchronous).                    */
int rt_get_counters(rt_context * ctx, rt_counters * out);

/* ---- results ----------------------------------------------------------------------------*/
/* Replaces get_aov(type).accumulator read-back (Main.cpp:226-249). Copies pitch*height
 * float4 to dst (host). accumulated=0 reads the per-frame framebuffer instead.             */
int rt_read_aov(rt_context * ctx, int aov_type, float * dst, int accumulated);
/* The final image (`accumulator` surface, Pathtracer.cu:24): pitch*height float4.          */
int rt_read_framebuffer(rt_context * ctx, float * dst);
/* Device pointer of the final image for zero-copy consumers (RCCL gather).                 */
int rt_framebuffer_device_ptr(rt_context * ctx, void ** out_ptr, size_t * out_bytes);
int rt_screen_pitch(rt_context * ctx);

/* Kulla-Conty LUTs as computed on the device at first use (kernel_integrate_* /
 * kernel_average_*, CUDA/KullaConty.h:83-240): 16^3, 16^3, 16^2, 16^2, 32^2, 32 floats.
 * Any pointer may be NULL. Synchronous.                                                     */
int rt_read_luts(rt_context * ctx, float * dielectric_directional_enter, float * dielectric_directional_leave,
                 float * dielectric_enter, float * dielectric_leave, float * conductor_directional, float * conductor);

/* ---- kernel-level entry points (parity tests and micro-benchmarks) ----------------------*/
/* kernel_trace_bvh8 / kernel_trace_shadow_bvh8 on caller-supplied rays (SoA host arrays).
 * hits: ray_count x uint4 {mesh_id, triangle_id, t bits, u16|v16<<16} (Buffers.h:25-32).
 * Synchronous. `repeat` > 1 re-runs the kernel for timing; out_ms = mean kernel ms.        */
int rt_trace_rays(rt_context * ctx, const float * ox, const float * oy, const float * oz,
                  const float * dx, const float * dy, const float * dz, size_t ray_count,
                  uint32_t * hits, int repeat, float * out_ms);
int rt_trace_shadow_rays(rt_context * ctx, const float * ox, const float * oy, const float * oz,
                         const float * dx, const float * dy, const float * dz, const float * max_distance,
                         size_t ray_count, uint8_t * occluded, int repeat, float * out_ms);
/* kernel_generate only: writes the primary rays of pixels [offset, offset+count).          */
int rt_generate_rays(rt_context * ctx, int sample_index, int pixel_offset, int pixel_count,
                     float * ox, float * oy, float * oz, float * dx, float * dy, float * dz,
                     uint32_t * pixel_index_and_flags);
/* random<Dim>() (CUDA/Sampling.h:44-84) for `count` (pixel_index) values: out = float2 each. */
int rt_random_samples(rt_context * ctx, int dimension, const uint32_t * pixel_indices, size_t count,
                      uint32_t bounce, uint32_t sample_index, float * out_xy);
/* Streaming-read bandwidth probe used as the measured HBM roofline (GB/s).                 */
int rt_measure_stream_bandwidth(rt_context * ctx, size_t bytes, int repeat, float * out_gbps);

#ifdef __cplusplus
}
#endif
#endif /* GPU_RAYTRACER_AMD_H */
