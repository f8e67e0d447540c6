// rt_types.h -- device-side data layout shared by the C ABI implementation and the kernels.
//
// HBM layout (all buffers are plain hipMalloc'd linear memory, 16-byte aligned):
//   triangles      float4[6 * T]   device triangle = 96 B (reference CUDA/Raytracing/Triangle.h:4-11)
//   bvh8_nodes     float4[5 * N]   CWBVH node = 80 B        (reference CUDA/Raytracing/BVH8.h:19-25)
//   per-instance   int[M], float4[3 * M] x3                 (reference CUDA/Raytracing/Mesh.h:29-36)
//   ray queues     SoA, one float/uint array per component, RT_BATCH_SIZE entries each
//                  (reference Pathtracer.cu:32-66) -- every queue access by consecutive lanes
//                  is a fully coalesced 256 B transaction per 64-lane wave.
//   AOVs           float4[pitch * height] framebuffer + accumulator per enabled AOV
// Material queues: the reference packs two material types into one allocation growing from
// both ends to save VRAM (Pathtracer.cu:73-90); with 288 GB of HBM3E each type simply gets
// its own queue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpu_raytracer_amd.h"

#define RT_WAVE_SIZE 64
#define RT_INFINITY __builtin_huge_valf()

struct RtVec3SoA { float * x, * y, * z; };

struct RtTraceBuffer { // Pathtracer.cu:33-46
	RtVec3SoA origin, direction;
	uint4   * hits;
	float   * cone_angle, * cone_width;
	int     * medium;
	unsigned * pixel_index_and_flags;
	RtVec3SoA throughput;
	float   * last_pdf;
};

struct RtMaterialBuffer { // Pathtracer.cu:49-61
	RtVec3SoA direction;
	uint4   * hits;
	float   * cone_angle, * cone_width;
	int     * medium;
	unsigned * pixel_index_and_flags;
	RtVec3SoA throughput;
};

struct RtShadowBuffer { // Pathtracer.cu:64-68, BVH.h:16-21
	RtVec3SoA origin, direction;
	float  * max_distance;
	float4 * illumination_and_pixel_index;
};

struct RtBufferSizes { // Pathtracer.cu:103-114
	int trace     [RT_MAX_BOUNCES];
	int diffuse   [RT_MAX_BOUNCES];
	int plastic   [RT_MAX_BOUNCES];
	int dielectric[RT_MAX_BOUNCES];
	int conductor [RT_MAX_BOUNCES];
	int shadow    [RT_MAX_BOUNCES];
	int rays_retired       [RT_MAX_BOUNCES];
	int rays_retired_shadow[RT_MAX_BOUNCES];
};

// ---- merged wavefront (rt_api.hip, "PathStream") -----------------------------------------------------------
// Instead of one launch chain per submission (generate -> (trace, sort, shade, shadow) x bounces), whose launches
// shrink bounce by bounce until they no longer fill the GPU, consecutive submissions feed ONE wavefront: iteration
// i traces / sorts / shades the rays of every submission in flight -- the primary rays of the newest next to bounce
// 1 of the one before, bounce 2 of the one before that ... -- so every launch has the size of a whole sample and
// there are no per-submission tails. A queue entry does not store its bounce: its virtual pixel index names a SAMPLE
// SLOT (v = slot * frame_pixels + pixel; the slot's per-sample AOV frame lives at the same offset), and the slot table
// says which sample that is and in which iteration it was generated; bounce = iteration - birth.
#define RT_STREAM_SAMPLE_SLOTS 512    // samples in flight (each owns one frame of the per-sample AOV buffers)
#define RT_STREAM_SUBMISSIONS  128    // ring of submissions whose per-bounce statistics are kept
enum { RT_STAT_TRACE = 0, RT_STAT_SHADOW, RT_STAT_DIFFUSE, RT_STAT_PLASTIC, RT_STAT_DIELECTRIC, RT_STAT_CONDUCTOR, RT_STAT_KINDS };

struct RtStreamSlot { int sample_index, birth_iteration, submission, index_in_submission; }; // one int4 per sample slot

struct RtStreamTable {                 // written by the host per submission (asynchronous copy in stream order)
	RtStreamSlot slots[RT_STREAM_SAMPLE_SLOTS];
	int submission_birth[RT_STREAM_SUBMISSIONS];
};

#define RT_ENDGAME_MAX_WAVES 16384     // waves of the persistent traversal grid that can own a region of a queue's end (256 CUs x 16 workgroups x 4)
struct RtStreamControl {               // queue sizes and ray-claim cursors; [iteration & 1] where two launches overlap in time
	int trace_count[2];                // rays in trace queue [i & 1]: appended by sort / shade of i - 1 and generate of i
	int material_count[4];
	int shadow_count[2];               // shadow rays emitted by the shade kernels of iteration i, traced by the launch of i + 1
	int cursor[2][2];                  // [i & 1][closest, shadow] cursors of the fused trace launch
	int pad[4];
	int stats[RT_STREAM_SUBMISSIONS][RT_STAT_KINDS][RT_MAX_BOUNCES]; // rays per submission, queue kind and bounce
	int endgame[2][2][RT_ENDGAME_MAX_WAVES];   // [i & 1][closest (or the mixed launch's one queue), shadow][region]: the end of a queue is dealt in regions, one per wave, that other waves help to finish (kernels_trace.hip: fetch_ray)
};

struct RtTexture {
	const uchar4 * texels;   // linear RGBA8, mip levels back to back; or BC1 blocks (uint2 each); or expanded blocks (16 texels each), see `format`
	int   width, height, mip_levels;
	float lod_bias;          // 0.5 * log2(width * height)   (Integrator.cpp:95)
	int   format;            // RT_TEXTURE_RGBA8 / RT_TEXTURE_BC1 / RT_TEXTURE_BC1_EXPANDED
	int   pad;
};
// Device-only format (rt_set_texture_expansion, on by default): a BC1 texture whose blocks rt_upload_textures has decoded ONCE,
// with the shade kernels' own bc1_texel, into 16 RGBA8 texels each -- 64 bytes per block, row-major inside the block, blocks and
// levels in the order of the compressed chain. The reference leaves the decode to NVIDIA's texture unit; this chip has none, and
// decoding a block per texel fetch was ~55 vector instructions x 8 texels x up to 16 probes per lookup: a third of a shade
// kernel's instructions. 8 x the bytes of BC1 (Sponza: 13 -> 105 MB of 288 GB), the same texel values to the bit, and a bilinear
// footprint still lies in one or two 64-byte blocks (a linear RGBA8 image would spread it over two rows).
#define RT_TEXTURE_BC1_EXPANDED 2

struct RtAOV { float4 * framebuffer, * accumulator; };

// Everything a kernel needs, passed BY VALUE as kernel argument (lives in the kernarg
// segment and is read with scalar loads; no __constant__ symbols, so several contexts
// can coexist in one process).
struct RtParams {
	// geometry
	const float4 * triangles;            // 6 float4: full shading triangle
	const float4 * triangle_positions;   // 3 float4: position_0, edge_1, edge_2 (traversal copy)
	const float4 * bvh8_nodes;
	const float4 * bvh2_nodes;  // 2 float4 per node
	const float4 * bvh4_nodes;  // 8 float4 per node
	// The TLAS (node indices [0, tlas_node_count) of whichever BVH type is selected) is versioned per
	// frame and lives outside the static node arrays, see rt_api.hip (SceneRing).
	const float4 * tlas_nodes;
	int tlas_node_count;
	int bvh_width;              // 8: CWBVH kernels (default), 4: 4-wide BVH kernels, 2: binary-BVH kernels
	const int    * mesh_bvh_root_indices;
	int mesh_count;                   // instances (the fused traversal launch keeps the root table of a small scene in LDS)
	int geometry_below_4gib;          // bvh8_nodes and triangle_positions are both shorter than 4 GiB: the flattened scene's engine addresses them with 32-bit offsets
	int entry_tlas_stack_size;        // RT_INVALID: rays start at the TLAS root; 0: node 0 is the root of the one world-space tree that holds the
	                                  // whole scene and rays start inside it, as instance row 0 (rt_set_static_geometry)
	int skip_behind_hit;              // rt_set_skip_behind_hit's wish (default 1); what the kernels do is rt_skip_walk(p) below
	int has_triangle_aliases;         // some triangles are copies that report the (instance, triangle) named in the padding of their
	                                  // position record instead of themselves (rt_upload_triangle_aliases)
	const int    * mesh_material_ids;
	const float4 * mesh_transforms, * mesh_transforms_inv, * mesh_transforms_prev;
	// TLAS built on the device (rt_build_tlas): scene index of an instance -> its position in TLAS order, for tables that
	// name instances by scene index (light_mesh_transform_indices); null when the host supplied everything in TLAS order
	const int    * mesh_position;
	// materials
	const uint8_t * material_types;
	const float4  * materials;  // 2 float4 per material
	const float4  * media;      // 2 float4 per medium
	const RtTexture * textures;
	int svgf_tiles;                   // 1 (default): the a-trous passes stage a workgroup's taps in LDS (rt_set_svgf_tiles; kernels_post.hip: kernel_svgf_atrous_tiled)
	int textures_compressed;          // 1: at least one texture holds BC1 blocks that a fetch has to decode (rt_set_texture_expansion(ctx, 0)); picks the material kernels' instantiation
	// lights
	const int   * light_triangle_indices;
	const float * light_triangle_cumulative_probability;
	const float * light_mesh_cumulative_probability;
	const int2  * light_mesh_triangle_span;
	const int   * light_mesh_transform_indices;
	int   light_mesh_count, light_triangle_count;
	float lights_total_weight;
	// rng
	const float2 * pmj_samples;
	const uchar2 * blue_noise;
	// sky
	const float4 * sky;
	int sky_width, sky_height;
	float sky_scale;
	// Kulla-Conty LUTs
	const float * lut_dielectric_directional_albedo_enter, * lut_dielectric_directional_albedo_leave;
	const float * lut_dielectric_albedo_enter, * lut_dielectric_albedo_leave;
	const float * lut_conductor_directional_albedo, * lut_conductor_albedo;
	// frame
	rt_camera     camera;
	rt_gpu_config config;
	float view_projection[16], view_projection_prev[16];
	int screen_width, screen_height, screen_pitch;
	// Sample batching: rt_render_samples renders `batch_samples` consecutive samples of every pixel as
	// ONE wavefront. The queues and the per-sample AOV frame buffers are addressed with a "virtual"
	// pixel index v = s * frame_pixels + pixel (s = sample within the batch, frame_pixels = pitch *
	// height); only the RNG needs (pixel, first_sample + s) back, see rt_split_virtual_pixel.
	unsigned frame_pixels, frame_pixels_magic; // magic = floor(2^32 / frame_pixels) + 1
	int batch_samples;
	// pixel query (PixelQuery of the reference, CUDA/Pathtracer.cu:345-348): the bounce-0 hit of this pixel
	// index (x + y * pitch, -1 = none) is written to pixel_query_out[0..1] = { mesh_id, triangle_id }
	int   pixel_query_pixel;
	int * pixel_query_out;
	// multi-GPU tile split: local pixel i -> scan-order pixel (tile_pixels == 0: identity)
	int tile_pixels, tile_first, tile_stride;
	// queues
	RtTraceBuffer    trace[2];
	RtMaterialBuffer material[4];   // diffuse, plastic, dielectric, conductor
	RtShadowBuffer   shadow;
	RtBufferSizes  * sizes;
	// merged wavefront only (null / 0 otherwise)
	RtStreamControl     * stream;
	const RtStreamTable * stream_table;
	int stream_iteration;
	int * xcd_counters;               // [RT_MAX_BOUNCES][2 (closest, shadow)][8 XCDs] ray-fetch cursors
	uint2 * stack_spill;                // traversal stack entries beyond the LDS part, [entry][grid lane]
	// outputs
	RtAOV    aovs[RT_AOV_COUNT];
	float4 * final_image;           // the reference's `accumulator` surface
	// SVGF / TAA
	float4 * gbuffer_normal_and_depth;
	int2   * gbuffer_mesh_id_and_triangle_id;
	float2 * gbuffer_screen_position_prev;
	float4 * frame_buffer_moment;
	int    * history_length;
	float4 * history_direct, * history_indirect, * history_moment, * history_normal_and_depth;
	float4 * taa_frame_prev, * taa_frame_curr, * taa_frame_next;   // next: where kernel_taa writes the history of the following frame (swapped with prev after every filtered frame)
	int * svgf_young_pixels;          // [0]: how many pixels this frame's kernel_svgf_reproject left with fewer than 4 frames of history, [RT_SVGF_YOUNG_HEADER ...]: their indices (kernel_svgf_variance_listed); emptied by kernel_svgf_finalize
	float2 * svgf_variance[2];        // (direct.w, indirect.w) of the radiance framebuffers [0] and accumulators [1], kept in step by the filter kernels
	float4 * svgf_normal_and_depth;   // (normal, depth) of the frame being filtered: decoded once by kernel_svgf_reproject for the variance / a-trous taps
};
// "Skip behind the hit" (kernels_trace.hip): closest-hit rays drop stacked groups of children that lie behind the hit they hold. Taken when the context wants it
// (rt_set_skip_behind_hit) AND the scene is ONE tree the flattened scene's engine walks (rt_set_static_geometry(ctx, 1), arrays below 4 GiB): every CWBVH
// closest-hit kernel -- the merged wavefront's launch, its counting variant, the per-bounce and the explicit kernels -- decides by this one rule.
static inline __host__ __device__ bool rt_skip_walk(const RtParams & p) { return p.skip_behind_hit != 0 && p.entry_tlas_stack_size == 0 && p.geometry_below_4gib != 0; }


// Scan-order index of local pixel i of this context: its tiles are tile_first, tile_first +
// tile_stride, ... each tile_pixels long (whole rows), see gpu-raytracer_amd/parallel.py.
__device__ __forceinline__ int rt_map_pixel(const RtParams & p, int i) {
	if (p.tile_pixels == 0) return i;
	return ((i / p.tile_pixels) * p.tile_stride + p.tile_first) * p.tile_pixels + i % p.tile_pixels;
}

// v -> pixel, and the sample within the batch (v < 2^30, at most 16 samples per batch)
__device__ __forceinline__ unsigned rt_split_virtual_pixel(const RtParams & p, unsigned v, unsigned & sample_in_batch) {
	unsigned s = __umulhi(v, p.frame_pixels_magic);   // floor(v / frame_pixels) or one too many
	if (s * p.frame_pixels > v) s--;
	sample_in_batch = s;
	return v - s * p.frame_pixels;
}

// merged wavefront: slot, sample and bounce of a queue entry (see RtStreamSlot). `sample_index_for_rng` is what
// random_sample() expects from its callers: it adds the slot number it splits off the virtual index itself.
struct RtPathInfo { int bounce, submission; unsigned sample_index_for_rng; bool first_of_submission; };
__device__ __forceinline__ RtPathInfo rt_stream_path_info(const RtParams & p, unsigned virtual_pixel) {
	unsigned slot;
	rt_split_virtual_pixel(p, virtual_pixel, slot);
	RtStreamSlot e = p.stream_table->slots[slot];
	RtPathInfo info;
	info.bounce = p.stream_iteration - e.birth_iteration;
	info.submission = e.submission;
	info.sample_index_for_rng = unsigned(e.sample_index) - slot;
	info.first_of_submission = e.index_in_submission == 0;
	return info;
}

#define RT_FLAG_ALLOW_NEE     (1u << 31)
#define RT_FLAG_INSIDE_MEDIUM (1u << 30)
#define RT_FLAGS_ALL          (RT_FLAG_ALLOW_NEE | RT_FLAG_INSIDE_MEDIUM)
#define RT_SHADOW_FLAG_BOUNCE_0 (1u << 30)   // merged wavefront: pixel word of a shadow ray emitted at bounce 0

// Launch helpers implemented in the kernel translation units
void rt_launch_generate(const RtParams & p, int sample_index, int pixel_offset, int pixel_count, hipStream_t stream);
void rt_launch_expand_bc1(const uint2 * blocks, uchar4 * texels, size_t block_count, hipStream_t stream);
void rt_launch_trace(const RtParams & p, int bounce, hipStream_t stream);
// merged wavefront (the iteration is p.stream_iteration); stats: null, or 10 x u64 as for the counting variants below
void rt_launch_generate_stream(const RtParams & p, int sample_index, int pixel_offset, int pixel_count, int slot_base, int queue_offset, int block_width, int band_rows, hipStream_t stream);
void rt_launch_stream_advance(RtStreamControl * control, int iteration, int generated, int * progress, int reset_ring_first, int reset_ring_count, hipStream_t stream);
void rt_launch_trace_stream(const RtParams & p, unsigned long long * stats, hipStream_t stream);
void rt_launch_sort_stream(const RtParams & p, hipStream_t stream);
void rt_launch_material_stream(const RtParams & p, int material_slot, hipStream_t stream);
void rt_launch_trace_shadow(const RtParams & p, int bounce, hipStream_t stream);
void rt_launch_ambient_occlusion(const RtParams & p, int sample_index, float ao_radius, hipStream_t stream);
void rt_launch_trace_shadow_ao(const RtParams & p, hipStream_t stream);
void rt_launch_sort(const RtParams & p, int bounce, int sample_index, hipStream_t stream);
void rt_launch_material(const RtParams & p, int material_slot, int bounce, int sample_index, hipStream_t stream);
void rt_launch_accumulate(const RtParams & p, float frames_accumulated, int pixel_offset, int pixel_count, hipStream_t stream);
#define RT_SVGF_YOUNG_HEADER 16
#define RT_ACCUMULATE_GROUP 8
struct RtAccumulateGroup { int count; int first_sample[RT_ACCUMULATE_GROUP], sample_count[RT_ACCUMULATE_GROUP], slot_base[RT_ACCUMULATE_GROUP]; };
void rt_launch_accumulate_group(const RtParams & p, const RtAccumulateGroup & group, int pixel_offset, int pixel_count, hipStream_t stream);
// mark(user, k, stream), if given, is called before and after kernel k = 0 reproject, 1 variance, 2 a-trous (each pass), 3 finalize, 4 TAA, 5 TAA finalize
void rt_launch_svgf_taa(const RtParams & p, int sample_index, hipStream_t stream, void (*mark)(void * user, int svgf_kernel, hipStream_t stream) = nullptr, void * user = nullptr);
void rt_launch_random(const RtParams & p, int dimension, const unsigned * pixel_indices, int count, unsigned bounce, unsigned sample_index, float2 * out, hipStream_t stream);
void rt_launch_integrate_luts(const RtParams & p, float * dielectric_dir_enter, float * dielectric_dir_leave, float * dielectric_enter, float * dielectric_leave,
                              float * conductor_dir, float * conductor, hipStream_t stream);
void rt_launch_pack_pixels(const RtParams & p, float4 * dst, int tile_pixels, int tile_first, int tile_stride, int tiles, hipStream_t stream);
void rt_launch_unpack_pixels(const RtParams & p, const float4 * src, int tile_pixels, int world, int tiles_per_rank, hipStream_t stream);
void rt_launch_pack_svgf(const RtParams & p, float4 * dst, int tile_pixels, int tile_first, int tile_stride, int tiles, hipStream_t stream);
void rt_launch_unpack_svgf(const RtParams & p, const float4 * src, int tile_pixels, int world, int tiles_per_rank, hipStream_t stream);
void rt_launch_stream_read(const float4 * src, size_t count, float * sink, hipStream_t stream);
// Counting variants: stats = 10 x u64 {closest: nodes, triangles, inst_xform, inst_ident, rays; shadow: same}
void rt_launch_trace_counting(const RtParams & p, int bounce, unsigned long long * stats, hipStream_t stream);
void rt_launch_trace_shadow_counting(const RtParams & p, int bounce, unsigned long long * stats, hipStream_t stream);
// Stand-alone trace on explicit ray arrays (rt_trace_rays / rt_trace_shadow_rays)
void rt_launch_trace_explicit(const RtParams & p, RtVec3SoA origin, RtVec3SoA direction, uint4 * hits, int ray_count, int * retired_counter, hipStream_t stream);
void rt_launch_trace_shadow_explicit(const RtParams & p, RtVec3SoA origin, RtVec3SoA direction, const float * max_distance, uint8_t * occluded, int ray_count, int * retired_counter, hipStream_t stream);
