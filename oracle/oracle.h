/*
 * oracle.h -- CPU restatement of the reference's device path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so; the product (gpu-raytracer_amd/) never links or calls it.
 *
 * Parity status: the reference ships no tests or golden vectors for this path and its
 * device code (CUDA + PTX, NVRTC) cannot run on a GPU here. The oracle is
 *   (i)   fed by scene data that the reference's own loaders and BVH builders, compiled verbatim
 *         (oracle/_ref/libref_scene.so, libref_bvh.so), reproduce bit for bit;
 *   (ii)  a line-by-line restatement of CUDA/Raytracing/{BVH8,BVH2,BVH4,Triangle,Mesh,BVH}.h,
 *         CUDA/{Camera,Sampling,Util,Buffers,AOV,Sky,Medium,RayCone,Material,BSDF,KullaConty}.h,
 *         CUDA/Pathtracer.cu and CUDA/SVGF/{SVGF,TAA}.h (each function cites its source);
 *   (iii) PINNED by those very sources: oracle/ref/ref_cuda_harness.cpp compiles
 *         Src/CUDA/Pathtracer.cu and all its headers verbatim for the host CPU and runs the
 *         reference's kernels thread by thread on the same inputs; tests/test_oracle.py
 *         (test_*reference*) requires identical queue sizes and frames within float noise.
 * Still "parity unpinned" (SURVEY.md 8c): NVIDIA's texture-unit filtering, which has no
 * definition to execute -- both sides use the software rules of oracle_shading.h.
 * (The AO integrator, AO.cu, is pinned the same way through oracle/_ref/libref_ao.so.)
 *
 * Arithmetic contract (shared with the HIP kernels so that traversal can be compared
 * bit for bit): IEEE fp32, no implicit contraction (-ffp-contract=off), explicit fmaf()
 * where a fused multiply-add is meant, IEEE division and sqrt.  The reference builds its
 * device code with --use_fast_math (CUDAModule.cpp:155), i.e. it leaves these choices to
 * the compiler; fixing them is what makes a bit-exact checker possible at all.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/gpu_raytracer_amd.h"   /* rt_gpu_config / rt_camera / enums: shared ABI structs */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_texture {
	const uint8_t * texels;     /* RGBA8 linear, all mip levels, level 0 first */
	int32_t width, height, mip_levels;
	int32_t lod_width, lod_height; /* size that enters the LOD bias; 0 = width / height (rt_texture_desc) */
} oracle_texture;

/* Flat views of exactly the arrays the device is given (see include/gpu_raytracer_amd.h). */
typedef struct oracle_scene {
	const float   * triangles;            /* 24 floats each, device layout                 */
	int32_t         triangle_count;
	const uint8_t * bvh8_nodes;           /* 80 B each (may be NULL when bvh_type == 2)     */
	const uint8_t * bvh2_nodes;           /* 32 B each (may be NULL when bvh_type == 8)     */
	const uint8_t * bvh4_nodes;           /* 128 B each (only needed when bvh_type == 4)    */
	int32_t         bvh_type;             /* 8, 4 or 2                                      */

	const int32_t * mesh_bvh_root_indices;
	const int32_t * mesh_material_ids;
	const float   * mesh_transforms;      /* 12 floats each */
	const float   * mesh_transforms_inv;
	const float   * mesh_transforms_prev;
	int32_t         mesh_count;

	const uint8_t * material_types;
	const float   * materials;            /* 8 floats each */
	int32_t         material_count;
	const float   * media;                /* 8 floats each */
	int32_t         medium_count;

	const oracle_texture * textures;
	int32_t         texture_count;

	const int32_t * light_triangle_indices;
	const float   * light_triangle_cumulative_probability;
	int32_t         light_triangle_count;
	const float   * light_mesh_cumulative_probability;
	const int32_t * light_mesh_triangle_span;     /* 2 per light mesh */
	const int32_t * light_mesh_transform_indices;
	int32_t         light_mesh_count;
	float           lights_total_weight;

	const float   * pmj_samples;          /* 64*4096*2 */
	const uint8_t * blue_noise;           /* 16*128*128*2 */

	/* Kulla-Conty energy-compensation tables (KullaConty.h:4-10), layouts i + r*16 + c*256 etc. */
	const float   * lut_dielectric_directional_albedo_enter; /* 16*16*16 */
	const float   * lut_dielectric_directional_albedo_leave;
	const float   * lut_dielectric_albedo_enter;             /* 16*16 */
	const float   * lut_dielectric_albedo_leave;
	const float   * lut_conductor_directional_albedo;        /* 32*32 */
	const float   * lut_conductor_albedo;                    /* 32 */

	const float   * sky;                  /* float4 equirect */
	int32_t         sky_width, sky_height;
	float           sky_scale;

	rt_camera       camera;
	rt_gpu_config   config;
	float           view_projection[16], view_projection_prev[16]; /* SVGF */

	int32_t screen_width, screen_height, screen_pitch;

	/* Flattened static geometry (rt_upload_triangle_aliases): one entry per triangle, or NULL. alias_mesh_ids[i] >= 0: triangle
	 * i is a copy, a closest hit on it is reported as instance alias_mesh_ids[i], triangle alias_triangle_ids[i].             */
	const int32_t * alias_mesh_ids;
	const int32_t * alias_triangle_ids;
	/* rt_set_static_geometry: node 0 is not a TLAS root but the root of ONE world-space tree that holds the whole scene;
	 * rays start inside it, as instance row 0. */
	int32_t static_whole_scene;
	/* rt_set_skip_behind_hit: closest-hit rays of a one-tree scene drop a stacked group of children whose bound (the least entry
	 * distance of the children left in it, 16 bits, rounded down) lies at or behind the hit already held. 0: the reference's walk. */
	int32_t skip_behind_hit;
} oracle_scene;

/* Per-ray work counters: define the ALGORITHMIC bytes of a trace:
 * 24 + 16 + 80*nodes + 48*triangles + 52*instances(non-identity) (+4 per identity entry). */
typedef struct oracle_trace_stats {
	uint64_t nodes, triangles, instances_transformed, instances_identity, rays;
} oracle_trace_stats;

/* Frame state owned by the caller: AOV framebuffer/accumulator pairs, pitch*height float4. */
typedef struct oracle_frame {
	float * framebuffer[RT_AOV_COUNT];   /* NULL = AOV disabled */
	float * accumulator[RT_AOV_COUNT];
	float * final_image;                 /* `accumulator` surface of Pathtracer.cu:24 */
	/* SVGF / TAA state (all pitch*height): */
	float * gbuffer_normal_and_depth;        /* float4 */
	int32_t * gbuffer_mesh_id_and_triangle_id; /* int2  */
	float * gbuffer_screen_position_prev;    /* float2 */
	float * frame_buffer_moment;             /* float4 */
	int32_t * history_length;
	float * history_direct, * history_indirect, * history_moment, * history_normal_and_depth; /* float4 */
	float * taa_frame_prev, * taa_frame_curr; /* float4 */
	float * scratch_direct, * scratch_indirect; /* extra ping-pong, float4 */
} oracle_frame;

typedef struct oracle_counters {
	int32_t trace[RT_MAX_BOUNCES], shadow[RT_MAX_BOUNCES];
	int32_t diffuse[RT_MAX_BOUNCES], plastic[RT_MAX_BOUNCES], dielectric[RT_MAX_BOUNCES], conductor[RT_MAX_BOUNCES];
	oracle_trace_stats trace_stats, shadow_stats;
} oracle_counters;

const char * oracle_version(void);

/* bvh8_trace / bvh2_trace (BVH8.h:113-274, BVH2.h:46-139): closest hit per ray.
 * hits: uint32[4] per ray {mesh_id, triangle_id, t bits, u16 | v16 << 16} (Buffers.h:25-32). */
void oracle_trace(const oracle_scene * scene, const float * ox, const float * oy, const float * oz,
                  const float * dx, const float * dy, const float * dz, size_t ray_count,
                  uint32_t * hits, oracle_trace_stats * stats, int threads);
/* bvh8_trace_shadow / bvh2_trace_shadow (BVH8.h:276-444): any hit closer than max_distance. */
void oracle_trace_shadow(const oracle_scene * scene, const float * ox, const float * oy, const float * oz,
                         const float * dx, const float * dy, const float * dz, const float * max_distance,
                         size_t ray_count, uint8_t * occluded, oracle_trace_stats * stats, int threads);
/* Threads a call with threads <= 0 uses: ORACLE_THREADS, else min(omp_get_max_threads(), 32). (A GPU box shows 256 logical
 * CPUs to the container and gives it about 16 cores' worth: the traversal loop peaks at 16-32 threads, 8 Mrays/s, and
 * falls to 3.5 Mrays/s at 256.) */
int oracle_default_threads(void);
/* kernel_generate (Pathtracer.cu:122-139) */
void oracle_generate(const oracle_scene * scene, int sample_index, int pixel_offset, int pixel_count,
                     float * ox, float * oy, float * oz, float * dx, float * dy, float * dz, uint32_t * pixel_index_and_flags);
/* random<Dim> (Sampling.h:44-84) */
void oracle_random(const oracle_scene * scene, int dimension, const uint32_t * pixel_indices, size_t count,
                   uint32_t bounce, uint32_t sample_index, float * out_xy);
/* Pathtracer::render for one sample over pixels [pixel_offset, pixel_offset+pixel_count)
 * (Pathtracer.cpp:738-855): batches, bounces, accumulate or SVGF/TAA. */
void oracle_render_sample(const oracle_scene * scene, oracle_frame * frame, int sample_index,
                          int pixel_offset, int pixel_count, oracle_counters * counters, int threads);
/* The two halves of an SVGF frame under the multi-GPU tile split: path tracing of a pixel range that leaves the per-frame
 * AOVs and g-buffers in place, and the filter stage over the whole frame (+ aovs_clear_to_zero). */
void oracle_render_sample_unfiltered(const oracle_scene * scene, oracle_frame * frame, int sample_index,
                                     int pixel_offset, int pixel_count, oracle_counters * counters, int threads);
void oracle_filter_frame(const oracle_scene * scene, oracle_frame * frame, int sample_index);
/* AO::render for one sample (Integrators/AO.cpp:148-200, CUDA/AO.cu): primary hit -> one cosine-
 * weighted occlusion ray of length ao_radius -> RADIANCE = 1 where it escapes; NORMAL / POSITION AOVs. */
void oracle_render_ao_sample(const oracle_scene * scene, oracle_frame * frame, int sample_index, float ao_radius,
                             int pixel_offset, int pixel_count, oracle_counters * counters, int threads);
/* Kulla-Conty LUT integration kernels (KullaConty.h:83-240); num_samples = 100000 in the reference. */
/* kernel_integrate_dielectric / kernel_integrate_conductor (KullaConty.h:83-148,179-226) for the
 * LUT cells [first_cell, first_cell + cell_count) (thread_index of the CUDA kernel); num_samples
 * = 100000 in the reference. entering: 1 = air -> material. */
void oracle_integrate_dielectric_cells(const oracle_scene * scene, int entering, int num_samples, int first_cell, int cell_count, float * out, int threads);
void oracle_integrate_conductor_cells(const oracle_scene * scene, int num_samples, int first_cell, int cell_count, float * out, int threads);
/* kernel_average_dielectric / kernel_average_conductor (KullaConty.h:150-165,228-240) */
void oracle_average_dielectric(const float * directional_16x16x16, float * out_16x16);
void oracle_average_conductor(const float * directional_32x32, float * out_32);

#ifdef __cplusplus
}
#endif
#endif
