// kernels_shade.hip -- ray generation, the per-bounce "sort" dispatch kernel and the four
// material kernels with next-event estimation.
//
// Replaces kernel_generate, kernel_sort and kernel_material_{diffuse,plastic,dielectric,
// conductor} of the reference (CUDA/Pathtracer.cu:122-139,199-463,465-773, CUDA/Camera.h:20-62,
// CUDA/BSDF.h).  Differences that are design, not behaviour:
//   * every queue append (next ray, shadow ray, material queue) is ONE returning atomic per
//     64-lane wave (wave_aggregated_append) instead of one per thread;
//   * kernels are grid-stride loops over a fixed grid sized to the machine, so that the deep,
//     nearly empty bounces do not launch ~3000 idle workgroups;
//   * albedo / sky / LUT fetches go through the software texture unit in rt_shading.h.
#include "rt_shading.h"

#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 256
#endif
#ifndef RT_SHADE_WAVES
// The material kernels wait on scattered triangle / texture / light-table reads; left alone the compiler uses 129-140 VGPRs
// (3 waves per SIMD). Asking for 4 waves caps them at 128 registers: one more wave to hide the latency behind.
#define RT_SHADE_WAVES 4
#endif
#ifndef RT_SHADE_ONE_APPEND
#define RT_SHADE_ONE_APPEND 1   // 1: the shadow-ray and the continuation-ray append of a shade round share barriers and atomic latency (block_bucketed_append2)
#endif
#ifndef RT_OCTANT_BUCKETS
#define RT_OCTANT_BUCKETS 1   // a shade workgroup appends its rays ordered by direction octant (rt_math.h: block_bucketed_append)
#endif
#if RT_OCTANT_BUCKETS
#define RT_RAY_BUCKET(d) direction_octant(d)
#else
#define RT_RAY_BUCKET(d) 0u
#endif
#ifndef RT_SORT_BLOCK
#define RT_SORT_BLOCK 1024  // kernel_sort: 16 waves share one atomic per material queue (256: 0.230 ms per step, 512: 0.143, 1024: 0.138; profiles/r04_shade_stage.txt)
#endif
#ifndef RT_SORT_WAVES
// Waves per SIMD asked of the sort kernels' register allocation. The kernel is a chain of dependent gathers (queue entry -> slot table -> sample
// tables of the roulette; hit -> instance material -> material type) at a fifth of the vector ALUs: what it needs is rays in flight. 8 = 64
// registers (the general form takes 100: one 1024-thread workgroup per CU, 4 waves per SIMD), two workgroups per CU; the 128 bytes of scratch
// that costs sit in the branches Sponza never takes (media, emitters seen by a BSDF ray). Measured, one box: sort 0.135 -> 0.103 ms per step
// (profiles/r04_shade_stage.txt, 5.); 512-thread workgroups at 6 waves (80 registers): 0.165.
#define RT_SORT_WAVES 8
#endif

struct HitInfo { float t, u, v; int mesh_id, triangle_id; };

RT_DEV HitInfo unpack_hit(uint4 h) { // Buffers.h:34-48
	HitInfo r;
	r.mesh_id = int(h.x); r.triangle_id = int(h.y);
	r.t = __uint_as_float(h.z);
	r.u = float(h.w & 0xffffu) / 65535.0f;
	r.v = float(h.w >> 16)     / 65535.0f;
	return r;
}

// ---- kernel_generate ----------------------------------------------------------------------------

// `taa_index`: the sample index that picks the TAA jitter (the merged wavefront passes a sample index shifted by the slot,
// which random_sample undoes).
RT_DEV void camera_generate_ray(const RtParams & p, int pixel_index, int sample_index, int x, int y, f3 & origin, f3 & direction, int taa_index) {
	f2 rand_filter   = random_sample(p, DIM_FILTER,   unsigned(pixel_index), 0, unsigned(sample_index));
	// (a pinhole camera -- aperture radius 0, uniform over the launch -- multiplies its lens sample by zero: the draw and the disk mapping are skipped, the ray is the same)
	const bool thin_lens = p.camera.aperture_radius != 0.0f;
	f2 rand_aperture = thin_lens ? random_sample(p, DIM_APERTURE, unsigned(pixel_index), 0, unsigned(sample_index)) : mk2(0.0f, 0.0f);

	f2 jitter;
	if (p.config.enable_svgf) {
		const float taa_halton_x[4] = { 0.3f, 0.7f, 0.2f, 0.8f };
		const float taa_halton_y[4] = { 0.2f, 0.8f, 0.7f, 0.3f };
		jitter.x = taa_halton_x[taa_index & 3];
		jitter.y = taa_halton_y[taa_index & 3];
	} else if (p.config.reconstruction_filter == RT_FILTER_BOX) {
		jitter = rand_filter;
	} else if (p.config.reconstruction_filter == RT_FILTER_TENT) {
		jitter.x = sample_tent(rand_filter.x);
		jitter.y = sample_tent(rand_filter.y);
	} else {
		f2 g = sample_gaussian(rand_filter.x, rand_filter.y);
		jitter.x = 0.5f + 0.5f * g.x;
		jitter.y = 0.5f + 0.5f * g.y;
	}
	float x_jittered = float(x) + jitter.x;
	float y_jittered = float(y) + jitter.y;

	f3 blc = mk3(p.camera.bottom_left_corner[0], p.camera.bottom_left_corner[1], p.camera.bottom_left_corner[2]);
	f3 xa  = mk3(p.camera.x_axis[0], p.camera.x_axis[1], p.camera.x_axis[2]);
	f3 ya  = mk3(p.camera.y_axis[0], p.camera.y_axis[1], p.camera.y_axis[2]);

	f3 focal_point = p.camera.focal_distance * normalize(blc + x_jittered * xa + y_jittered * ya);
	f2 lens_point  = thin_lens ? p.camera.aperture_radius * sample_disk(rand_aperture.x, rand_aperture.y) : mk2(0.0f, 0.0f);

	f3 offset = xa * lens_point.x + ya * lens_point.y;
	direction = normalize(focal_point - offset);
	origin = mk3(p.camera.position[0], p.camera.position[1], p.camera.position[2]) + offset;
}

__global__ void __launch_bounds__(RT_SHADE_BLOCK) kernel_generate(RtParams p, int sample_index, int pixel_offset, int pixel_count) {
	const int ray_count = pixel_count * p.batch_samples; // sample s of the batch occupies queue entries [s * pixel_count, (s+1) * pixel_count)
	if (blockIdx.x == 0 && threadIdx.x == 0) p.sizes->trace[0] = ray_count; // BufferSizes::reset (Pathtracer.h:143)
	for (int index = blockIdx.x * blockDim.x + threadIdx.x; index < ray_count; index += gridDim.x * blockDim.x) {
		int sample_in_batch = index / pixel_count;
		int index_offset = rt_map_pixel(p, index - sample_in_batch * pixel_count + pixel_offset);
		int x = index_offset % p.screen_width;
		int y = index_offset / p.screen_width;
		int pixel_index = x + y * p.screen_pitch;
		unsigned virtual_pixel = unsigned(sample_in_batch) * p.frame_pixels + unsigned(pixel_index);

		f3 origin, direction;
		camera_generate_ray(p, int(virtual_pixel), sample_index, x, y, origin, direction, sample_index + sample_in_batch);

		store3(p.trace[0].origin,    index, origin);
		store3(p.trace[0].direction, index, direction);
		p.trace[0].pixel_index_and_flags[index] = virtual_pixel;
	}
}

// Merged wavefront: the primary rays of a new submission join the trace queue of the current iteration behind
// whatever the previous iteration's sort / shade kernels appended (kernel_stream_advance adds the count afterwards).
// Sample s of the submission renders into sample slot slot_base + s.
// `queue_offset`: rays generated by earlier submissions of the same iteration (several small submissions can enter together).
// `block_width` / `band_rows` (0: scan lines, the reference's order): the ORDER of a sample's primary rays in the queue. A pixel's ray does not
// depend on its place in the queue (the random numbers are keyed on the pixel index), but the traversal launch deals
// consecutive rays to the lanes of one wave and the sort / shade kernels keep neighbours together: the frame is walked in bands of
// `band_rows` scan lines, a band in blocks of `block_width` columns, a block row by row -- 64 consecutive rays cover an 8 x 8 patch
// of the screen instead of a 64 x 1 strip. The host asks for it only when the pixel list is made of whole bands (rt_api.hip: stream_generate).
__global__ void __launch_bounds__(RT_SHADE_BLOCK) kernel_generate_stream(RtParams p, int sample_index, int pixel_offset, int pixel_count, int slot_base, int queue_offset, int block_width, int band_rows) {
	const int ray_count = pixel_count * p.batch_samples;
	const RtTraceBuffer & out = p.trace[p.stream_iteration & 1];
	const int base = p.stream->trace_count[p.stream_iteration & 1] + queue_offset;
	for (int index = blockIdx.x * blockDim.x + threadIdx.x; index < ray_count; index += gridDim.x * blockDim.x) {
		int sample_in_batch = index / pixel_count;
		int index_offset = rt_map_pixel(p, index - sample_in_batch * pixel_count + pixel_offset);
		int x = index_offset % p.screen_width;
		int y = index_offset / p.screen_width;
		if (band_rows > 0) {
			int band = y / band_rows;
			int rows = min(band_rows, p.screen_height - band * band_rows);   // the last band may be lower
			int in_band = index_offset - band * band_rows * p.screen_width;
			if (band_rows == 8 && block_width == 8 && rows == 8 && (in_band >> 6) * 8 + 8 <= p.screen_width) {   // a whole 8 x 8 patch (the shipped shape): shifts
				x = (in_band >> 6) * 8 + (in_band & 7);
				y = band * 8 + ((in_band >> 3) & 7);
			} else {
				int block = in_band / (block_width * rows);
				int in_block = in_band - block * block_width * rows;
				int columns = min(block_width, p.screen_width - block * block_width); // the last block may be narrower
				x = block * block_width + in_block % columns;
				y = band * band_rows + in_block / columns;
			}
		}
		int pixel_index = x + y * p.screen_pitch;
		unsigned slot = unsigned(slot_base + sample_in_batch);
		unsigned virtual_pixel = slot * p.frame_pixels + unsigned(pixel_index);

		f3 origin, direction;
		camera_generate_ray(p, int(virtual_pixel), sample_index + sample_in_batch - int(slot), x, y, origin, direction, sample_index + sample_in_batch); // random_sample adds the slot back

		store3(out.origin,    base + index, origin);
		store3(out.direction, base + index, direction);
		out.pixel_index_and_flags[base + index] = virtual_pixel;
	}
}

// Bookkeeping between the iterations of the merged wavefront, one thread: the generated rays are counted in, the
// queues this iteration appends to start empty, the cursors of its trace launch are reset, and the host gets the
// size of the wavefront (it bounds what is in flight with it before admitting the next submission).
// ... and the statistics rows of the submissions that join with this iteration start at zero (reset_ring_count rows from reset_ring_first, modulo the ring).
__global__ void __launch_bounds__(256) kernel_stream_advance(RtStreamControl * control, int iteration, int generated, volatile int * progress, int reset_ring_first, int reset_ring_count) {
	for (int k = int(threadIdx.x); k < reset_ring_count * RT_STAT_KINDS * RT_MAX_BOUNCES; k += int(blockDim.x))
		(&control->stats[(reset_ring_first + k / (RT_STAT_KINDS * RT_MAX_BOUNCES)) % RT_STREAM_SUBMISSIONS][0][0])[k % (RT_STAT_KINDS * RT_MAX_BOUNCES)] = 0;
	const int q = iteration & 1;
	for (int k = int(threadIdx.x); k < 2 * RT_ENDGAME_MAX_WAVES; k += int(blockDim.x)) (&control->endgame[q][0][0])[k] = 0;   // the region cursors of this iteration's traversal launch
	if (threadIdx.x != 0) return;
	int total = control->trace_count[q] + generated;
	control->trace_count[q] = total;
	control->trace_count[q ^ 1] = 0;
	for (int m = 0; m < 4; m++) control->material_count[m] = 0;
	control->shadow_count[q] = 0;
	control->cursor[q][0] = 0; control->cursor[q][1] = 0;
	if (progress) { progress[1] = total; __threadfence_system(); progress[0] = iteration; }
}

// rt_upload_textures, RT_TEXTURE_BC1_EXPANDED: one thread per texel, the same bc1_texel the per-fetch path runs.
__global__ void kernel_expand_bc1(const uint2 * __restrict__ blocks, uchar4 * __restrict__ texels, size_t texel_count) {
	for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < texel_count; i += size_t(gridDim.x) * blockDim.x)
		texels[i] = bc1_texel(blocks[i >> 4], int(i & 3), int((i >> 2) & 3));
}
void rt_launch_expand_bc1(const uint2 * blocks, uchar4 * texels, size_t block_count, hipStream_t stream) {
	if (block_count == 0) return;
	size_t texel_count = block_count * 16;
	size_t groups = (texel_count + 255) / 256;
	unsigned grid = unsigned(groups < 65536 ? groups : 65536);
	hipLaunchKernelGGL(kernel_expand_bc1, dim3(grid), dim3(256), 0, stream, blocks, texels, texel_count);
}

__global__ void kernel_random(RtParams p, int dimension, const unsigned * pixel_indices, int count, unsigned bounce, unsigned sample_index, float2 * out) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	f2 r = random_sample(p, dimension, pixel_indices[i], bounce, sample_index);
	out[i] = make_float2(r.x, r.y);
}

// ---- kernel_sort -----------------------------------------------------------------------------------

// Returns true if the path terminates (Pathtracer.cu:199-218)
RT_DEV bool russian_roulette(const RtParams & p, int pixel_index, int bounce, int sample_index, f3 & throughput) {
	if (bounce == p.config.num_bounces - 1) return true;
	if (p.config.enable_russian_roulette && bounce > 0) {
		f3 t = throughput;
		if (p.config.enable_svgf) t *= mk3(aov_get(p, RT_AOV_ALBEDO, pixel_index));
		float survival_probability = saturate(fmaxf(fmaxf(t.x, t.y), t.z));
		float r = random_sample(p, DIM_RUSSIAN_ROULETTE, unsigned(pixel_index), unsigned(bounce), unsigned(sample_index)).x;
		if (r > survival_probability) return true;
		throughput /= survival_probability;
	}
	return false;
}

RT_DEV void add_radiance(const RtParams & p, int bounce, int pixel_index, f3 illumination, f3 bounce0_value) {
	if (bounce == 0) {
		aov_set(p, RT_AOV_ALBEDO,          pixel_index, mk4(1.0f));
		aov_set(p, RT_AOV_RADIANCE,        pixel_index, mk4(bounce0_value));
		aov_set(p, RT_AOV_RADIANCE_DIRECT, pixel_index, mk4(bounce0_value));
	} else if (bounce == 1) {
		aov_add(p, RT_AOV_RADIANCE,        pixel_index, mk4(illumination));
		aov_add(p, RT_AOV_RADIANCE_DIRECT, pixel_index, mk4(illumination));
	} else {
		aov_add(p, RT_AOV_RADIANCE,          pixel_index, mk4(illumination));
		aov_add(p, RT_AOV_RADIANCE_INDIRECT, pixel_index, mk4(illumination));
	}
}

struct TriangleFull {
	f3 position_0, position_edge_1, position_edge_2;
	f3 normal_0, normal_edge_1, normal_edge_2;
	f2 tex_coord_0, tex_coord_edge_1, tex_coord_edge_2;
};
RT_DEV void triangle_get_positions(const RtParams & p, int index, f3 & p0, f3 & e1, f3 & e2) {
	const float4 * t = p.triangles + size_t(index) * 6;
	float4 a = t[0], b = t[1], c = t[2];
	p0 = mk3(a.x, a.y, a.z); e1 = mk3(a.w, b.x, b.y); e2 = mk3(b.z, b.w, c.x);
}
RT_DEV TriangleFull triangle_get_full(const RtParams & p, int index) {
	const float4 * t = p.triangles + size_t(index) * 6;
	float4 a = t[0], b = t[1], c = t[2], d = t[3], e = t[4], f = t[5];
	TriangleFull r;
	r.position_0 = mk3(a.x, a.y, a.z); r.position_edge_1 = mk3(a.w, b.x, b.y); r.position_edge_2 = mk3(b.z, b.w, c.x);
	r.normal_0 = mk3(c.y, c.z, c.w); r.normal_edge_1 = mk3(d.x, d.y, d.z); r.normal_edge_2 = mk3(d.w, e.x, e.y);
	r.tex_coord_0 = mk2(e.z, e.w); r.tex_coord_edge_1 = mk2(f.x, f.y); r.tex_coord_edge_2 = mk2(f.z, f.w);
	return r;
}
RT_DEV f3 barycentric(float u, float v, f3 base, f3 e1, f3 e2) { return base + u * e1 + v * e2; }
RT_DEV f2 barycentric(float u, float v, f2 base, f2 e1, f2 e2) { return base + u * e1 + v * e2; }

RT_DEV f3 m_position(const float4 * m, f3 v) {
	float4 r0 = m[0], r1 = m[1], r2 = m[2];
	return mk3(r0.x * v.x + r0.y * v.y + r0.z * v.z + r0.w, r1.x * v.x + r1.y * v.y + r1.z * v.z + r1.w, r2.x * v.x + r2.y * v.y + r2.z * v.z + r2.w);
}
RT_DEV f3 m_direction(const float4 * m, f3 v) {
	float4 r0 = m[0], r1 = m[1], r2 = m[2];
	return mk3(r0.x * v.x + r0.y * v.y + r0.z * v.z, r1.x * v.x + r1.y * v.y + r1.z * v.z, r2.x * v.x + r2.y * v.y + r2.z * v.z);
}

// SVGF g-buffers (CUDA/SVGF/SVGF.h:61-84)
RT_DEV f2 oct_encode_normal(f3 n) {
	n /= (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
	if (n.z < 0.0f) {
		n.x = (1.0f - fabsf(n.y)) * (n.x >= 0.0f ? +1.0f : -1.0f);
		n.y = (1.0f - fabsf(n.x)) * (n.y >= 0.0f ? +1.0f : -1.0f);
	}
	return mk2(0.5f + 0.5f * n.x, 0.5f + 0.5f * n.y);
}
RT_DEV f4 mat4_mul(const float * m, f4 v) {
	return mk4(
		m[ 0] * v.x + m[ 1] * v.y + m[ 2] * v.z + m[ 3] * v.w,
		m[ 4] * v.x + m[ 5] * v.y + m[ 6] * v.z + m[ 7] * v.w,
		m[ 8] * v.x + m[ 9] * v.y + m[10] * v.z + m[11] * v.w,
		m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w);
}
RT_DEV void svgf_set_gbuffers(const RtParams & p, int x, int y, const HitInfo & hit, f3 hit_point, f3 normal, f3 hit_point_prev) {
	f4 u_curr = mat4_mul(p.view_projection,      mk4(hit_point.x, hit_point.y, hit_point.z, 1.0f));
	f4 u_prev = mat4_mul(p.view_projection_prev, mk4(hit_point_prev.x, hit_point_prev.y, hit_point_prev.z, 1.0f));
	f2 oct = oct_encode_normal(normal);
	int idx = x + y * p.screen_pitch;
	p.gbuffer_normal_and_depth[idx] = make_float4(oct.x, oct.y, u_curr.z, u_prev.z);
	p.gbuffer_mesh_id_and_triangle_id[idx] = make_int2(hit.mesh_id, hit.triangle_id);
	p.gbuffer_screen_position_prev[idx] = make_float2(u_prev.x / u_prev.w, u_prev.y / u_prev.w);
}

RT_DEV void material_queue_store(const RtParams & p, int slot, int index_out, int bounce, f3 ray_direction, int medium_id, float cone_angle, float cone_width, uint4 hit, int pixel_index, f3 throughput) {
	const RtMaterialBuffer & q = p.material[slot];
	store3(q.direction, index_out, ray_direction);
	if (medium_id != RT_INVALID) q.medium[index_out] = medium_id;
	if (bounce > 0 && p.config.enable_mipmapping) { q.cone_angle[index_out] = cone_angle; q.cone_width[index_out] = cone_width; }
	q.hits[index_out] = hit;
	unsigned flags = unsigned(medium_id != RT_INVALID) << 30;
	q.pixel_index_and_flags[index_out] = unsigned(pixel_index) | flags;
	if (bounce > 0) store3(q.throughput, index_out, throughput);
}

// ---- per-submission statistics of the merged wavefront ------------------------------------------------------
// Rays per (submission, queue kind, bounce), what rt_get_counters reports. A workgroup tallies in LDS -- per wave one
// LDS add for every distinct submission among its lanes (queues are mostly sorted by submission, so usually one) --
// and flushes its non-zero cells with one global add each when it is done.
struct StreamStatsLDS { int count[RT_STREAM_SUBMISSIONS][RT_STAT_KINDS]; };

RT_DEV void stream_stats_clear(StreamStatsLDS & lds) {
	for (int i = threadIdx.x; i < RT_STREAM_SUBMISSIONS * RT_STAT_KINDS; i += blockDim.x) (&lds.count[0][0])[i] = 0;
	__syncthreads();
}
// called in converged or diverged flow by any subset of lanes: lanes with `counted` add one to (submission, kind)
RT_DEV void stream_stats_add(StreamStatsLDS & lds, bool counted, int submission, int kind) {
	unsigned long long remaining = __ballot(counted);
	while (remaining) {
		int leader = __ffsll((long long)remaining) - 1;
		int s = __shfl(submission, leader);
		unsigned long long same = __ballot(counted && submission == s) & remaining;
		if (int(lane_id()) == leader) atomicAdd(&lds.count[s][kind], __popcll(same));
		remaining &= ~same;
	}
}
RT_DEV void stream_stats_flush(const RtParams & p, StreamStatsLDS & lds) {
	__syncthreads();
	for (int i = threadIdx.x; i < RT_STREAM_SUBMISSIONS * RT_STAT_KINDS; i += blockDim.x) {
		int n = (&lds.count[0][0])[i];
		if (n == 0) continue;
		int submission = i / RT_STAT_KINDS, kind = i % RT_STAT_KINDS;
		int bounce = p.stream_iteration - p.stream_table->submission_birth[submission];
		if (bounce >= 0 && bounce < RT_MAX_BOUNCES) atomicAdd(&p.stream->stats[submission][kind][bounce], n);
	}
}

// MERGED: the launch processes the trace queue of iteration p.stream_iteration, whose entries are at different
// bounces and belong to different samples (rt_stream_path_info); the launch arguments are ignored.
template<bool MERGED>
RT_DEV void sort_rays(const RtParams & p, int launch_bounce, int launch_sample_index) {
	const int q = MERGED ? (p.stream_iteration & 1) : (launch_bounce & 1);
	const int ray_count = MERGED ? p.stream->trace_count[q] : p.sizes->trace[launch_bounce];
	const RtTraceBuffer & in  = p.trace[q];
	const RtTraceBuffer & out = p.trace[q ^ 1];
	__shared__ BlockAppendLDS<4, RT_SORT_BLOCK / RT_WAVE_SIZE> append_lds;
	__shared__ StreamStatsLDS stats_lds;
	int * const material_counters[4] = {
		MERGED ? &p.stream->material_count[0] : &p.sizes->diffuse[launch_bounce],    MERGED ? &p.stream->material_count[1] : &p.sizes->plastic[launch_bounce],
		MERGED ? &p.stream->material_count[2] : &p.sizes->dielectric[launch_bounce], MERGED ? &p.stream->material_count[3] : &p.sizes->conductor[launch_bounce] };
	int * const next_trace_counter = MERGED ? &p.stream->trace_count[q ^ 1] : &p.sizes->trace[launch_bounce + 1];
	if (MERGED) stream_stats_clear(stats_lds);

	// every thread of the workgroup makes the same number of rounds (block_aggregated_append has barriers)
	for (int first = blockIdx.x * blockDim.x; first < ray_count; first += gridDim.x * blockDim.x) {
		const int index = first + int(threadIdx.x);
		f3 ray_direction = mk3(0.0f); uint4 packed_hit = make_uint4(0, 0, 0, 0);
		float ray_cone_angle = 0.0f, ray_cone_width = 0.0f;
		int pixel_index = 0, medium_id = RT_INVALID;
		f3 throughput = mk3(1.0f);
		int bounce = launch_bounce, sample_index = launch_sample_index, submission = 0;

		// classify one ray: the material queue it continues in, or -1 (missed, hit a light, scattered, terminated)
		auto classify = [&]() -> int {
		if (index >= ray_count) return -1;
		unsigned pixel_index_and_flags = in.pixel_index_and_flags[index];
		pixel_index = int(pixel_index_and_flags & ~RT_FLAGS_ALL);
		bool first_of_submission = true;
		if (MERGED) {
			RtPathInfo info = rt_stream_path_info(p, unsigned(pixel_index));
			bounce = info.bounce; sample_index = int(info.sample_index_for_rng); submission = info.submission; first_of_submission = info.first_of_submission;
		}
		ray_direction = load3(in.direction, index);
		packed_hit = in.hits[index];
		HitInfo hit = unpack_hit(packed_hit);

		// Merged wavefront: the bounce of an entry comes out of the slot table, i.e. behind two dependent loads; the entry's other fields
		// are fetched beside them, not after them (every queue array has a slot for every entry; what bounce 0 never wrote is not used).
		if (MERGED ? p.config.enable_mipmapping : (bounce > 0 && p.config.enable_mipmapping)) {
			float angle = in.cone_angle[index], width = in.cone_width[index];
			if (bounce > 0) { ray_cone_angle = angle; ray_cone_width = width; }
		}

		int x = pixel_index % p.screen_pitch;
		int y = pixel_index / p.screen_pitch;

		bool allow_nee     = pixel_index_and_flags & RT_FLAG_ALLOW_NEE;
		bool inside_medium = pixel_index_and_flags & RT_FLAG_INSIDE_MEDIUM;

		if (MERGED) { f3 carried = load3(in.throughput, index); throughput = bounce == 0 ? mk3(1.0f) : carried; }
		else throughput = bounce == 0 ? mk3(1.0f) : load3(in.throughput, index);

		if (inside_medium) {
			medium_id = in.medium[index];
			HomogeneousMedium medium = medium_as_homogeneous(p, medium_id);
			bool medium_can_scatter = (medium.sigma_s.x + medium.sigma_s.y + medium.sigma_s.z) > 0.0f;
			if (medium_can_scatter) {
				f2 rand_scatter = random_sample(p, DIM_BSDF_0, unsigned(pixel_index), unsigned(bounce), unsigned(sample_index));
				f2 rand_phase   = random_sample(p, DIM_BSDF_1, unsigned(pixel_index), unsigned(bounce), unsigned(sample_index));
				f3 sigma_t = medium.sigma_a + medium.sigma_s;

				float throughput_sum = throughput.x + throughput.y + throughput.z;
				f3 wavelength_pdf = throughput / throughput_sum;

				float sigma_t_used;
				if      (rand_scatter.x * throughput_sum < throughput.x)                sigma_t_used = sigma_t.x;
				else if (rand_scatter.x * throughput_sum < throughput.x + throughput.y) sigma_t_used = sigma_t.y;
				else                                                                   sigma_t_used = sigma_t.z;

				float scatter_distance = sample_exp(sigma_t_used, rand_scatter.y);
				f3 transmittance = beer_lambert(sigma_t, fminf(scatter_distance, hit.t));

				if (scatter_distance < hit.t) {
					f3 pdf = wavelength_pdf * sigma_t * transmittance;
					throughput *= medium.sigma_s * transmittance / (pdf.x + pdf.y + pdf.z);
					if (russian_roulette(p, pixel_index, bounce, sample_index, throughput)) return -1;

					f3 direction_out = sample_henyey_greenstein(-ray_direction, medium.g, rand_phase.x, rand_phase.y);
					f3 origin_out = load3(in.origin, index) + scatter_distance * ray_direction;

					int index_out = wave_aggregated_append(next_trace_counter);
					store3(out.origin,    index_out, origin_out);
					store3(out.direction, index_out, direction_out);
					out.medium[index_out] = medium_id;
					if (p.config.enable_mipmapping) {
						if (bounce == 0) { ray_cone_angle = p.camera.pixel_spread_angle; ray_cone_width = p.camera.pixel_spread_angle * scatter_distance; }
						out.cone_angle[index_out] = ray_cone_angle;
						out.cone_width[index_out] = ray_cone_width;
					}
					out.pixel_index_and_flags[index_out] = unsigned(pixel_index) | RT_FLAG_INSIDE_MEDIUM;
					store3(out.throughput, index_out, throughput);
					return -1;
				} else {
					f3 pdf = wavelength_pdf * transmittance;
					throughput *= transmittance / (pdf.x + pdf.y + pdf.z);
				}
			} else {
				throughput *= beer_lambert(medium.sigma_a, hit.t);
			}
		}

		if (hit.triangle_id == RT_INVALID) { // miss: sky
			f3 illumination = throughput * sample_sky(p, ray_direction);
			add_radiance(p, bounce, pixel_index, illumination, illumination);
			return -1;
		}

		if (bounce == 0 && first_of_submission && p.pixel_query_pixel == (MERGED ? int(unsigned(pixel_index) % p.frame_pixels) : pixel_index)) { // Pathtracer.cu:345-348 (sample 0 of a batch: virtual == real index)
			p.pixel_query_out[0] = hit.mesh_id;
			p.pixel_query_out[1] = hit.triangle_id;
		}

		int material_id = p.mesh_material_ids[hit.mesh_id];
		int material_type = p.material_types[material_id];

		if (material_type == RT_MATERIAL_LIGHT) {
			f3 p0, e1, e2;
			triangle_get_positions(p, hit.triangle_id, p0, e1, e2);
			f3 light_point = barycentric(hit.u, hit.v, p0, e1, e2);
			f3 light_point_prev = light_point;
			f3 light_geometric_normal = cross(e1, e2);

			const float4 * world = p.mesh_transforms + size_t(hit.mesh_id) * 3;
			light_point = m_position(world, light_point);
			light_geometric_normal = normalize(m_direction(world, light_geometric_normal));

			if (bounce == 0 && p.config.enable_svgf) {
				light_point_prev = m_position(p.mesh_transforms_prev + size_t(hit.mesh_id) * 3, light_point_prev);
				svgf_set_gbuffers(p, x, y, hit, light_point, light_geometric_normal, light_point_prev);
			}

			f3 emission = mk3(p.materials[2 * material_id]);

			bool count_light = p.config.enable_next_event_estimation ? !allow_nee : true;
			if (count_light) {
				add_radiance(p, bounce, pixel_index, throughput * emission, emission);
				return -1;
			}
			if (p.config.enable_multiple_importance_sampling) {
				float cos_theta_light = abs_dot(ray_direction, light_geometric_normal);
				float distance_to_light_squared = hit.t * hit.t;
				float brdf_pdf = in.last_pdf[index];
				float light_power = luminance(emission.x, emission.y, emission.z);
				float light_pdf = light_power * distance_to_light_squared / (cos_theta_light * p.lights_total_weight);
				if (!pdf_is_valid(light_pdf)) return -1;
				float mis_weight = power_heuristic(brdf_pdf, light_pdf);
				f3 illumination = throughput * emission * mis_weight;
				aov_add(p, RT_AOV_RADIANCE, pixel_index, mk4(illumination));
				if (bounce == 1) aov_add(p, RT_AOV_RADIANCE_DIRECT,   pixel_index, mk4(illumination));
				else             aov_add(p, RT_AOV_RADIANCE_INDIRECT, pixel_index, mk4(illumination));
			}
			return -1;
		}

		if (russian_roulette(p, pixel_index, bounce, sample_index, throughput)) return -1;

		switch (material_type) {
			case RT_MATERIAL_DIFFUSE:    return 0;
			case RT_MATERIAL_PLASTIC:    return 1;
			case RT_MATERIAL_DIELECTRIC: return 2;
			case RT_MATERIAL_CONDUCTOR:  return 3;
		}
		return -1;
		};

		int slot = classify();
		if (MERGED) {
			stream_stats_add(stats_lds, index < ray_count, submission, RT_STAT_TRACE);
			#pragma unroll
			for (int m = 0; m < 4; m++) stream_stats_add(stats_lds, slot == m, submission, RT_STAT_DIFFUSE + m);
		}
		int index_out = block_aggregated_append(slot, material_counters, append_lds);
		if (slot >= 0) material_queue_store(p, slot, index_out, bounce, ray_direction, medium_id, ray_cone_angle, ray_cone_width, packed_hit, pixel_index, throughput);
	}
	if (MERGED) stream_stats_flush(p, stats_lds);
}

__global__ void __launch_bounds__(RT_SORT_BLOCK, RT_SORT_WAVES) kernel_sort(RtParams p, int bounce, int sample_index) { sort_rays<false>(p, bounce, sample_index); }
__global__ void __launch_bounds__(RT_SORT_BLOCK, RT_SORT_WAVES) kernel_sort_stream(RtParams p) { sort_rays<true>(p, 0, 0); }

// ---- BSDFs (CUDA/BSDF.h) ----------------------------------------------------------------------------------

struct TextureLOD { f2 gradient_1, gradient_2; float lod; };

template<bool COMPRESSED>
RT_DEV f3 sample_albedo(const RtParams & p, int bounce, f3 diffuse, int texture_id, const RtTexture & tex, f2 tex_coord, const TextureLOD & lod) { // RayCone.h:19-29
	if (texture_id == RT_INVALID) return diffuse;
	if (p.config.enable_mipmapping) {
		if (bounce == 0) return diffuse * mk3(texture_get_grad<COMPRESSED>(tex, tex_coord.x, tex_coord.y, lod.gradient_1, lod.gradient_2));
		return diffuse * mk3(texture_get_lod<COMPRESSED>(tex, tex_coord.x, tex_coord.y, lod.lod + tex.lod_bias));
	}
	return diffuse * mk3(texture_get<COMPRESSED>(tex, tex_coord.x, tex_coord.y));
}
template<bool COMPRESSED>
RT_DEV f3 sample_albedo(const RtParams & p, int bounce, f3 diffuse, int texture_id, f2 tex_coord, const TextureLOD & lod) {
	if (texture_id == RT_INVALID) return diffuse;
	const RtTexture tex = p.textures[texture_id];
	return sample_albedo<COMPRESSED>(p, bounce, diffuse, texture_id, tex, tex_coord, lod);
}

struct BSDFCommon {
	int pixel_index, bounce, sample_index;
	RandomPath rng;   // random_path(pixel_index, sample_index): what every random_sample of this hit shares
	f3 tangent, bitangent, normal, omega_i;
};

// COMPRESSED: whether a texture may hold BC1 blocks to be decoded per fetch (see texture_bilinear)
template<bool COMPRESSED>
struct BSDFDiffuseT : BSDFCommon {
	static constexpr bool HAS_ALBEDO = true;
	f3 diffuse; int texture_id; f3 albedo;
	RT_DEV void init(const RtParams & p, bool, int material_id) { float4 m = p.materials[2 * material_id]; diffuse = mk3(m.x, m.y, m.z); texture_id = __float_as_int(m.w); }
	RT_DEV void calc_albedo(const RtParams & p, f3 & throughput, f2 tex_coord, const TextureLOD & lod) {
		albedo = sample_albedo<COMPRESSED>(p, bounce, diffuse, texture_id, tex_coord, lod);
		if (bounce == 0) aov_set(p, RT_AOV_ALBEDO, pixel_index, mk4(albedo));
		if (!(p.config.enable_svgf && bounce == 0)) throughput *= albedo;
	}
	RT_DEV bool eval(const RtParams &, f3, float cos_theta_o, f3 & bsdf, float & pdf) const {
		if (cos_theta_o <= 0.0f) return false;
		bsdf = mk3(cos_theta_o * RT_ONE_OVER_PI);
		pdf  = cos_theta_o * RT_ONE_OVER_PI;
		return pdf_is_valid(pdf);
	}
	RT_DEV bool sample(const RtParams & p, f3 &, int &, f3 & direction_out, float & pdf) const {
		f2 r = random_sample(p, rng, DIM_BSDF_0, unsigned(bounce));
		f3 omega_o = sample_cosine_weighted_direction(r.x, r.y);
		direction_out = local_to_world(omega_o, tangent, bitangent, normal);
		pdf = omega_o.z * RT_ONE_OVER_PI;
		return pdf_is_valid(pdf);
	}
	RT_DEV bool has_texture() const { return texture_id != RT_INVALID; }
	RT_DEV bool allow_nee() const { return true; }
};

typedef BSDFDiffuseT<true> BSDFDiffuse;

template<bool COMPRESSED>
struct BSDFPlasticT : BSDFCommon {
	static constexpr bool HAS_ALBEDO = true;
	static constexpr float IOR = 1.5f;
	static constexpr float ETA = 1.0f / IOR;
	f3 diffuse; int texture_id; float linear_roughness; f3 albedo;
	float F_i, lambda_i, G1_i;
	RT_DEV void init(const RtParams & p, bool, int material_id) {
		float4 m = p.materials[2 * material_id]; diffuse = mk3(m.x, m.y, m.z); texture_id = __float_as_int(m.w);
		linear_roughness = p.materials[2 * material_id + 1].x;
		// what the light sample's evaluation (eval) and the bounce (sample) both need of the incoming direction, computed once per hit
		float ax = roughness_to_alpha(linear_roughness);
		F_i = fresnel_dielectric(omega_i.z, ETA);
		lambda_i = ggx_lambda(omega_i, ax, ax);
		G1_i = 1.0f / (1.0f + lambda_i);   // ggx_G1(omega_i, ax, ay)
	}
	// ggx_G2(omega_o, omega_i, omega_m, ax, ay) with the cached lambda of omega_i (the sum in the function's order)
	RT_DEV float G2_with(f3 omega_o, f3 omega_m, float ax, float ay) const {
		bool i_back = dot(omega_i, omega_m) * omega_i.z <= 0.0f;
		bool o_back = dot(omega_o, omega_m) * omega_o.z <= 0.0f;
		if (i_back || o_back) return 0.0f;
		return 1.0f / (1.0f + ggx_lambda(omega_o, ax, ay) + lambda_i);
	}
	RT_DEV void calc_albedo(const RtParams & p, f3 &, f2 tex_coord, const TextureLOD & lod) {
		albedo = sample_albedo<COMPRESSED>(p, bounce, diffuse, texture_id, tex_coord, lod);
		if (bounce == 0) aov_set(p, RT_AOV_ALBEDO, pixel_index, mk4(albedo));
	}
	RT_DEV f3 diffuse_lobe(float F_i, float F_o, float cos_o) const {
		float F_avg = average_fresnel(IOR);
		float internal_scattering_factor = 1.0f - (1.0f - F_avg) * square(ETA);
		return ETA * ETA * (1.0f - F_i) * (1.0f - F_o) * albedo * RT_ONE_OVER_PI / (1.0f - albedo * internal_scattering_factor) * cos_o;
	}
	RT_DEV bool eval(const RtParams &, f3 to_light, float cos_theta_o, f3 & bsdf, float & pdf) const {
		if (cos_theta_o <= 0.0f) return false;
		f3 omega_o = world_to_local(to_light, tangent, bitangent, normal);
		f3 omega_m = normalize(omega_i + omega_o);
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		float F  = fresnel_dielectric(dot(omega_i, omega_m), ETA);
		float D  = ggx_D(omega_m, ax, ay);
		float G1 = G1_i;
		float G2 = G2_with(omega_o, omega_m, ax, ay);
		f3 brdf_specular = mk3(F * G2 * D / (4.0f * omega_i.z));
		float F_o = fresnel_dielectric(omega_o.z, ETA);
		f3 brdf_diffuse = diffuse_lobe(F_i, F_o, omega_o.z);
		float pdf_specular = G1 * D / (4.0f * omega_i.z);
		float pdf_diffuse  = omega_o.z * RT_ONE_OVER_PI;
		pdf  = lerp_ref(pdf_diffuse, pdf_specular, F_i);
		bsdf = brdf_specular + brdf_diffuse;
		return pdf_is_valid(pdf);
	}
	RT_DEV bool sample(const RtParams & p, f3 & throughput, int &, f3 & direction_out, float & pdf) const {
		float rand_fresnel = random_sample(p, rng, DIM_BSDF_0, unsigned(bounce)).x;
		f2    rand_brdf    = random_sample(p, rng, DIM_BSDF_1, unsigned(bounce));
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		f3 omega_m, omega_o;
		if (rand_fresnel < F_i) {
			omega_m = sample_visible_normals_ggx(omega_i, ax, ay, rand_brdf.x, rand_brdf.y);
			omega_o = reflect_direction(omega_i, omega_m);
		} else {
			omega_o = sample_cosine_weighted_direction(rand_brdf.x, rand_brdf.y);
			omega_m = normalize(omega_i + omega_o);
		}
		if (omega_m.z < 0.0f) return false;
		float F  = fresnel_dielectric(dot(omega_i, omega_m), ETA);
		float D  = ggx_D(omega_m, ax, ay);
		float G1 = G1_i;
		float G2 = G2_with(omega_o, omega_m, ax, ay);
		f3 brdf_specular = mk3(F * G2 * D / (4.0f * omega_i.z));
		float F_o = fresnel_dielectric(omega_o.z, ETA);
		f3 brdf_diffuse = diffuse_lobe(F_i, F_o, omega_o.z);
		float pdf_specular = G1 * D / (4.0f * omega_i.z);
		float pdf_diffuse  = omega_o.z * RT_ONE_OVER_PI;
		pdf = lerp_ref(pdf_diffuse, pdf_specular, F_i);
		throughput *= (brdf_specular + brdf_diffuse) / pdf;
		direction_out = local_to_world(omega_o, tangent, bitangent, normal);
		return pdf_is_valid(pdf);
	}
	RT_DEV bool has_texture() const { return texture_id != RT_INVALID; }
	RT_DEV bool allow_nee() const { return true; }
};

typedef BSDFPlasticT<true> BSDFPlastic;

struct BSDFDielectric : BSDFCommon {
	static constexpr bool HAS_ALBEDO = false;
	int medium_id_material; float ior, linear_roughness, eta;
	RT_DEV void init(const RtParams & p, bool entering_material, int material_id) {
		float4 m = p.materials[2 * material_id];
		medium_id_material = __float_as_int(m.x); ior = m.y; linear_roughness = m.z;
		eta = entering_material ? 1.0f / ior : ior;
	}
	RT_DEV void calc_albedo(const RtParams &, f3 &, f2, const TextureLOD &) { }

	struct Lobes { float bsdf_single, bsdf_multi, pdf_single, pdf_multi; };
	RT_DEV Lobes lobes(const RtParams & p, bool reflected, bool entering_material, f3 omega_o, f3 omega_m, float F, float E_i, float ratio, float E_avg_enter, float E_avg_leave) const {
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		float D  = ggx_D(omega_m, ax, ay);
		float G1 = ggx_G1(omega_i, ax, ay);
		float G2 = ggx_G2(omega_o, omega_i, omega_m, ax, ay);
		float i_dot_m = abs_dot(omega_i, omega_m);
		float o_dot_m = abs_dot(omega_o, omega_m);
		Lobes l;
		if (reflected) {
			l.bsdf_single = F * G2 * D / (4.0f * omega_i.z);
			l.pdf_single  = F * G1 * D / (4.0f * omega_i.z);
			float E_o   = dielectric_directional_albedo(p, ior, linear_roughness, omega_o.z, entering_material);
			float E_avg = entering_material ? E_avg_enter : E_avg_leave;
			l.bsdf_multi = (1.0f - ratio) * fabsf(omega_o.z) * kulla_conty_multiscatter_lobe(E_i, E_o, E_avg);
			l.pdf_multi  = (1.0f - ratio) * fabsf(omega_o.z) * RT_ONE_OVER_PI;
		} else {
			l.bsdf_single = (1.0f - F) * G2 * D * i_dot_m * o_dot_m / (omega_i.z * square(eta * i_dot_m + o_dot_m) * square(eta));
			l.pdf_single  = (1.0f - F) * G1 * D * i_dot_m * o_dot_m / (omega_i.z * square(eta * i_dot_m + o_dot_m));
			float E_o   = dielectric_directional_albedo(p, ior, linear_roughness, omega_o.z, !entering_material);
			float E_avg = entering_material ? E_avg_leave : E_avg_enter; // inverted on purpose (BSDF.h:281)
			l.bsdf_multi = ratio * fabsf(omega_o.z) * kulla_conty_multiscatter_lobe(E_i, E_o, E_avg);
			l.pdf_multi  = ratio * fabsf(omega_o.z) * RT_ONE_OVER_PI;
		}
		return l;
	}
	RT_DEV void common(const RtParams & p, bool & entering_material, float & E_i, float & ratio, float & E_avg_enter, float & E_avg_leave) const {
		entering_material = eta < 1.0f;
		E_i = dielectric_directional_albedo(p, ior, linear_roughness, omega_i.z, entering_material);
		float F_avg = average_fresnel(ior);
		if (!entering_material) F_avg = 1.0f - (1.0f - F_avg) / square(ior);
		E_avg_enter = dielectric_albedo(p, ior, linear_roughness, true);
		E_avg_leave = dielectric_albedo(p, ior, linear_roughness, false);
		float x = kulla_conty_dielectric_reciprocity_factor(E_avg_enter, E_avg_leave);
		ratio = (entering_material ? x : (1.0f - x)) * (1.0f - F_avg);
	}
	RT_DEV bool eval(const RtParams & p, f3 to_light, float, f3 & bsdf, float & pdf) const {
		f3 omega_o = world_to_local(to_light, tangent, bitangent, normal);
		bool reflected = omega_o.z >= 0.0f;
		f3 omega_m = reflected ? normalize(omega_i + omega_o) : normalize(eta * omega_i + omega_o);
		omega_m *= sign_of(omega_m.z);
		float F = fresnel_dielectric(abs_dot(omega_i, omega_m), eta);
		bool entering_material; float E_i, ratio, E_avg_enter, E_avg_leave;
		common(p, entering_material, E_i, ratio, E_avg_enter, E_avg_leave);
		Lobes l = lobes(p, reflected, entering_material, omega_o, omega_m, F, E_i, ratio, E_avg_enter, E_avg_leave);
		bsdf = mk3(l.bsdf_single + l.bsdf_multi);
		pdf = lerp_ref(l.pdf_multi, l.pdf_single, E_i);
		return pdf_is_valid(pdf);
	}
	RT_DEV bool sample(const RtParams & p, f3 & throughput, int & medium_id, f3 & direction_out, float & pdf) const {
		f2 r0 = random_sample(p, rng, DIM_BSDF_0, unsigned(bounce));
		f2 r1 = random_sample(p, rng, DIM_BSDF_1, unsigned(bounce));
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		bool entering_material; float E_i, ratio, E_avg_enter, E_avg_leave;
		common(p, entering_material, E_i, ratio, E_avg_enter, E_avg_leave);

		float F; bool reflected; f3 omega_m, omega_o;
		if (r0.x < E_i) {
			omega_m = sample_visible_normals_ggx(omega_i, ax, ay, r1.x, r1.y);
			F = fresnel_dielectric(abs_dot(omega_i, omega_m), eta);
			reflected = r0.y < F;
			omega_o = reflected ? reflect_direction(omega_i, omega_m) : refract_direction(omega_i, omega_m, eta);
		} else {
			omega_o = sample_cosine_weighted_direction(r1.x, r1.y);
			reflected = r0.y > ratio;
			if (reflected) {
				omega_m = normalize(omega_i + omega_o);
			} else {
				omega_o = -omega_o;
				omega_m = normalize(eta * omega_i + omega_o);
			}
			omega_m *= sign_of(omega_m.z);
			F = fresnel_dielectric(abs_dot(omega_i, omega_m), eta);
		}
		if (reflected ^ (omega_o.z >= 0.0f)) return false;

		Lobes l = lobes(p, reflected, entering_material, omega_o, omega_m, F, E_i, ratio, E_avg_enter, E_avg_leave);
		if (!reflected) medium_id = entering_material ? medium_id_material : RT_INVALID;
		pdf = lerp_ref(l.pdf_multi, l.pdf_single, E_i);
		throughput *= (l.bsdf_single + l.bsdf_multi) / pdf;
		direction_out = local_to_world(omega_o, tangent, bitangent, normal);
		return pdf_is_valid(pdf);
	}
	RT_DEV bool has_texture() const { return false; }
	RT_DEV bool allow_nee() const { return linear_roughness >= RT_ROUGHNESS_CUTOFF; }
};

struct BSDFConductor : BSDFCommon {
	static constexpr bool HAS_ALBEDO = false;
	f3 eta3, k3; float linear_roughness;
	RT_DEV void init(const RtParams & p, bool, int material_id) {
		float4 a = p.materials[2 * material_id], b = p.materials[2 * material_id + 1];
		eta3 = mk3(a.x, a.y, a.z); linear_roughness = a.w; k3 = mk3(b.x, b.y, b.z);
	}
	RT_DEV void calc_albedo(const RtParams &, f3 &, f2, const TextureLOD &) { }
	RT_DEV void lobes(const RtParams & p, f3 omega_o, f3 omega_m, float o_dot_m, float E_i, f3 & brdf, float & pdf) const {
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		f3    F  = fresnel_conductor(o_dot_m, eta3, k3);
		float D  = ggx_D(omega_m, ax, ay);
		float G1 = ggx_G1(omega_i, ax, ay);
		float G2 = ggx_G2(omega_o, omega_i, omega_m, ax, ay);
		f3    brdf_single = F * G2 * D / (4.0f * omega_i.z);
		float pdf_single  =     G1 * D / (4.0f * omega_i.z);
		float E_o   = conductor_directional_albedo(p, linear_roughness, omega_o.z);
		float E_avg = conductor_albedo(p, linear_roughness);
		f3 F_avg = average_fresnel(eta3, k3);
		f3 F_ms  = fresnel_multiscatter(F_avg, E_avg);
		f3    brdf_multi = F_ms * kulla_conty_multiscatter_lobe(E_i, E_o, E_avg) * omega_o.z;
		float pdf_multi  = omega_o.z * RT_ONE_OVER_PI;
		brdf = brdf_single + brdf_multi;
		pdf = lerp_ref(pdf_multi, pdf_single, E_i);
	}
	RT_DEV bool eval(const RtParams & p, f3 to_light, float cos_theta_o, f3 & bsdf, float & pdf) const {
		if (cos_theta_o <= 0.0f) return false;
		f3 omega_o = world_to_local(to_light, tangent, bitangent, normal);
		f3 omega_m = normalize(omega_o + omega_i);
		float o_dot_m = dot(omega_o, omega_m);
		if (o_dot_m <= 0.0f) return false;
		float E_i = conductor_directional_albedo(p, linear_roughness, omega_i.z);
		lobes(p, omega_o, omega_m, o_dot_m, E_i, bsdf, pdf);
		return pdf_is_valid(pdf);
	}
	RT_DEV bool sample(const RtParams & p, f3 & throughput, int &, f3 & direction_out, float & pdf) const {
		f2 r0 = random_sample(p, rng, DIM_BSDF_0, unsigned(bounce));
		f2 r1 = random_sample(p, rng, DIM_BSDF_1, unsigned(bounce));
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		float E_i = conductor_directional_albedo(p, linear_roughness, omega_i.z);
		f3 omega_m, omega_o;
		if (r0.x < E_i) {
			omega_m = sample_visible_normals_ggx(omega_i, ax, ay, r1.x, r1.y);
			omega_o = reflect_direction(omega_i, omega_m);
		} else {
			omega_o = sample_cosine_weighted_direction(r1.x, r1.y);
			omega_m = normalize(omega_i + omega_o);
		}
		float o_dot_m = dot(omega_o, omega_m);
		if (o_dot_m <= 0.0f || omega_o.z < 0.0f) return false;
		f3 brdf;
		lobes(p, omega_o, omega_m, o_dot_m, E_i, brdf, pdf);
		throughput *= brdf / pdf;
		direction_out = local_to_world(omega_o, tangent, bitangent, normal);
		return pdf_is_valid(pdf);
	}
	RT_DEV bool has_texture() const { return false; }
	RT_DEV bool allow_nee() const { return linear_roughness >= RT_ROUGHNESS_CUTOFF; }
};

// ---- next event estimation (Pathtracer.cu:465-555) ----------------------------------------------------------

// The two cumulative tables of light sampling, copied into LDS by the workgroup when they fit (Sponza: 2 meshes, 480 triangles):
// the two binary searches are chains of ~2 + ~9 dependent loads per hit, each a round trip to L2 from a kernel that runs 3-4 waves
// per SIMD; from LDS a step costs a twentieth of that. Same floats, same comparisons.
#ifndef RT_LIGHT_TABLES_LDS
#define RT_LIGHT_TABLES_LDS 1
#endif
#define RT_LIGHT_MESHES_IN_LDS    64
#define RT_LIGHT_TRIANGLES_IN_LDS 2048
struct LightTablesLDS { float mesh_cdf[RT_LIGHT_MESHES_IN_LDS]; float triangle_cdf[RT_LIGHT_TRIANGLES_IN_LDS]; };
RT_DEV bool light_tables_fit_lds(const RtParams & p) { return p.light_mesh_count <= RT_LIGHT_MESHES_IN_LDS && p.light_triangle_count <= RT_LIGHT_TRIANGLES_IN_LDS; }
RT_DEV void light_tables_to_lds(const RtParams & p, LightTablesLDS & lds) {   // whole workgroup; the caller's next barrier publishes the copy
	if (!light_tables_fit_lds(p)) return;
	for (int i = threadIdx.x; i < p.light_mesh_count;     i += blockDim.x) lds.mesh_cdf[i]     = p.light_mesh_cumulative_probability[i];
	for (int i = threadIdx.x; i < p.light_triangle_count; i += blockDim.x) lds.triangle_cdf[i] = p.light_triangle_cumulative_probability[i];
}
typedef const __attribute__((address_space(3))) float * LdsFloatTable;   // (typed: ds_read_b32, not a FLAT load)

RT_DEV int sample_light(const RtParams & p, const LightTablesLDS * lds, float u1, float u2, int & transform_id) { // Sampling.h:180-190
	const bool from_lds = lds != nullptr && light_tables_fit_lds(p);   // uniform
	int light_mesh_id = from_lds ? binary_search((LdsFloatTable)lds->mesh_cdf, 0, p.light_mesh_count - 1, u1)
	                             : binary_search(p.light_mesh_cumulative_probability, 0, p.light_mesh_count - 1, u1);
	transform_id = p.light_mesh_transform_indices[light_mesh_id];
	if (p.mesh_position) transform_id = p.mesh_position[transform_id]; // device-built TLAS: the host only knows scene indices
	int2 span = p.light_mesh_triangle_span[light_mesh_id];
	int light_triangle_id = from_lds ? binary_search((LdsFloatTable)lds->triangle_cdf, span.x, span.y, u2)
	                                 : binary_search(p.light_triangle_cumulative_probability, span.x, span.y, u2);
	return p.light_triangle_indices[light_triangle_id];
}

struct ShadowRay { f3 origin, direction; float max_distance; f3 illumination; };

// The light sample of a hit: a point on an emitter chosen by the two cumulative tables, in world space, with the emitter's normal and emission.
// It depends on the path's random numbers alone, not on the surface. (Measured, profiles/r04_shade_stage.txt 5.: taking it BEFORE the surface
// set-up, so that its loads run beside the hit's own chain, changes nothing -- the material kernels are not waiting on that chain.)
struct LightSample { f3 point, geometric_normal, emission; };
RT_DEV LightSample nee_pick_light(const RtParams & p, const LightTablesLDS * light_lds, const RandomPath & rng, int bounce) {
	f2 rand_light    = random_sample(p, rng, DIM_NEE_LIGHT,    unsigned(bounce));
	f2 rand_triangle = random_sample(p, rng, DIM_NEE_TRIANGLE, unsigned(bounce));

	int light_mesh_id;
	int light_triangle_id = sample_light(p, light_lds, rand_light.x, rand_light.y, light_mesh_id);
	f2 light_uv = sample_triangle(rand_triangle.x, rand_triangle.y);

	f3 p0, e1, e2;
	triangle_get_positions(p, light_triangle_id, p0, e1, e2);
	LightSample light;
	light.point = barycentric(light_uv.x, light_uv.y, p0, e1, e2);
	light.geometric_normal = cross(e1, e2);

	const float4 * light_world = p.mesh_transforms + size_t(light_mesh_id) * 3;
	light.point = m_position(light_world, light.point);
	light.geometric_normal = normalize(m_direction(light_world, light.geometric_normal));

	int light_material_id = p.mesh_material_ids[light_mesh_id];
	light.emission = mk3(p.materials[2 * light_material_id]);
	return light;
}

// Connects a hit with its light sample. Returns true and fills `shadow` when the sample has to be traced (the caller appends it).
template<typename BSDF>
RT_DEV bool nee_connect(const RtParams & p, const LightSample & light, const BSDF & bsdf, f3 hit_point, f3 normal, f3 geometric_normal, f3 throughput, ShadowRay & shadow) {
	f3 light_point = light.point, light_geometric_normal = light.geometric_normal;
	hit_point   = ray_origin_epsilon_offset(hit_point,   light_point - hit_point, geometric_normal);
	light_point = ray_origin_epsilon_offset(light_point, hit_point - light_point, light_geometric_normal);

	f3 to_light = light_point - hit_point;
	float distance_to_light = length(to_light);
	to_light /= distance_to_light;

	float cos_theta_light = abs_dot(to_light, light_geometric_normal);
	float cos_theta_hit = dot(to_light, normal);

	f3 emission = light.emission;

	f3 bsdf_value; float bsdf_pdf;
	if (!bsdf.eval(p, to_light, cos_theta_hit, bsdf_value, bsdf_pdf)) return false;

	float light_power = luminance(emission.x, emission.y, emission.z);
	float light_pdf   = light_power * square(distance_to_light) / (cos_theta_light * p.lights_total_weight);
	if (!pdf_is_valid(light_pdf)) return false;

	float mis_weight = p.config.enable_multiple_importance_sampling ? power_heuristic(light_pdf, bsdf_pdf) : 1.0f;
	shadow.illumination = throughput * bsdf_value * emission * mis_weight / light_pdf;
	shadow.origin = hit_point;
	shadow.direction = to_light;
	shadow.max_distance = distance_to_light;
	return true;
}
template<typename BSDF>
RT_DEV bool next_event_estimation(const RtParams & p, const LightTablesLDS * light_lds, int pixel_index, int bounce, int sample_index, const BSDF & bsdf, f3 hit_point, f3 normal, f3 geometric_normal, f3 throughput, ShadowRay & shadow) {
	return nee_connect(p, nee_pick_light(p, light_lds, bsdf.rng, bounce), bsdf, hit_point, normal, geometric_normal, throughput, shadow);
}

// ---- shade_material<BSDF> (Pathtracer.cu:557-757) -------------------------------------------------------------

RT_DEV float triangle_get_lod(float double_area_world_inv, f2 te1, f2 te2) {
	float area_texel = fabsf(te1.x * te2.y - te2.x * te1.y);
	return sqrtf(area_texel * double_area_world_inv);
}
RT_DEV float triangle_get_curvature(f3 pe1, f3 pe2, f3 ne1, f3 ne2) {
	f3 ne0 = ne1 - ne2;
	f3 pe0 = pe1 - pe2;
	float k_01 = dot(ne1, pe1) / dot(pe1, pe1);
	float k_02 = dot(ne2, pe2) / dot(pe2, pe2);
	float k_12 = dot(ne0, pe0) / dot(pe0, pe0);
	return (k_01 + k_02 + k_12) * (1.0f / 3.0f);
}
RT_DEV void ray_cone_get_ellipse_axes(f3 ray_direction, f3 geometric_normal, float cone_width, f3 & axis_1, f3 & axis_2) {
	f3 h_1 = ray_direction - dot(geometric_normal, ray_direction) * geometric_normal;
	f3 h_2 = cross(geometric_normal, h_1);
	axis_1 = cone_width / fmaxf(0.0001f, length(h_1 - dot(ray_direction, h_1) * ray_direction)) * h_1;
	axis_2 = cone_width / fmaxf(0.0001f, length(h_2 - dot(ray_direction, h_2) * ray_direction)) * h_2;
}
RT_DEV f2 ray_cone_ellipse_axis_to_gradient(const TriangleFull & tri, float double_area_inv, f3 geometric_normal, f3 hit_point, f2 hit_tex_coord, f3 ellipse_axis) {
	f3 e_p = hit_point + ellipse_axis - tri.position_0;
	float u = dot(geometric_normal, cross(e_p, tri.position_edge_2)) * double_area_inv;
	float v = dot(geometric_normal, cross(tri.position_edge_1, e_p)) * double_area_inv;
	return barycentric(u, v, tri.tex_coord_0, tri.tex_coord_edge_1, tri.tex_coord_edge_2) - hit_tex_coord;
}
RT_DEV float ray_cone_get_lod(f3 ray_direction, f3 geometric_normal, float cone_width) { return fabsf(cone_width / dot(ray_direction, geometric_normal)); }

// MERGED: the queue holds the surface hits of every submission in flight (see sort_rays); bounce and sample come
// from the slot table, the launch arguments are ignored.
template<typename BSDF, int SLOT, bool MERGED>
RT_DEV void shade_material(const RtParams & p, int launch_bounce, int launch_sample_index) {
	const RtMaterialBuffer & q = p.material[SLOT];
	const int iq = MERGED ? (p.stream_iteration & 1) : (launch_bounce & 1);
	const RtTraceBuffer & out = p.trace[iq ^ 1];
	const int buffer_size = MERGED ? p.stream->material_count[SLOT]
	                               : (SLOT == 0 ? p.sizes->diffuse[launch_bounce] : SLOT == 1 ? p.sizes->plastic[launch_bounce] : SLOT == 2 ? p.sizes->dielectric[launch_bounce] : p.sizes->conductor[launch_bounce]);

	__shared__ BlockBucketLDS<RT_SHADE_BLOCK / RT_WAVE_SIZE> append_lds;
#if RT_SHADE_ONE_APPEND
	__shared__ BlockBucket2LDS<RT_SHADE_BLOCK / RT_WAVE_SIZE> append2_lds;
#endif
	__shared__ StreamStatsLDS stats_lds;
	int * const shadow_counter = MERGED ? &p.stream->shadow_count[iq]    : &p.sizes->shadow[launch_bounce];
	int * const trace_counter  = MERGED ? &p.stream->trace_count[iq ^ 1] : &p.sizes->trace[launch_bounce + 1];
	const bool nee_enabled = p.config.enable_next_event_estimation && p.lights_total_weight > 0.0f; // uniform
#if RT_LIGHT_TABLES_LDS
	__shared__ LightTablesLDS light_lds;
	if (nee_enabled && blockIdx.x * blockDim.x < unsigned(buffer_size)) light_tables_to_lds(p, light_lds);
	const LightTablesLDS * const light_tables = &light_lds;
#else
	const LightTablesLDS * const light_tables = nullptr;
#endif
	if (MERGED) stream_stats_clear(stats_lds); else __syncthreads();   // (either way a barrier: the tables are in place)

	// every thread of the workgroup makes the same number of rounds (block_aggregated_append has barriers)
	for (int first = blockIdx.x * blockDim.x; first < buffer_size; first += gridDim.x * blockDim.x) {
		const int index = first + int(threadIdx.x);
		// what survives from the surface set-up to the two queue appends
		BSDF bsdf;
		int pixel_index = 0, medium_id = RT_INVALID;
		f3 throughput = mk3(1.0f), hit_point = mk3(0.0f), geometric_normal = mk3(0.0f);
		float cone_angle = 0.0f, cone_width = 0.0f;
		ShadowRay shadow; bool has_shadow_ray = false;
		int bounce = launch_bounce, sample_index = launch_sample_index, submission = 0;

		auto set_up_surface = [&]() -> bool { // false: the path ends here
		if (index >= buffer_size) return false;
		unsigned pixel_index_and_flags = q.pixel_index_and_flags[index];
		pixel_index = int(pixel_index_and_flags & ~RT_FLAGS_ALL);
		if (MERGED) {
			RtPathInfo info = rt_stream_path_info(p, unsigned(pixel_index));
			bounce = info.bounce; sample_index = int(info.sample_index_for_rng); submission = info.submission;
		}
		f3 ray_direction = load3(q.direction, index);
		HitInfo hit = unpack_hit(q.hits[index]);

		bool inside_medium = pixel_index_and_flags & RT_FLAG_INSIDE_MEDIUM;
		medium_id = inside_medium ? q.medium[index] : RT_INVALID;

		if (MERGED) { f3 carried = load3(q.throughput, index); throughput = bounce == 0 ? mk3(1.0f) : carried; }   // (beside the slot-table lookup, see sort_rays)
		else throughput = bounce == 0 ? mk3(1.0f) : load3(q.throughput, index);

		TriangleFull tri = triangle_get_full(p, hit.triangle_id);
		hit_point = barycentric(hit.u, hit.v, tri.position_0, tri.position_edge_1, tri.position_edge_2);
		f3 normal    = barycentric(hit.u, hit.v, tri.normal_0,   tri.normal_edge_1,   tri.normal_edge_2);
		f2 tex_coord = barycentric(hit.u, hit.v, tri.tex_coord_0, tri.tex_coord_edge_1, tri.tex_coord_edge_2);
		f3 hit_point_local = hit_point;

		const float4 * world = p.mesh_transforms + size_t(hit.mesh_id) * 3;
		hit_point = m_position(world, hit_point);
		normal = normalize(m_direction(world, normal));

		float4 world_row_0 = world[0];
		float mesh_scale_inv = 1.0f / length(mk3(world_row_0.x, world_row_0.y, world_row_0.z));

		float curvature = 0.0f;
		if (p.config.enable_mipmapping) {
			float carried_angle = 0.0f, carried_width = 0.0f;
			if (MERGED || bounce > 0) { carried_angle = q.cone_angle[index]; carried_width = q.cone_width[index]; }
			if (bounce == 0) { cone_angle = p.camera.pixel_spread_angle; cone_width = cone_angle * hit.t; }
			else             { cone_angle = carried_angle; cone_width = carried_width + cone_angle * hit.t; }
			curvature = triangle_get_curvature(tri.position_edge_1, tri.position_edge_2, tri.normal_edge_1, tri.normal_edge_2) * mesh_scale_inv;
		}

		tri.position_edge_1 = m_direction(world, tri.position_edge_1);
		tri.position_edge_2 = m_direction(world, tri.position_edge_2);

		geometric_normal = cross(tri.position_edge_1, tri.position_edge_2);
		float triangle_double_area_inv = 1.0f / length(geometric_normal);
		geometric_normal *= triangle_double_area_inv;

		bool entering_material = dot(ray_direction, geometric_normal) < 0.0f;
		if (!entering_material) { normal = -normal; curvature = -curvature; }

		f3 tangent, bitangent;
		orthonormal_basis(normal, tangent, bitangent);
		f3 omega_i = world_to_local(-ray_direction, tangent, bitangent, normal);
		if (omega_i.z <= 0.0f) return false;

		int material_id = p.mesh_material_ids[hit.mesh_id];

		bsdf.pixel_index = pixel_index; bsdf.bounce = bounce; bsdf.sample_index = sample_index; bsdf.rng = random_path(p, unsigned(pixel_index), unsigned(sample_index));
		bsdf.tangent = tangent; bsdf.bitangent = bitangent; bsdf.normal = normal; bsdf.omega_i = omega_i;
		bsdf.init(p, entering_material, material_id);

		if (BSDF::HAS_ALBEDO) {
			TextureLOD lod = { mk2(0.0f, 0.0f), mk2(0.0f, 0.0f), 0.0f };
			if (p.config.enable_mipmapping && bsdf.has_texture()) {
				if (bounce == 0) {
					f3 axis_1, axis_2;
					ray_cone_get_ellipse_axes(ray_direction, geometric_normal, cone_width, axis_1, axis_2);
					lod.gradient_1 = ray_cone_ellipse_axis_to_gradient(tri, triangle_double_area_inv, geometric_normal, hit_point, tex_coord, axis_1);
					lod.gradient_2 = ray_cone_ellipse_axis_to_gradient(tri, triangle_double_area_inv, geometric_normal, hit_point, tex_coord, axis_2);
				} else {
					float lod_triangle = triangle_get_lod(triangle_double_area_inv, tri.tex_coord_edge_1, tri.tex_coord_edge_2);
					float lod_ray_cone = ray_cone_get_lod(ray_direction, geometric_normal, cone_width);
					lod.lod = log2f(lod_triangle * lod_ray_cone);
				}
			}
			bsdf.calc_albedo(p, throughput, tex_coord, lod);
		} else if (bounce == 0) {
			aov_set(p, RT_AOV_ALBEDO, pixel_index, mk4(1.0f));
		}

		if (bounce == 0) {
			aov_set(p, RT_AOV_NORMAL,   pixel_index, mk4(normal));
			aov_set(p, RT_AOV_POSITION, pixel_index, mk4(hit_point));
		}

		if (p.config.enable_mipmapping) cone_angle -= 2.0f * curvature * fabsf(cone_width) / dot(normal, ray_direction);

		if (bounce == 0 && p.config.enable_svgf) {
			f3 hit_point_prev = m_position(p.mesh_transforms_prev + size_t(hit.mesh_id) * 3, hit_point_local);
			int x = pixel_index % p.screen_pitch, y = pixel_index / p.screen_pitch;
			svgf_set_gbuffers(p, x, y, hit, hit_point, normal, hit_point_prev);
		}

		if (nee_enabled && bsdf.allow_nee()) {
			has_shadow_ray = next_event_estimation(p, light_tables, pixel_index, bounce, sample_index, bsdf, hit_point, normal, geometric_normal, throughput, shadow);
		}
		return true;
		};

		bool alive = set_up_surface();

		auto store_shadow_ray = [&](int shadow_ray_index) {
			store3(p.shadow.origin,    shadow_ray_index, shadow.origin);
			store3(p.shadow.direction, shadow_ray_index, shadow.direction);
			p.shadow.max_distance[shadow_ray_index] = shadow.max_distance;
			// merged wavefront: the trace launch that consumes the ray cannot know its bounce from a launch argument
			unsigned pixel_word = unsigned(pixel_index) | (MERGED && bounce == 0 ? RT_SHADOW_FLAG_BOUNCE_0 : 0u);
			p.shadow.illumination_and_pixel_index[shadow_ray_index] = make_float4(shadow.illumination.x, shadow.illumination.y, shadow.illumination.z, __uint_as_float(pixel_word));
		};
		if (nee_enabled && MERGED) stream_stats_add(stats_lds, has_shadow_ray, submission, RT_STAT_SHADOW);
#if !RT_SHADE_ONE_APPEND
		if (nee_enabled) {
			// (both ray queues: the workgroup's rays ordered by direction octant, see block_bucketed_append)
			int shadow_ray_index = block_bucketed_append(has_shadow_ray, RT_RAY_BUCKET(shadow.direction), shadow_counter, append_lds);
			if (has_shadow_ray) store_shadow_ray(shadow_ray_index);
		}
#endif

		f3 direction_out = mk3(0.0f); float pdf = 0.0f;
		bool continues = alive && bsdf.sample(p, throughput, medium_id, direction_out, pdf);

		int index_out;
#if RT_SHADE_ONE_APPEND
		// the round's two appends (shadow ray, continuation ray) in one: two workgroup barriers and one atomic latency instead of four and two
		if (nee_enabled) {
			int shadow_ray_index;
			block_bucketed_append2(has_shadow_ray, RT_RAY_BUCKET(shadow.direction), shadow_counter, continues, RT_RAY_BUCKET(direction_out), trace_counter, append2_lds, shadow_ray_index, index_out);
			if (has_shadow_ray) store_shadow_ray(shadow_ray_index);
		} else
#endif
		index_out = block_bucketed_append(continues, RT_RAY_BUCKET(direction_out), trace_counter, append_lds);
		if (!continues) continue;

		f3 origin_out = ray_origin_epsilon_offset(hit_point, direction_out, geometric_normal);
		store3(out.origin,    index_out, origin_out);
		store3(out.direction, index_out, direction_out);
		if (medium_id != RT_INVALID) out.medium[index_out] = medium_id;
		if (p.config.enable_mipmapping) { out.cone_angle[index_out] = cone_angle; out.cone_width[index_out] = cone_width; }

		bool allow_nee = bsdf.allow_nee();
		unsigned flags = 0;
		if (allow_nee)               flags |= RT_FLAG_ALLOW_NEE;
		if (medium_id != RT_INVALID) flags |= RT_FLAG_INSIDE_MEDIUM;
		out.pixel_index_and_flags[index_out] = unsigned(pixel_index) | flags;
		store3(out.throughput, index_out, throughput);
		if (allow_nee) out.last_pdf[index_out] = pdf;
	}
	if (MERGED) stream_stats_flush(p, stats_lds);
}

__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_diffuse(RtParams p, int bounce, int sample_index)    { shade_material<BSDFDiffuse,    0, false>(p, bounce, sample_index); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_plastic(RtParams p, int bounce, int sample_index)    { shade_material<BSDFPlastic,    1, false>(p, bounce, sample_index); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_dielectric(RtParams p, int bounce, int sample_index) { shade_material<BSDFDielectric, 2, false>(p, bounce, sample_index); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_conductor(RtParams p, int bounce, int sample_index)  { shade_material<BSDFConductor,  3, false>(p, bounce, sample_index); }
// ..._texels: no texture on the device holds compressed blocks (RtParams::textures_compressed == 0)
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_diffuse_texels(RtParams p, int bounce, int sample_index) { shade_material<BSDFDiffuseT<false>, 0, false>(p, bounce, sample_index); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_plastic_texels(RtParams p, int bounce, int sample_index) { shade_material<BSDFPlasticT<false>, 1, false>(p, bounce, sample_index); }
#ifndef RT_SHADE_WAVES_DIFFUSE
#define RT_SHADE_WAVES_DIFFUSE RT_SHADE_WAVES
#endif
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES_DIFFUSE) kernel_material_diffuse_stream_texels(RtParams p) { shade_material<BSDFDiffuseT<false>, 0, true>(p, 0, 0); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_plastic_stream_texels(RtParams p) { shade_material<BSDFPlasticT<false>, 1, true>(p, 0, 0); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_diffuse_stream(RtParams p)    { shade_material<BSDFDiffuse,    0, true>(p, 0, 0); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_plastic_stream(RtParams p)    { shade_material<BSDFPlastic,    1, true>(p, 0, 0); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_dielectric_stream(RtParams p) { shade_material<BSDFDielectric, 2, true>(p, 0, 0); }
__global__ void __launch_bounds__(RT_SHADE_BLOCK, RT_SHADE_WAVES) kernel_material_conductor_stream(RtParams p)  { shade_material<BSDFConductor,  3, true>(p, 0, 0); }

// ---- ambient occlusion (CUDA/AO.cu:103-159) -------------------------------------------------------
// One cosine-weighted occlusion ray of length ao_radius per primary hit; the AO shadow kernel sets
// RADIANCE to 1 where it escapes. Same block-aggregated queue append as the material kernels.
__global__ void __launch_bounds__(RT_SHADE_BLOCK) kernel_ambient_occlusion(RtParams p, int sample_index, float ao_radius) {
	const int ray_count = p.sizes->trace[0];
	const RtTraceBuffer & in = p.trace[0];
	__shared__ BlockAppendLDS<1, RT_SHADE_BLOCK / RT_WAVE_SIZE> append_lds;
	int * const shadow_counter[1] = { &p.sizes->shadow[0] };

	for (int first = blockIdx.x * blockDim.x; first < ray_count; first += gridDim.x * blockDim.x) {
		const int index = first + int(threadIdx.x);
		ShadowRay shadow; int pixel_index = 0;
		bool emit = false;
		if (index < ray_count) {
			HitInfo hit = unpack_hit(in.hits[index]);
			pixel_index = int(in.pixel_index_and_flags[index] & ~RT_FLAGS_ALL);
			if (hit.triangle_id != RT_INVALID) {
				if (p.pixel_query_pixel == pixel_index) { p.pixel_query_out[0] = hit.mesh_id; p.pixel_query_out[1] = hit.triangle_id; } // AO.cu:115-118
				f3 ray_direction = load3(in.direction, index);
				TriangleFull tri = triangle_get_full(p, hit.triangle_id);
				f3 geometric_normal = normalize(cross(tri.position_edge_1, tri.position_edge_2)); // object space, as in AO.cu:123
				f3 hit_point  = barycentric(hit.u, hit.v, tri.position_0, tri.position_edge_1, tri.position_edge_2);
				f3 hit_normal = barycentric(hit.u, hit.v, tri.normal_0,   tri.normal_edge_1,   tri.normal_edge_2);

				const float4 * world = p.mesh_transforms + size_t(hit.mesh_id) * 3;
				hit_point  = m_position(world, hit_point);
				hit_normal = normalize(m_direction(world, hit_normal));
				if (dot(ray_direction, hit_normal) > 0.0f) hit_normal = -hit_normal;

				aov_set(p, RT_AOV_NORMAL,   pixel_index, mk4(hit_normal));
				aov_set(p, RT_AOV_POSITION, pixel_index, mk4(hit_point));

				f3 tangent, bitangent;
				orthonormal_basis(hit_normal, tangent, bitangent);
				f2 rand_brdf = random_sample(p, DIM_BSDF_0, unsigned(pixel_index), 0, unsigned(sample_index));
				f3 omega_o = sample_cosine_weighted_direction(rand_brdf.x, rand_brdf.y);
				f3 direction_out = local_to_world(omega_o, tangent, bitangent, hit_normal);
				float pdf = omega_o.z * RT_ONE_OVER_PI;
				if (pdf_is_valid(pdf)) {
					emit = true;
					shadow.origin = ray_origin_epsilon_offset(hit_point, direction_out, geometric_normal);
					shadow.direction = direction_out;
					shadow.max_distance = ao_radius;
				}
			}
		}
		int shadow_ray_index = block_aggregated_append(emit ? 0 : -1, shadow_counter, append_lds);
		if (emit) {
			store3(p.shadow.origin,    shadow_ray_index, shadow.origin);
			store3(p.shadow.direction, shadow_ray_index, shadow.direction);
			p.shadow.max_distance[shadow_ray_index] = shadow.max_distance;
			p.shadow.illumination_and_pixel_index[shadow_ray_index] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(pixel_index));
		}
	}
}

// ---- launchers ---------------------------------------------------------------------------------------------------

static int streaming_grid(int work_items) {
	// cap at ~8 workgroups per CU and grid-stride the rest (cdna_hip_programming.md G11)
	int blocks = (work_items + RT_SHADE_BLOCK - 1) / RT_SHADE_BLOCK;
	if (blocks < 1) blocks = 1;
	if (blocks > 2048) blocks = 2048;
	return blocks;
}

void rt_launch_generate(const RtParams & p, int sample_index, int pixel_offset, int pixel_count, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_generate, dim3(streaming_grid(pixel_count)), dim3(RT_SHADE_BLOCK), 0, stream, p, sample_index, pixel_offset, pixel_count);
}
void rt_launch_sort(const RtParams & p, int bounce, int sample_index, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_sort, dim3(2048 * RT_SHADE_BLOCK / RT_SORT_BLOCK), dim3(RT_SORT_BLOCK), 0, stream, p, bounce, sample_index);
}
void rt_launch_material(const RtParams & p, int material_slot, int bounce, int sample_index, hipStream_t stream) {
	dim3 grid(2048), block(RT_SHADE_BLOCK);
	switch (material_slot) {
		case 0: if (p.textures_compressed) hipLaunchKernelGGL(kernel_material_diffuse, grid, block, 0, stream, p, bounce, sample_index); else hipLaunchKernelGGL(kernel_material_diffuse_texels, grid, block, 0, stream, p, bounce, sample_index); break;
		case 1: if (p.textures_compressed) hipLaunchKernelGGL(kernel_material_plastic, grid, block, 0, stream, p, bounce, sample_index); else hipLaunchKernelGGL(kernel_material_plastic_texels, grid, block, 0, stream, p, bounce, sample_index); break;
		case 2: hipLaunchKernelGGL(kernel_material_dielectric, grid, block, 0, stream, p, bounce, sample_index); break;
		case 3: hipLaunchKernelGGL(kernel_material_conductor,  grid, block, 0, stream, p, bounce, sample_index); break;
	}
}
void rt_launch_generate_stream(const RtParams & p, int sample_index, int pixel_offset, int pixel_count, int slot_base, int queue_offset, int block_width, int band_rows, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_generate_stream, dim3(streaming_grid(pixel_count * p.batch_samples)), dim3(RT_SHADE_BLOCK), 0, stream, p, sample_index, pixel_offset, pixel_count, slot_base, queue_offset, block_width, band_rows);
}
void rt_launch_stream_advance(RtStreamControl * control, int iteration, int generated, int * progress, int reset_ring_first, int reset_ring_count, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_stream_advance, dim3(1), dim3(256), 0, stream, control, iteration, generated, (volatile int *)progress, reset_ring_first, reset_ring_count);
}
#ifndef RT_STREAM_SHADE_GRID
#define RT_STREAM_SHADE_GRID 8192   // workgroups of the grid-stride shade launches of the merged wavefront: 2 048 (two rounds of
                                    // resident workgroups) left 2.7 of 4 waves per SIMD resident on average; 8 192: shade stage -9 %, step -2.4 %
#endif
#ifndef RT_STREAM_SORT_GRID
#define RT_STREAM_SORT_GRID 2048    // ... and of its sort launch (in units of RT_SHADE_BLOCK threads)
#endif
void rt_launch_sort_stream(const RtParams & p, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_sort_stream, dim3(RT_STREAM_SORT_GRID * RT_SHADE_BLOCK / RT_SORT_BLOCK), dim3(RT_SORT_BLOCK), 0, stream, p);
}
void rt_launch_material_stream(const RtParams & p, int material_slot, hipStream_t stream) {
	dim3 grid(RT_STREAM_SHADE_GRID), block(RT_SHADE_BLOCK);
	switch (material_slot) {
		case 0: if (p.textures_compressed) hipLaunchKernelGGL(kernel_material_diffuse_stream, grid, block, 0, stream, p); else hipLaunchKernelGGL(kernel_material_diffuse_stream_texels, grid, block, 0, stream, p); break;
		case 1: if (p.textures_compressed) hipLaunchKernelGGL(kernel_material_plastic_stream, grid, block, 0, stream, p); else hipLaunchKernelGGL(kernel_material_plastic_stream_texels, grid, block, 0, stream, p); break;
		case 2: hipLaunchKernelGGL(kernel_material_dielectric_stream, grid, block, 0, stream, p); break;
		case 3: hipLaunchKernelGGL(kernel_material_conductor_stream,  grid, block, 0, stream, p); break;
	}
}
void rt_launch_ambient_occlusion(const RtParams & p, int sample_index, float ao_radius, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_ambient_occlusion, dim3(2048), dim3(RT_SHADE_BLOCK), 0, stream, p, sample_index, ao_radius);
}
void rt_launch_random(const RtParams & p, int dimension, const unsigned * pixel_indices, int count, unsigned bounce, unsigned sample_index, float2 * out, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_random, dim3((count + 255) / 256), dim3(256), 0, stream, p, dimension, pixel_indices, count, bounce, sample_index, out);
}
