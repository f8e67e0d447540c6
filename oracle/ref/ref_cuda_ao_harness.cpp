// oracle/ref/ref_cuda_ao_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The reference's ambient-occlusion integrator, Src/CUDA/AO.cu, compiled verbatim for the host CPU and run one CUDA
// thread at a time (see ref_cuda_harness.cpp for the method). AO.cu defines device globals with the same names as
// Pathtracer.cu (accumulator, camera, buffer_sizes, ...), so it lives in a shared object of its own, built with
// hidden visibility. The launch sequence of AO::render (Renderer/Integrators/AO.cpp:148-200) is restated below.
#include "ref_cuda_common.h"

#include "AO.cu"

#include <vector>
#include "../oracle.h"

#define REF_API extern "C" __attribute__((visibility("default")))

uint2    shared_stack_bvh8[SHARED_STACK_SIZE * WARP_SIZE];
int      shared_stack_bvh2[SHARED_STACK_SIZE * WARP_SIZE];
unsigned shared_stack_bvh4[SHARED_STACK_SIZE * WARP_SIZE * 2];

namespace {
struct SurfaceObject { unsigned char * data; int pitch_bytes, height; };
}
// AO.cu samples no texture; these exist only because Util.h's Texture<T> wrappers name them
extern "C" void grt_tex_fetch_1d  (cudaTextureObject_t, float, float out[4]) { out[0] = out[1] = out[2] = out[3] = 0.0f; }
extern "C" void grt_tex_fetch_2d  (cudaTextureObject_t, float, float, float out[4]) { out[0] = out[1] = out[2] = out[3] = 0.0f; }
extern "C" void grt_tex_fetch_3d  (cudaTextureObject_t, float, float, float, float out[4]) { out[0] = out[1] = out[2] = out[3] = 0.0f; }
extern "C" void grt_tex_fetch_lod (cudaTextureObject_t, float, float, float, float out[4]) { out[0] = out[1] = out[2] = out[3] = 0.0f; }
extern "C" void grt_tex_fetch_grad(cudaTextureObject_t, float, float, const float[2], const float[2], float out[4]) { out[0] = out[1] = out[2] = out[3] = 0.0f; }
extern "C" void grt_surf_read(cudaSurfaceObject_t s, int x_bytes, int y, int, void * dst, int bytes) {
	const SurfaceObject & o = *reinterpret_cast<const SurfaceObject *>(s);
	if (x_bytes < 0) x_bytes = 0;
	if (x_bytes > o.pitch_bytes - bytes) x_bytes = o.pitch_bytes - bytes;
	if (y < 0) y = 0;
	if (y > o.height - 1) y = o.height - 1;
	memcpy(dst, o.data + size_t(y) * o.pitch_bytes + x_bytes, size_t(bytes));
}
extern "C" void grt_surf_write(cudaSurfaceObject_t s, int x_bytes, int y, int, const void * src, int bytes) {
	const SurfaceObject & o = *reinterpret_cast<const SurfaceObject *>(s);
	memcpy(o.data + size_t(y) * o.pitch_bytes + x_bytes, src, size_t(bytes));
}

namespace {
struct Frame {
	const oracle_scene * scene;
	std::vector<std::vector<unsigned char>> pool;
	std::vector<float4> accumulator_image;
	SurfaceObject accumulator_surface;

	template<typename T> T * alloc(size_t count) { pool.emplace_back(count * sizeof(T) + 64, (unsigned char)0); return reinterpret_cast<T *>(pool.back().data()); }
	Vector3_SoA soa(size_t n) { Vector3_SoA v; v.x = alloc<float>(n); v.y = alloc<float>(n); v.z = alloc<float>(n); return v; }
};

template<typename Kernel, typename... Args> void launch_1d(int threads, Kernel kernel, Args... args) {
	blockDim = grt_dim3(); gridDim = grt_dim3(); threadIdx = grt_dim3 { 0, 0, 0 }; blockIdx = grt_dim3 { 0, 0, 0 };
	for (int i = 0; i < threads; i++) { blockIdx.x = unsigned(i); kernel(args...); }
}
template<typename Kernel, typename... Args> void launch_2d(int width, int height, Kernel kernel, Args... args) {
	blockDim = grt_dim3(); gridDim = grt_dim3(); threadIdx = grt_dim3 { 0, 0, 0 }; blockIdx = grt_dim3 { 0, 0, 0 };
	for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) { blockIdx.x = unsigned(x); blockIdx.y = unsigned(y); kernel(args...); }
}
template<typename Kernel> void launch_persistent(Kernel kernel) {
	blockDim = grt_dim3(); gridDim = grt_dim3(); threadIdx = grt_dim3 { 0, 0, 0 }; blockIdx = grt_dim3 { 0, 0, 0 };
	kernel();
}
}

REF_API void * ref_ao_frame_create(const oracle_scene * s) {
	Frame * f = new Frame();
	f->scene = s;
	screen_width = s->screen_width; screen_pitch = s->screen_pitch; screen_height = s->screen_height;
	size_t pixels = size_t(s->screen_pitch) * s->screen_height;

	config = GPUConfig();
	config.reconstruction_filter = ReconstructionFilter(s->config.reconstruction_filter);
	config.aov_mask = s->config.aov_mask;
	config.enable_svgf = false;
	memcpy(&camera, &s->camera, sizeof(Camera));

	triangles  = reinterpret_cast<const Triangle *>(s->triangles);
	bvh8_nodes = reinterpret_cast<const BVH8Node *>(s->bvh8_nodes);
	bvh2_nodes = reinterpret_cast<BVH2Node *>(const_cast<uint8_t *>(s->bvh2_nodes));
	bvh4_nodes = reinterpret_cast<BVH4Node *>(const_cast<uint8_t *>(s->bvh4_nodes));
	mesh_bvh_root_indices = const_cast<int *>(s->mesh_bvh_root_indices);
	mesh_material_ids     = const_cast<int *>(s->mesh_material_ids);
	mesh_transforms      = reinterpret_cast<Matrix3x4 *>(const_cast<float *>(s->mesh_transforms));
	mesh_transforms_inv  = reinterpret_cast<Matrix3x4 *>(const_cast<float *>(s->mesh_transforms_inv));
	mesh_transforms_prev = reinterpret_cast<Matrix3x4 *>(const_cast<float *>(s->mesh_transforms_prev));
	pmj_samples = reinterpret_cast<float2 *>(const_cast<float *>(s->pmj_samples));
	blue_noise_textures = reinterpret_cast<uchar2 *>(const_cast<uint8_t *>(s->blue_noise));

	for (int a = 0; a < int(AOVType::COUNT); a++) {
		bool enabled = a == int(AOVType::RADIANCE) || (s->config.aov_mask & (1u << a));
		aovs[a].framebuffer = enabled ? f->alloc<float4>(pixels) : nullptr;
		aovs[a].accumulator = enabled ? f->alloc<float4>(pixels) : nullptr;
	}
	f->accumulator_image.assign(pixels, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
	f->accumulator_surface = { reinterpret_cast<unsigned char *>(f->accumulator_image.data()), int(s->screen_pitch * sizeof(float4)), s->screen_height };
	accumulator.surface = reinterpret_cast<cudaSurfaceObject_t>(&f->accumulator_surface);

	ray_buffer_trace.traversal_data.ray_origin = f->soa(BATCH_SIZE); ray_buffer_trace.traversal_data.ray_direction = f->soa(BATCH_SIZE);
	ray_buffer_trace.traversal_data.hits.hits = f->alloc<uint4>(BATCH_SIZE);
	ray_buffer_trace.pixel_index = f->alloc<int>(BATCH_SIZE);
	ray_buffer_shadow.traversal_data.ray_origin = f->soa(BATCH_SIZE); ray_buffer_shadow.traversal_data.ray_direction = f->soa(BATCH_SIZE);
	ray_buffer_shadow.traversal_data.max_distance = f->alloc<float>(BATCH_SIZE);
	ray_buffer_shadow.pixel_index = f->alloc<int>(BATCH_SIZE);
	pixel_query = { INVALID, INVALID, INVALID };
	return f;
}

REF_API void ref_ao_frame_free(void * frame) { delete static_cast<Frame *>(frame); }

// AO::render (AO.cpp:148-200) for one sample; counters_out (may be NULL) = { primary rays, occlusion rays }
REF_API void ref_ao_render_sample(void * frame, int sample_index, float ao_radius, int * counters_out) {
	Frame * f = static_cast<Frame *>(frame);
	const oracle_scene * s = f->scene;
	int pixel_count = s->screen_width * s->screen_height;
	int batch_size  = pixel_count < BATCH_SIZE ? pixel_count : BATCH_SIZE;
	if (counters_out) counters_out[0] = counters_out[1] = 0;
	for (int pixels_left = pixel_count; pixels_left > 0; pixels_left -= batch_size) {
		int pixel_offset = pixel_count - pixels_left;
		int count = batch_size < pixels_left ? batch_size : pixels_left;
		memset(&buffer_sizes, 0, sizeof(buffer_sizes));
		buffer_sizes.trace = count;
		launch_1d(BATCH_SIZE, kernel_generate, sample_index, pixel_offset, count);
		switch (s->bvh_type) {
			case 2:  launch_persistent(kernel_trace_bvh2); break;
			case 4:  launch_persistent(kernel_trace_bvh4); break;
			default: launch_persistent(kernel_trace_bvh8);
		}
		launch_1d(BATCH_SIZE, kernel_ambient_occlusion, sample_index, ao_radius);
		switch (s->bvh_type) {
			case 2:  launch_persistent(kernel_trace_shadow_bvh2); break;
			case 4:  launch_persistent(kernel_trace_shadow_bvh4); break;
			default: launch_persistent(kernel_trace_shadow_bvh8);
		}
		if (counters_out) { counters_out[0] += buffer_sizes.trace; counters_out[1] += buffer_sizes.shadow; }
	}
	launch_2d(s->screen_pitch, s->screen_height, kernel_accumulate, float(sample_index));
	size_t pixels = size_t(s->screen_pitch) * s->screen_height;
	for (int a = 0; a < int(AOVType::COUNT); a++) if (aovs[a].framebuffer) memset(aovs[a].framebuffer, 0, pixels * sizeof(float4));
}

REF_API void ref_ao_read_frame(void * frame, float * dst) { Frame * f = static_cast<Frame *>(frame); memcpy(dst, f->accumulator_image.data(), f->accumulator_image.size() * sizeof(float4)); }
