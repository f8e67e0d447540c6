// oracle/ref/ref_cuda_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Runs the reference's OWN device code -- /root/reference/Src/CUDA/Pathtracer.cu and every header it includes,
// compiled verbatim from where they lie -- on the host CPU, one CUDA thread at a time, over the same staged
// arrays that the product hands to the MI355X (the oracle_scene of oracle/oracle.h). This is the pin for the
// restated oracle (oracle/*.cpp) and, through it, for the HIP kernels: the same frame rendered by the
// reference's kernels and by the oracle must agree to floating-point noise.
//
// What is NOT the reference here, and why:
//  * cuda_shim/cuda_on_cpu.h gives the CUDA keywords and built-ins their documented meaning on a CPU.
//  * The six helpers of Util.h:280-341 are PTX inline assembly. They are compiled under other names (inline,
//    never referenced, hence never emitted) and replaced below by C++ with the semantics of the PTX
//    instructions they wrap (prmt.b32 with selector 0xBA98, bfind.u32, vmin / vmax .s32.s32.s32 with a second
//    min / max).
//  * Texture and surface fetches go to the oracle's software texture unit: NVIDIA's
//    texture unit has no definition to run.
//  * The launch sequence of Pathtracer::render (Renderer/Integrators/Pathtracer.cpp:738-855) is host code that
//    cannot be compiled here (it needs the CUDA driver API); ref_cuda_render_sample below restates its few lines.
#include "ref_cuda_common.h"

#include "Pathtracer.cu"

#include <vector>
#include "../oracle.h"

// "extern __shared__" stacks of the traversal kernels: one warp of one lane (threadIdx = 0)
uint2 shared_stack_bvh8[SHARED_STACK_SIZE * WARP_SIZE];
int   shared_stack_bvh2[SHARED_STACK_SIZE * WARP_SIZE];
unsigned shared_stack_bvh4[SHARED_STACK_SIZE * WARP_SIZE * 2];

// ---- texture / surface objects ----------------------------------------------------------------------------------
extern "C" {
	void  oracle_tex2d(const oracle_texture * tex, float s, float t, float out[4]);
	void  oracle_tex2d_lod(const oracle_texture * tex, float s, float t, float lod, float out[4]);
	void  oracle_tex2d_grad(const oracle_texture * tex, float s, float t, const float dx[2], const float dy[2], float out[4]);
	float oracle_lut_1d(const float * lut, int nx, float s);
	float oracle_lut_2d(const float * lut, int nx, int ny, float s, float t);
	float oracle_lut_3d(const float * lut, int nx, int ny, int nz, float s, float t, float r);
	void  oracle_image_bilinear_clamp(const float * rgba, int width, int height, float u, float v, float out[4]);
}

namespace {
struct TextureObject {
	enum Kind { MATERIAL, SKY, LUT } kind;
	const oracle_texture * material = nullptr;
	const float * data = nullptr;
	int nx = 0, ny = 0, nz = 0;
};
struct SurfaceObject { unsigned char * data; int pitch_bytes, height; int depth = 1; };

const TextureObject & texture_of(cudaTextureObject_t t) { return *reinterpret_cast<const TextureObject *>(t); }
}

extern "C" void grt_tex_fetch_1d(cudaTextureObject_t t, float s, float out[4]) {
	const TextureObject & o = texture_of(t);
	out[0] = oracle_lut_1d(o.data, o.nx, s); out[1] = out[2] = out[3] = 0.0f;
}
extern "C" void grt_tex_fetch_2d(cudaTextureObject_t t, float s, float u, float out[4]) {
	const TextureObject & o = texture_of(t);
	switch (o.kind) {
		case TextureObject::MATERIAL: oracle_tex2d(o.material, s, u, out); break;
		case TextureObject::SKY:      oracle_image_bilinear_clamp(o.data, o.nx, o.ny, s, u, out); break;
		default:                      out[0] = oracle_lut_2d(o.data, o.nx, o.ny, s, u); out[1] = out[2] = out[3] = 0.0f;
	}
}
extern "C" void grt_tex_fetch_3d(cudaTextureObject_t t, float s, float u, float r, float out[4]) {
	const TextureObject & o = texture_of(t);
	out[0] = oracle_lut_3d(o.data, o.nx, o.ny, o.nz, s, u, r); out[1] = out[2] = out[3] = 0.0f;
}
extern "C" void grt_tex_fetch_lod(cudaTextureObject_t t, float s, float u, float lod, float out[4]) { oracle_tex2d_lod(texture_of(t).material, s, u, lod, out); }
extern "C" void grt_tex_fetch_grad(cudaTextureObject_t t, float s, float u, const float dx[2], const float dy[2], float out[4]) { oracle_tex2d_grad(texture_of(t).material, s, u, dx, dy, out); }
extern "C" void grt_surf_read(cudaSurfaceObject_t s, int x_bytes, int y, int z, void * dst, int bytes) {
	// every read in the reference uses cudaBoundaryModeClamp: coordinates outside the surface read its border texel
	const SurfaceObject & o = *reinterpret_cast<const SurfaceObject *>(s);
	if (x_bytes < 0) x_bytes = 0;
	if (x_bytes > o.pitch_bytes - bytes) x_bytes = o.pitch_bytes - bytes;
	if (y < 0) y = 0;
	if (y > o.height - 1) y = o.height - 1;
	if (z < 0) z = 0;
	if (z > o.depth - 1) z = o.depth - 1;
	memcpy(dst, o.data + (size_t(z) * o.height + size_t(y)) * o.pitch_bytes + x_bytes, size_t(bytes));
}
extern "C" void grt_surf_write(cudaSurfaceObject_t s, int x_bytes, int y, int z, const void * src, int bytes) {
	const SurfaceObject & o = *reinterpret_cast<const SurfaceObject *>(s);
	memcpy(o.data + (size_t(z) * o.height + size_t(y)) * o.pitch_bytes + x_bytes, src, size_t(bytes));
}

// ---- one frame's worth of device state --------------------------------------------------------------------------
namespace {

template<typename T> T * alloc(std::vector<std::vector<unsigned char>> & pool, size_t count) {
	pool.emplace_back(count * sizeof(T) + 64, (unsigned char)0);
	return reinterpret_cast<T *>(pool.back().data());
}

struct Frame {
	const oracle_scene * scene;
	std::vector<std::vector<unsigned char>> pool;
	std::vector<TextureObject> texture_objects;
	std::vector<Texture<float4>> texture_table;
	TextureObject sky_object, lut_objects[6];
	SurfaceObject accumulator_surface;
	std::vector<float4> accumulator_image;
	SurfaceObject gbuffer_surfaces[3];
	std::vector<MaterialBuffer> material_buffers;
	int batch_capacity = 0;
	bool has_diffuse = false, has_plastic = false, has_dielectric = false, has_conductor = false, has_lights = false;

	Vector3_SoA soa(size_t n) { Vector3_SoA v; v.x = alloc<float>(pool, n); v.y = alloc<float>(pool, n); v.z = alloc<float>(pool, n); return v; }

	void init_trace_buffer(TraceBuffer & b, size_t n) {
		b.traversal_data.ray_origin = soa(n); b.traversal_data.ray_direction = soa(n);
		b.traversal_data.hits.hits = alloc<uint4>(pool, n);
		b.cone_angle = alloc<float>(pool, n); b.cone_width = alloc<float>(pool, n);
		b.medium = alloc<int>(pool, n);
		b.pixel_index_and_flags = alloc<unsigned>(pool, n);
		b.throughput = soa(n);
		b.last_pdf = alloc<float>(pool, n);
	}
	void init_material_buffer(MaterialBuffer & b, size_t n) {
		b.ray_direction = soa(n);
		b.hits.hits = alloc<uint4>(pool, n);
		b.cone_angle = alloc<float>(pool, n); b.cone_width = alloc<float>(pool, n);
		b.medium = alloc<int>(pool, n);
		b.pixel_index_and_flags = alloc<int>(pool, n);
		b.throughput = soa(n);
	}
};

// launches: one thread at a time, threadIdx = 0, the block index carries the thread index
template<typename Kernel, typename... Args> void launch_1d(int threads, Kernel kernel, Args... args) {
	blockDim = grt_dim3(); gridDim = grt_dim3(); threadIdx = grt_dim3 { 0, 0, 0 }; blockIdx = grt_dim3 { 0, 0, 0 };
	gridDim.x = unsigned(threads);
	for (int i = 0; i < threads; i++) { blockIdx.x = unsigned(i); kernel(args...); }
}
template<typename Kernel, typename... Args> void launch_2d(int width, int height, Kernel kernel, Args... args) {
	blockDim = grt_dim3(); gridDim = grt_dim3(); threadIdx = grt_dim3 { 0, 0, 0 }; blockIdx = grt_dim3 { 0, 0, 0 };
	gridDim.x = unsigned(width); gridDim.y = unsigned(height);
	for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) { blockIdx.x = unsigned(x); blockIdx.y = unsigned(y); kernel(args...); }
}
template<typename Kernel, typename... Args> void launch_persistent(Kernel kernel, Args... args) { // one "warp" of one lane drains the whole queue
	blockDim = grt_dim3(); gridDim = grt_dim3(); threadIdx = grt_dim3 { 0, 0, 0 }; blockIdx = grt_dim3 { 0, 0, 0 };
	kernel(args...);
}

} // namespace

extern "C" {

// Binds the reference's device globals to the arrays of `s` and allocates its wavefront buffers
// (Pathtracer.cpp: init_geometry / init_materials / resize_init, Integrator.cpp:101-283). Frame lifetime.
void * ref_cuda_frame_create(const oracle_scene * s) {
	Frame * f = new Frame();
	f->scene = s;

	screen_width = s->screen_width; screen_pitch = s->screen_pitch; screen_height = s->screen_height;
	size_t pixels = size_t(s->screen_pitch) * s->screen_height;

	config = GPUConfig();
	config.reconstruction_filter = ReconstructionFilter(s->config.reconstruction_filter);
	config.aov_mask    = s->config.aov_mask;
	config.num_bounces = s->config.num_bounces;
	config.enable_mipmapping                   = s->config.enable_mipmapping != 0;
	config.enable_next_event_estimation        = s->config.enable_next_event_estimation != 0;
	config.enable_multiple_importance_sampling = s->config.enable_multiple_importance_sampling != 0;
	config.enable_russian_roulette             = s->config.enable_russian_roulette != 0;
	config.enable_svgf             = s->config.enable_svgf != 0;
	config.enable_spatial_variance = s->config.enable_spatial_variance != 0;
	config.enable_taa              = s->config.enable_taa != 0;
	config.alpha_colour = s->config.alpha_colour; config.alpha_moment = s->config.alpha_moment;
	config.num_atrous_iterations = s->config.num_atrous_iterations;
	config.sigma_z = s->config.sigma_z; config.sigma_n = s->config.sigma_n; config.sigma_l = s->config.sigma_l;

	static_assert(sizeof(Camera) == sizeof(rt_camera), "camera layouts differ");
	memcpy(&camera, &s->camera, sizeof(Camera));

	// geometry and instances: the staged arrays have the reference's device layouts
	static_assert(sizeof(Triangle) == 96 && sizeof(BVH8Node) == 80 && sizeof(Matrix3x4) == 48, "device struct sizes");
	triangles  = reinterpret_cast<const Triangle *>(s->triangles);
	bvh8_nodes = reinterpret_cast<const BVH8Node *>(s->bvh8_nodes);
	bvh2_nodes = reinterpret_cast<BVH2Node *>(const_cast<uint8_t *>(s->bvh2_nodes));
	bvh4_nodes = reinterpret_cast<BVH4Node *>(const_cast<uint8_t *>(s->bvh4_nodes));
	mesh_bvh_root_indices = const_cast<int *>(s->mesh_bvh_root_indices);
	mesh_material_ids     = const_cast<int *>(s->mesh_material_ids);
	mesh_transforms      = reinterpret_cast<Matrix3x4 *>(const_cast<float *>(s->mesh_transforms));
	mesh_transforms_inv  = reinterpret_cast<Matrix3x4 *>(const_cast<float *>(s->mesh_transforms_inv));
	mesh_transforms_prev = reinterpret_cast<Matrix3x4 *>(const_cast<float *>(s->mesh_transforms_prev));

	static_assert(sizeof(Material) == 32 && sizeof(Medium) == 32, "material / medium sizes");
	material_types = reinterpret_cast<const MaterialType *>(s->material_types);
	materials      = reinterpret_cast<const Material *>(s->materials);
	media          = reinterpret_cast<Medium *>(const_cast<float *>(s->media));
	for (int i = 0; i < s->material_count; i++) {
		switch (MaterialType(s->material_types[i])) {
			case MaterialType::LIGHT:      f->has_lights     = true; break;
			case MaterialType::DIFFUSE:    f->has_diffuse    = true; break;
			case MaterialType::PLASTIC:    f->has_plastic    = true; break;
			case MaterialType::DIELECTRIC: f->has_dielectric = true; break;
			case MaterialType::CONDUCTOR:  f->has_conductor  = true; break;
		}
	}
	f->has_lights = s->light_mesh_count > 0;

	// textures
	f->texture_objects.resize(size_t(s->texture_count));
	f->texture_table  .resize(size_t(s->texture_count));
	for (int i = 0; i < s->texture_count; i++) {
		f->texture_objects[i].kind = TextureObject::MATERIAL;
		f->texture_objects[i].material = &s->textures[i];
		f->texture_table[i].texture = reinterpret_cast<cudaTextureObject_t>(&f->texture_objects[i]);
		int lod_width  = s->textures[i].lod_width  > 0 ? s->textures[i].lod_width  : s->textures[i].width;
		int lod_height = s->textures[i].lod_height > 0 ? s->textures[i].lod_height : s->textures[i].height;
		f->texture_table[i].lod_bias = 0.5f * log2f(float(lod_width * lod_height)); // Integrator.cpp:95
	}
	textures = f->texture_table.data();

	f->sky_object.kind = TextureObject::SKY; f->sky_object.data = s->sky; f->sky_object.nx = s->sky_width; f->sky_object.ny = s->sky_height;
	sky_texture.texture = reinterpret_cast<cudaTextureObject_t>(&f->sky_object);
	sky_scale = s->sky_scale;

	struct { Texture<float> * global; const float * data; int nx, ny, nz; } luts[6] = {
		{ &lut_dielectric_directional_albedo_enter, s->lut_dielectric_directional_albedo_enter, 16, 16, 16 },
		{ &lut_dielectric_directional_albedo_leave, s->lut_dielectric_directional_albedo_leave, 16, 16, 16 },
		{ &lut_dielectric_albedo_enter, s->lut_dielectric_albedo_enter, 16, 16, 1 },
		{ &lut_dielectric_albedo_leave, s->lut_dielectric_albedo_leave, 16, 16, 1 },
		{ &lut_conductor_directional_albedo, s->lut_conductor_directional_albedo, 32, 32, 1 },
		{ &lut_conductor_albedo, s->lut_conductor_albedo, 32, 1, 1 },
	};
	for (int i = 0; i < 6; i++) {
		f->lut_objects[i].kind = TextureObject::LUT; f->lut_objects[i].data = luts[i].data;
		f->lut_objects[i].nx = luts[i].nx; f->lut_objects[i].ny = luts[i].ny; f->lut_objects[i].nz = luts[i].nz;
		luts[i].global->texture = reinterpret_cast<cudaTextureObject_t>(&f->lut_objects[i]);
	}

	// lights and random numbers
	lights_total_weight = s->lights_total_weight;
	light_triangle_indices = s->light_triangle_indices;
	light_triangle_cumulative_probability = s->light_triangle_cumulative_probability;
	light_mesh_count = s->light_mesh_count;
	light_mesh_cumulative_probability = s->light_mesh_cumulative_probability;
	light_mesh_triangle_span = reinterpret_cast<const int2 *>(s->light_mesh_triangle_span);
	light_mesh_transform_indices = s->light_mesh_transform_indices;
	pmj_samples = reinterpret_cast<float2 *>(const_cast<float *>(s->pmj_samples));
	blue_noise_textures = reinterpret_cast<uchar2 *>(const_cast<uint8_t *>(s->blue_noise));

	// AOVs (Integrator.cpp: init_aovs / aov_enable) and the display accumulator
	for (int a = 0; a < int(AOVType::COUNT); a++) {
		bool enabled = a == int(AOVType::RADIANCE) || (s->config.aov_mask & (1u << a));
		if (s->config.enable_svgf && (a == int(AOVType::RADIANCE_DIRECT) || a == int(AOVType::RADIANCE_INDIRECT) || a == int(AOVType::ALBEDO))) enabled = true; // svgf_init
		aovs[a].framebuffer = enabled ? alloc<float4>(f->pool, pixels) : nullptr;
		aovs[a].accumulator = enabled ? alloc<float4>(f->pool, pixels) : nullptr;
	}
	f->accumulator_image.assign(pixels, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
	f->accumulator_surface = { reinterpret_cast<unsigned char *>(f->accumulator_image.data()), int(s->screen_pitch * sizeof(float4)), s->screen_height };
	accumulator.surface = reinterpret_cast<cudaSurfaceObject_t>(&f->accumulator_surface);

	if (s->config.enable_svgf) { // Pathtracer::svgf_init (Pathtracer.cpp:316-357)
		unsigned char * normal_and_depth = alloc<unsigned char>(f->pool, pixels * sizeof(float4));
		unsigned char * ids              = alloc<unsigned char>(f->pool, pixels * sizeof(int2));
		unsigned char * position_prev    = alloc<unsigned char>(f->pool, pixels * sizeof(float2));
		f->gbuffer_surfaces[0] = { normal_and_depth, int(s->screen_pitch * sizeof(float4)), s->screen_height };
		f->gbuffer_surfaces[1] = { ids,              int(s->screen_pitch * sizeof(int2)),   s->screen_height };
		f->gbuffer_surfaces[2] = { position_prev,    int(s->screen_pitch * sizeof(float2)), s->screen_height };
		gbuffer_normal_and_depth       .surface = reinterpret_cast<cudaSurfaceObject_t>(&f->gbuffer_surfaces[0]);
		gbuffer_mesh_id_and_triangle_id.surface = reinterpret_cast<cudaSurfaceObject_t>(&f->gbuffer_surfaces[1]);
		gbuffer_screen_position_prev   .surface = reinterpret_cast<cudaSurfaceObject_t>(&f->gbuffer_surfaces[2]);
		frame_buffer_moment      = alloc<float4>(f->pool, pixels);
		history_length           = alloc<int>   (f->pool, pixels);
		history_direct           = alloc<float4>(f->pool, pixels);
		history_indirect         = alloc<float4>(f->pool, pixels);
		history_moment           = alloc<float4>(f->pool, pixels);
		history_normal_and_depth = alloc<float4>(f->pool, pixels);
		taa_frame_prev           = alloc<float4>(f->pool, pixels);
		taa_frame_curr           = alloc<float4>(f->pool, pixels);
	}

	// wavefront buffers (Pathtracer.cpp:540-660): BATCH_SIZE entries each; two material types share one
	// allocation, the second filling it from the back (PackedMaterialBuffer's low bit)
	f->batch_capacity = BATCH_SIZE;
	f->init_trace_buffer(ray_buffer_trace_0, BATCH_SIZE);
	f->init_trace_buffer(ray_buffer_trace_1, BATCH_SIZE);
	ray_buffer_shadow.traversal_data.ray_origin = f->soa(BATCH_SIZE); ray_buffer_shadow.traversal_data.ray_direction = f->soa(BATCH_SIZE);
	ray_buffer_shadow.traversal_data.max_distance = alloc<float>(f->pool, BATCH_SIZE);
	ray_buffer_shadow.illumination_and_pixel_index = alloc<float4>(f->pool, BATCH_SIZE);

	int needed = (int(f->has_diffuse) + int(f->has_plastic) + int(f->has_dielectric) + int(f->has_conductor) + 1) / 2;
	f->material_buffers.resize(size_t(needed));
	for (MaterialBuffer & b : f->material_buffers) f->init_material_buffer(b, BATCH_SIZE);
	int index = 0;
	auto bind = [&](PackedMaterialBuffer & packed) { packed = PackedMaterialBuffer(reinterpret_cast<uintptr_t>(&f->material_buffers[size_t(index / 2)]) | uintptr_t(index & 1)); index++; };
	if (f->has_diffuse)    bind(material_buffer_diffuse);
	if (f->has_plastic)    bind(material_buffer_plastic);
	if (f->has_dielectric) bind(material_buffer_dielectric);
	if (f->has_conductor)  bind(material_buffer_conductor);

	pixel_query = { INVALID, INVALID, INVALID };
	return f;
}

void ref_cuda_frame_free(void * frame) { delete static_cast<Frame *>(frame); }

// One sample of the whole frame: the launch sequence of Pathtracer::render (Pathtracer.cpp:738-855) without SVGF.
// counters_out (may be NULL): 6 * MAX_BOUNCES ints = trace, diffuse, plastic, dielectric, conductor, shadow, summed over batches.
void ref_cuda_render_sample(void * frame, int sample_index, int * counters_out) {
	Frame * f = static_cast<Frame *>(frame);
	const oracle_scene * s = f->scene;
	int pixel_count = s->screen_width * s->screen_height;
	int batch_size  = pixel_count < BATCH_SIZE ? pixel_count : BATCH_SIZE;
	if (counters_out) memset(counters_out, 0, 6 * MAX_BOUNCES * sizeof(int));

	memcpy(&camera, &s->camera, sizeof(Camera)); // the camera constant is re-uploaded whenever it moved (Integrator.cpp:454-481)
	static_assert(sizeof(SVGFData) == 128, "two row-major 4x4 matrices");
	memcpy(&svgf_data.view_projection,      s->view_projection,      64); // uploaded by Pathtracer::update before the frame (Pathtracer.cpp:707-717)
	memcpy(&svgf_data.view_projection_prev, s->view_projection_prev, 64);

	int pixels_left = pixel_count;
	while (pixels_left > 0) {
		int pixel_offset = pixel_count - pixels_left;
		int count = batch_size < pixels_left ? batch_size : pixels_left;
		memset(&buffer_sizes, 0, sizeof(buffer_sizes));
		buffer_sizes.trace[0] = count;

		launch_1d(BATCH_SIZE, kernel_generate, sample_index, pixel_offset, count);
		for (int bounce = 0; bounce < config.num_bounces; bounce++) {
			switch (s->bvh_type) {
				case 2:  launch_persistent(kernel_trace_bvh2, bounce); break;
				case 4:  launch_persistent(kernel_trace_bvh4, bounce); break;
				default: launch_persistent(kernel_trace_bvh8, bounce);
			}
			launch_1d(BATCH_SIZE, kernel_sort, bounce, sample_index);
			if (f->has_diffuse)    launch_1d(BATCH_SIZE, kernel_material_diffuse,    bounce, sample_index);
			if (f->has_plastic)    launch_1d(BATCH_SIZE, kernel_material_plastic,    bounce, sample_index);
			if (f->has_dielectric) launch_1d(BATCH_SIZE, kernel_material_dielectric, bounce, sample_index);
			if (f->has_conductor)  launch_1d(BATCH_SIZE, kernel_material_conductor,  bounce, sample_index);
			if (f->has_lights && config.enable_next_event_estimation) {
				switch (s->bvh_type) {
					case 2:  launch_persistent(kernel_trace_shadow_bvh2, bounce); break;
					case 4:  launch_persistent(kernel_trace_shadow_bvh4, bounce); break;
					default: launch_persistent(kernel_trace_shadow_bvh8, bounce);
				}
			}
		}
		if (counters_out) {
			const int * groups[6] = { buffer_sizes.trace, buffer_sizes.diffuse, buffer_sizes.plastic, buffer_sizes.dielectric, buffer_sizes.conductor, buffer_sizes.shadow };
			for (int g = 0; g < 6; g++) for (int b = 0; b < MAX_BOUNCES; b++) counters_out[g * MAX_BOUNCES + b] += groups[g][b];
		}
		pixels_left -= batch_size;
	}
	if (config.enable_svgf) { // Pathtracer.cpp:796-838
		int w = s->screen_pitch, h = s->screen_height;
		launch_2d(w, h, kernel_svgf_reproject, sample_index);
		float4 * direct_in    = aovs[int(AOVType::RADIANCE_DIRECT)]  .framebuffer;
		float4 * indirect_in  = aovs[int(AOVType::RADIANCE_INDIRECT)].framebuffer;
		float4 * direct_out   = aovs[int(AOVType::RADIANCE_DIRECT)]  .accumulator;
		float4 * indirect_out = aovs[int(AOVType::RADIANCE_INDIRECT)].accumulator;
		if (config.enable_spatial_variance) {
			launch_2d(w, h, kernel_svgf_variance, (const float4 *)direct_in, (const float4 *)indirect_in, direct_out, indirect_out);
			std::swap(direct_in, direct_out); std::swap(indirect_in, indirect_out);
		}
		for (int i = 0; i < config.num_atrous_iterations; i++) {
			launch_2d(w, h, kernel_svgf_atrous, (const float4 *)direct_in, (const float4 *)indirect_in, direct_out, indirect_out, 1 << i);
			std::swap(direct_in, direct_out); std::swap(indirect_in, indirect_out);
		}
		launch_2d(w, h, kernel_svgf_finalize, (const float4 *)direct_in, (const float4 *)indirect_in);
		if (config.enable_taa) {
			launch_2d(w, h, kernel_taa, sample_index);
			launch_2d(w, h, kernel_taa_finalize);
		}
	} else {
		launch_2d(s->screen_pitch, s->screen_height, kernel_accumulate, float(sample_index));
	}

	// aovs_clear_to_zero (Integrator.cpp): the per-frame buffers start the next sample from zero
	size_t pixels = size_t(s->screen_pitch) * s->screen_height;
	for (int a = 0; a < int(AOVType::COUNT); a++) if (aovs[a].framebuffer) memset(aovs[a].framebuffer, 0, pixels * sizeof(float4));
}

// The displayed frame (the `accumulator` surface) and an AOV's accumulator: pitch * height float4
void ref_cuda_read_frame(void * frame, float * dst) { Frame * f = static_cast<Frame *>(frame); memcpy(dst, f->accumulator_image.data(), f->accumulator_image.size() * sizeof(float4)); }
// Kulla-Conty table integration (KullaConty.h:83-240), cells [first, first + count) of the 16^3 dielectric table
// (entering / leaving) or of the 32^2 conductor table, 100 000 samples each as in the reference; and the cosine-weighted
// averages over the last axis. Needs a frame (the kernels draw from pmj_samples / blue_noise_textures).
void ref_cuda_integrate_dielectric_cells(void *, int entering, int first, int count, float * out) {
	std::vector<float> table(16 * 16 * 16, 0.0f);
	SurfaceObject surface = { reinterpret_cast<unsigned char *>(table.data()), int(16 * sizeof(float)), 16, 16 };
	Surface<float> lut; lut.surface = reinterpret_cast<cudaSurfaceObject_t>(&surface);
	blockDim = grt_dim3(); gridDim = grt_dim3(); threadIdx = grt_dim3 { 0, 0, 0 }; blockIdx = grt_dim3 { 0, 0, 0 };
	for (int cell = first; cell < first + count; cell++) { blockIdx.x = unsigned(cell); kernel_integrate_dielectric(entering != 0, lut); }
	memcpy(out, table.data() + first, size_t(count) * sizeof(float));
}
void ref_cuda_integrate_conductor_cells(void *, int first, int count, float * out) {
	std::vector<float> table(32 * 32, 0.0f);
	blockDim = grt_dim3(); gridDim = grt_dim3(); threadIdx = grt_dim3 { 0, 0, 0 }; blockIdx = grt_dim3 { 0, 0, 0 };
	for (int cell = first; cell < first + count; cell++) { blockIdx.x = unsigned(cell); kernel_integrate_conductor(table.data()); }
	memcpy(out, table.data() + first, size_t(count) * sizeof(float));
}
void ref_cuda_average_dielectric(const float * directional_16x16x16, float * albedo_16x16) {
	std::vector<float> in(directional_16x16x16, directional_16x16x16 + 4096);
	SurfaceObject in_surface  = { reinterpret_cast<unsigned char *>(in.data()), int(16 * sizeof(float)), 16, 16 };
	SurfaceObject out_surface = { reinterpret_cast<unsigned char *>(albedo_16x16), int(16 * sizeof(float)), 16, 1 };
	Surface<float> lut_in, lut_out;
	lut_in.surface = reinterpret_cast<cudaSurfaceObject_t>(&in_surface); lut_out.surface = reinterpret_cast<cudaSurfaceObject_t>(&out_surface);
	launch_1d(256, kernel_average_dielectric, lut_in, lut_out);
}
void ref_cuda_average_conductor(const float * directional_32x32, float * albedo_32) {
	launch_1d(32, kernel_average_conductor, directional_32x32, albedo_32);
}

void ref_cuda_read_history_length(void * frame, int * dst) {
	Frame * f = static_cast<Frame *>(frame);
	if (history_length) memcpy(dst, history_length, size_t(f->scene->screen_pitch) * f->scene->screen_height * sizeof(int));
}
int  ref_cuda_read_aov(void * frame, int aov, float * dst) {
	Frame * f = static_cast<Frame *>(frame);
	if (aov < 0 || aov >= int(AOVType::COUNT) || !aovs[aov].accumulator) return 0;
	memcpy(dst, aovs[aov].accumulator, size_t(f->scene->screen_pitch) * f->scene->screen_height * sizeof(float4));
	return 1;
}

} // extern "C"

// Debug / parity aid: primary rays of sample `sample_index` through kernel_generate + kernel_trace_bvh8; hits_out = count * 4 uints
extern "C" void ref_cuda_primary_hits(void * frame, int sample_index, unsigned * hits_out) {
	Frame * f = static_cast<Frame *>(frame);
	const oracle_scene * s = f->scene;
	int count = s->screen_width * s->screen_height;
	if (count > BATCH_SIZE) count = BATCH_SIZE; // the wavefront buffers hold one batch
	memset(&buffer_sizes, 0, sizeof(buffer_sizes));
	buffer_sizes.trace[0] = count;
	launch_1d(BATCH_SIZE, kernel_generate, sample_index, 0, count);
	launch_persistent(kernel_trace_bvh8, 0);
	memcpy(hits_out, ray_buffer_trace_0.traversal_data.hits.hits, size_t(count) * sizeof(uint4));
}

extern "C" void ref_cuda_primary_rays(void * frame, int sample_index, float * origins3xn, float * directions3xn) {
	Frame * f = static_cast<Frame *>(frame);
	const oracle_scene * s = f->scene;
	int count = s->screen_width * s->screen_height;
	if (count > BATCH_SIZE) count = BATCH_SIZE; // the wavefront buffers hold one batch
	memset(&buffer_sizes, 0, sizeof(buffer_sizes));
	buffer_sizes.trace[0] = count;
	launch_1d(BATCH_SIZE, kernel_generate, sample_index, 0, count);
	const TraversalData & t = ray_buffer_trace_0.traversal_data;
	for (int c = 0; c < 3; c++) {
		const float * o = c == 0 ? t.ray_origin.x : (c == 1 ? t.ray_origin.y : t.ray_origin.z);
		const float * d = c == 0 ? t.ray_direction.x : (c == 1 ? t.ray_direction.y : t.ray_direction.z);
		memcpy(origins3xn + size_t(c) * count, o, size_t(count) * 4);
		memcpy(directions3xn + size_t(c) * count, d, size_t(count) * 4);
	}
}
