// oracle/ref/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C-ABI wrapper around the *verbatim* reference BVH builder
// (/root/reference/Src/BVH/**, compiled from where it lies; nothing is copied).
// It yields golden BVHNode2[] / BVHNode8[] / indices[] that the product's own
// builder (gpu-raytracer_amd/host/BVH.cpp) must match byte-for-byte, and it is
// the "reference" CPU baseline timed by bench.py (cpu_baseline.kind="reference").
//
// Entry points used: BVH::create_from_triangles  (Src/BVH/BVH.cpp:14-36)
//                    BVH8Converter::convert       (Src/BVH/Converters/BVH8Converter.cpp:7-22)
//                    SAHBuilder::build(meshes)    (Src/BVH/Builders/SAHBuilder.cpp:102-104)
//                    BVH4Converter::convert       (Src/BVH/Converters/BVH4Converter.cpp:3-78)
//                    SBVHBuilder::build           (Src/BVH/Builders/SBVHBuilder.cpp:13-68, via BVH::create_from_triangles)
//                    BVHCollapser::collapse       (Src/BVH/BVHCollapser.cpp:97-114)
//                    Mipmap::downsample           (Src/Math/Mipmap.cpp:154-160)
#include "Core/Format.h"
#include "BVH/BVH.h"
#include "BVH/Builders/SAHBuilder.h"
#include "BVH/Converters/BVH8Converter.h"
#include "BVH/Converters/BVH4Converter.h"
#include "BVH/BVHCollapser.h"
#include "Renderer/Mesh.h"

#include <vector>
#include <chrono>
#include <atomic>
#include <thread>
#include <unistd.h>
#include <fcntl.h>

// The reference builder prints progress lines (IO::print); mute fd 1 around it.
struct MuteStdout {
	int saved;
	MuteStdout() { fflush(stdout); saved = dup(1); int n = open("/dev/null", O_WRONLY); dup2(n, 1); close(n); }
	~MuteStdout() { fflush(stdout); dup2(saved, 1); close(saved); }
};

// Core/Format.cpp cannot be compiled here (it includes Core/Parser.h whose macros
// need MSVC's empty-__VA_ARGS__ comma elision). Only IO::print's "{}" scanner is
// needed, so provide the one missing symbol.
Format::Spec Format::parse_fmt(StringView fmt) const {
	Spec spec = { }; spec.fmt_end = fmt.end; spec.restart = fmt.end;
	for (const char * c = fmt.start; c < fmt.end; c++) if (*c == '{') { spec.fmt_end = c; while (c < fmt.end && *c != '}') c++; spec.restart = c + 1; break; }
	return spec;
}

// Renderer/Mesh.cpp pulls in the whole asset manager; the TLAS build only reads
// Mesh::aabb, so the trivial constructor is supplied here instead.
Mesh::Mesh(String name, Handle<MeshData> mesh_data_handle, Handle<Material> material_handle) : name(std::move(name)), mesh_data_handle(mesh_data_handle), material_handle(material_handle) { }

static_assert(sizeof(Triangle) == 96, "reference host Triangle is 24 floats");
static_assert(sizeof(BVHNode2) == 32, "BVHNode2");
static_assert(sizeof(BVHNode8) == 80, "BVHNode8");
static_assert(sizeof(BVHNode4) == 128, "BVHNode4");

struct RefBVH {
	BVH2 bvh2;
	BVH8 bvh8;
	BVH4 bvh4;
	double ms_bvh2 = 0.0, ms_bvh8 = 0.0;
};

static double now_ms() {
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" {

// tris24: n * 24 floats {p0,p1,p2,n0,n1,n2,t0,t1,t2}, already run through the
// Triangle-constructor rules (Renderer/Triangle.h:47-93) by the caller.
void * ref_bvh_build_triangles(const float * tris24, int n) {
	cpu_config.bvh_type = BVHType::BVH8;
	Array<Triangle> triangles(n);
	memcpy((void *)triangles.data(), tris24, size_t(n) * sizeof(Triangle));

	RefBVH * r = new RefBVH();
	MuteStdout mute;
	double t0 = now_ms();
	r->bvh2 = BVH::create_from_triangles(triangles);
	double t1 = now_ms();
	BVH8Converter(r->bvh8, r->bvh2).convert();
	double t2 = now_ms();
	BVH4Converter(r->bvh4, r->bvh2).convert();
	r->ms_bvh2 = t1 - t0;
	r->ms_bvh8 = t2 - t1;
	return r;
}

// The reference's load-time schedule for the CPU baseline of bench.py: AssetManager (Assets/AssetManager.cpp:57) owns a
// thread pool of hardware_concurrency workers and gives it ONE JOB PER MESH; a job builds the SAH BVH2 of its mesh
// (BVH::create_from_triangles) and converts it to the 8-wide tree (BVH8Converter). Same code, same granularity; the
// pool itself is replaced by `threads` std::threads pulling mesh indices from an atomic counter (Util/ThreadPool.cpp
// loses wake-ups on jobs this short). Returns the wall-clock milliseconds; totals = { BVH2 nodes, BVH8 nodes }.
double ref_bvh_build_many(const float * const * tris24, const int * counts, int meshes, int threads, long long * totals) {
	cpu_config.bvh_type = BVHType::BVH8;
	if (threads < 1) threads = 1;
	std::vector<RefBVH> results; results.resize(size_t(meshes));
	std::atomic<int> next(0);
	MuteStdout mute;
	double t0 = now_ms();
	auto worker = [&]() {
		for (int m = next.fetch_add(1); m < meshes; m = next.fetch_add(1)) {
			Array<Triangle> triangles(counts[m]);
			memcpy((void *)triangles.data(), tris24[m], size_t(counts[m]) * sizeof(Triangle));
			results[size_t(m)].bvh2 = BVH::create_from_triangles(triangles);
			BVH8Converter(results[size_t(m)].bvh8, results[size_t(m)].bvh2).convert();
		}
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < threads; t++) pool.emplace_back(worker);
	worker();
	for (std::thread & t : pool) t.join();
	double wall = now_ms() - t0;
	if (totals) { totals[0] = totals[1] = 0; for (const RefBVH & r : results) { totals[0] += (long long)r.bvh2.nodes.size(); totals[1] += (long long)r.bvh8.nodes.size(); } }
	return wall;
}

// The binary tree the reference hands to the device for bvh_type = BVH (sbvh = 0) or SBVH
// (sbvh = 1), leaf-collapsed like a file-loaded mesh when collapse != 0
// (Assets/AssetManager.cpp:80-89), and the BVH4 built from it. bvh8 stays empty.
void * ref_bvh_build_binary_variant(const float * tris24, int n, int sbvh, int collapse, float sbvh_alpha) {
	Array<Triangle> triangles(n);
	memcpy((void *)triangles.data(), tris24, size_t(n) * sizeof(Triangle));

	RefBVH * r = new RefBVH();
	MuteStdout mute;
	float saved_alpha = cpu_config.sbvh_alpha;
	cpu_config.bvh_type   = sbvh ? BVHType::SBVH : BVHType::BVH;
	cpu_config.sbvh_alpha = sbvh_alpha;
	double t0 = now_ms();
	r->bvh2 = BVH::create_from_triangles(triangles);
	if (collapse) BVHCollapser::collapse(r->bvh2);
	double t1 = now_ms();
	BVH4Converter(r->bvh4, r->bvh2).convert();
	cpu_config.bvh_type   = BVHType::BVH8;
	cpu_config.sbvh_alpha = saved_alpha;
	r->ms_bvh2 = t1 - t0;
	return r;
}

// BVH::create_from_triangles with cpu_config.enable_bvh_optimization (BVH/BVH.cpp:31-33, BVH/BVHOptimizer.cpp) and the
// device forms converted from the optimised tree. The optimiser seeds its random phase from time(): its output is a
// function of the input only while it selects by measure, which max_batches <= 4 guarantees.
void * ref_bvh_build_optimized(const float * tris24, int n, int sbvh, int max_batches) {
	Array<Triangle> triangles(n);
	memcpy((void *)triangles.data(), tris24, size_t(n) * sizeof(Triangle));

	RefBVH * r = new RefBVH();
	MuteStdout mute;
	int saved_batches = cpu_config.bvh_optimizer_max_num_batches;
	cpu_config.bvh_type = sbvh ? BVHType::SBVH : BVHType::BVH;
	cpu_config.enable_bvh_optimization       = true;
	cpu_config.bvh_optimizer_max_num_batches = max_batches;
	double t0 = now_ms();
	r->bvh2 = BVH::create_from_triangles(triangles);
	double t1 = now_ms();
	if (!sbvh) BVH8Converter(r->bvh8, r->bvh2).convert();
	BVH4Converter(r->bvh4, r->bvh2).convert();
	cpu_config.bvh_type = BVHType::BVH8;
	cpu_config.enable_bvh_optimization       = false;
	cpu_config.bvh_optimizer_max_num_batches = saved_batches;
	r->ms_bvh2 = t1 - t0;
	return r;
}

// aabbs6: n * 6 floats {min.xyz, max.xyz} = Mesh::aabb after Mesh::update().
void * ref_bvh_build_meshes(const float * aabbs6, int n) {
	Array<Mesh> meshes;
	for (int i = 0; i < n; i++) {
		Mesh & m = meshes.emplace_back(String(), Handle<MeshData> { 0 }, Handle<Material> { 0 });
		m.aabb.min = Vector3(aabbs6[6*i+0], aabbs6[6*i+1], aabbs6[6*i+2]);
		m.aabb.max = Vector3(aabbs6[6*i+3], aabbs6[6*i+4], aabbs6[6*i+5]);
	}
	RefBVH * r = new RefBVH();
	// Mirrors Integrator::init_geometry / build_tlas (Src/Renderer/Integrators/Integrator.cpp:243-245,399-402)
	r->bvh2.indices.resize(n);
	r->bvh2.nodes  .resize(size_t(n) * 2);
	double t0 = now_ms();
	SAHBuilder builder(r->bvh2, n);
	builder.build(meshes);
	double t1 = now_ms();
	BVH8Converter(r->bvh8, r->bvh2).convert();
	double t2 = now_ms();
	BVH4Converter(r->bvh4, r->bvh2).convert();
	r->ms_bvh2 = t1 - t0;
	r->ms_bvh8 = t2 - t1;
	return r;
}

int  ref_bvh4_node_count (void * h) { return int(((RefBVH *)h)->bvh4.nodes.size()); }
void ref_bvh4_copy_nodes  (void * h, void * dst) { RefBVH * r = (RefBVH *)h; memcpy(dst, r->bvh4.nodes.data(), r->bvh4.nodes.size() * sizeof(BVHNode4)); }
int  ref_bvh2_node_count (void * h) { return int(((RefBVH *)h)->bvh2.nodes.size()); }
int  ref_bvh2_index_count(void * h) { return int(((RefBVH *)h)->bvh2.indices.size()); }
int  ref_bvh8_node_count (void * h) { return int(((RefBVH *)h)->bvh8.nodes.size()); }
int  ref_bvh8_index_count(void * h) { return int(((RefBVH *)h)->bvh8.indices.size()); }
void ref_bvh2_copy_nodes  (void * h, void * dst) { RefBVH * r = (RefBVH *)h; memcpy(dst, r->bvh2.nodes.data(),   r->bvh2.nodes.size()   * sizeof(BVHNode2)); }
void ref_bvh2_copy_indices(void * h, int  * dst) { RefBVH * r = (RefBVH *)h; memcpy(dst, r->bvh2.indices.data(), r->bvh2.indices.size() * sizeof(int)); }
void ref_bvh8_copy_nodes  (void * h, void * dst) { RefBVH * r = (RefBVH *)h; memcpy(dst, r->bvh8.nodes.data(),   r->bvh8.nodes.size()   * sizeof(BVHNode8)); }
void ref_bvh8_copy_indices(void * h, int  * dst) { RefBVH * r = (RefBVH *)h; memcpy(dst, r->bvh8.indices.data(), r->bvh8.indices.size() * sizeof(int)); }
double ref_bvh_ms_bvh2(void * h) { return ((RefBVH *)h)->ms_bvh2; }
double ref_bvh_ms_bvh8(void * h) { return ((RefBVH *)h)->ms_bvh8; }
void ref_bvh_free(void * h) { delete (RefBVH *)h; }

} // extern "C"

// Mip chain generator, verbatim (Src/Math/Mipmap.cpp:154-160). filter: 0 box, 1 lanczos, 2 kaiser.
// src: width_src * height_src float4, dst: width_dst * height_dst float4.
#include "Math/Mipmap.h"
extern "C" void ref_mipmap_downsample(int filter, int width_src, int height_src, int width_dst, int height_dst, const float * src, float * dst) {
	MipmapFilterType saved = cpu_config.mipmap_filter;
	cpu_config.mipmap_filter = filter == 0 ? MipmapFilterType::BOX : (filter == 1 ? MipmapFilterType::LANCZOS : MipmapFilterType::KAISER);
	std::vector<Vector4> temp(size_t(width_dst) * height_src);
	Mipmap::downsample(width_src, height_src, width_dst, height_dst, reinterpret_cast<const Vector4 *>(src), reinterpret_cast<Vector4 *>(dst), temp.data());
	cpu_config.mipmap_filter = saved;
}

// Primitive shapes of the Mitsuba loader, verbatim (Src/Util/Geometry.cpp). shape: 0 rectangle, 1 cube, 2 disk,
// 3 cylinder (p0, p1, radius), 4 sphere (subdivisions in `detail`). transform: 16 floats, row-major Matrix4 cells.
// Returns the triangle count; dst (may be NULL) receives count * 24 floats.
#include "Util/Geometry.h"
extern "C" int ref_geometry_shape(int shape, const float * transform16, const float * p0, const float * p1, float radius, int detail, float * dst, int dst_triangles) {
	Matrix4 transform;
	memcpy(transform.cells, transform16, 16 * sizeof(float));
	Array<Triangle> triangles;
	switch (shape) {
		case 0: triangles = Geometry::rectangle(transform); break;
		case 1: triangles = Geometry::cube(transform); break;
		case 2: triangles = detail > 0 ? Geometry::disk(transform, detail) : Geometry::disk(transform); break;
		case 3: triangles = detail > 0 ? Geometry::cylinder(transform, Vector3(p0[0], p0[1], p0[2]), Vector3(p1[0], p1[1], p1[2]), radius, detail)
		                               : Geometry::cylinder(transform, Vector3(p0[0], p0[1], p0[2]), Vector3(p1[0], p1[1], p1[2]), radius); break;
		default: triangles = detail >= 0 ? Geometry::sphere(transform, detail) : Geometry::sphere(transform); break;
	}
	int count = int(triangles.size());
	if (dst && dst_triangles >= count) memcpy(dst, triangles.data(), size_t(count) * sizeof(Triangle));
	return count;
}

// Camera, verbatim (Src/Renderer/Camera.cpp). No keys are ever down here; the stubs below stand in for the
// SDL-backed Input namespace. out: bottom_left_corner_rotated(3) x_axis_rotated(3) y_axis_rotated(3)
// pixel_spread_angle(1) projection(16) view_projection(16) view_projection_prev(16) = 58 floats, after
// resize(width, height) and `updates` calls of update(0).
#include "Renderer/Camera.h"
#include "Input.h"
bool Input::is_key_down   (SDL_Scancode) { return false; }
bool Input::is_key_pressed(SDL_Scancode) { return false; }
extern "C" void ref_camera_state(float fov, int width, int height, const float * position3, const float * rotation4, int updates, float * out58) {
	MuteStdout mute;
	Camera camera(fov);
	camera.resize(width, height);
	camera.position = Vector3(position3[0], position3[1], position3[2]);
	camera.rotation = Quaternion(rotation4[0], rotation4[1], rotation4[2], rotation4[3]);
	memset(camera.view_projection.cells, 0, sizeof(camera.view_projection.cells));
	for (int i = 0; i < updates; i++) camera.update(0.0f);
	float * o = out58;
	for (const Vector3 & v : { camera.bottom_left_corner_rotated, camera.x_axis_rotated, camera.y_axis_rotated }) { *o++ = v.x; *o++ = v.y; *o++ = v.z; }
	*o++ = camera.pixel_spread_angle;
	for (const Matrix4 * m : { &camera.projection, &camera.view_projection, &camera.view_projection_prev }) { memcpy(o, m->cells, 64); o += 16; }
}

// Instance transform as Mesh::update computes it (Src/Renderer/Mesh.cpp:16-33; that file cannot be compiled
// alone -- it includes the whole Scene -- so its three expressions are spelled out here over the reference's
// own Matrix4 / Quaternion / AABB code). out: transform(16) transform_inv(16) aabb min(3) max(3) = 38 floats.
extern "C" void ref_mesh_transform(const float * position3, const float * rotation4, float scale, const float * aabb6, float * out38) {
	Vector3    position(position3[0], position3[1], position3[2]);
	Quaternion rotation(rotation4[0], rotation4[1], rotation4[2], rotation4[3]);
	Matrix4 transform     = Matrix4::create_translation(position) * Matrix4::create_rotation(rotation) * Matrix4::create_scale(scale);
	Matrix4 transform_inv = Matrix4::create_scale(1.0f / scale) * Matrix4::create_rotation(Quaternion::conjugate(rotation)) * Matrix4::create_translation(-position);
	AABB untransformed; untransformed.min = Vector3(aabb6[0], aabb6[1], aabb6[2]); untransformed.max = Vector3(aabb6[3], aabb6[4], aabb6[5]);
	AABB aabb = AABB::transform(untransformed, transform);
	aabb.fix_if_needed();
	memcpy(out38, transform.cells, 64); memcpy(out38 + 16, transform_inv.cells, 64);
	out38[32] = aabb.min.x; out38[33] = aabb.min.y; out38[34] = aabb.min.z; out38[35] = aabb.max.x; out38[36] = aabb.max.y; out38[37] = aabb.max.z;
}

// Medium parameterisation, verbatim header (Src/Renderer/Medium.h:16-37): (sigma_a, sigma_s, g) -> (C, mfp) -> sigmas again.
#include "Renderer/Medium.h"
extern "C" void ref_medium_round_trip(const float * sigma_a3, const float * sigma_s3, float g, float * out12) {
	Medium medium;
	medium.g = g;
	medium.from_sigmas(Vector3(sigma_a3[0], sigma_a3[1], sigma_a3[2]), Vector3(sigma_s3[0], sigma_s3[1], sigma_s3[2]));
	Vector3 a, s;
	medium.to_sigmas(a, s);
	const Vector3 * v[4] = { &medium.C, &medium.mfp, &a, &s };
	for (int i = 0; i < 4; i++) { out12[3 * i] = v[i]->x; out12[3 * i + 1] = v[i]->y; out12[3 * i + 2] = v[i]->z; }
}
