// oracle_pathtrace.cpp -- CPU restatement of the wavefront path tracer. TEST INFRASTRUCTURE ONLY.
//
// Follows CUDA/Pathtracer.cu:122-796 kernel by kernel (generate, sort, shade_material<BSDF>,
// next_event_estimation, the shadow-miss lambda, accumulate), CUDA/Camera.h:20-62,
// CUDA/BSDF.h:8-525, CUDA/AOV.h:15-46 and the launch loop of Pathtracer::render
// (Src/Renderer/Integrators/Pathtracer.cpp:738-855).  Queues are processed in index order;
// the CUDA version appends with atomicAdd, so only the ORDER inside a queue differs, and no
// result depends on it (one live path per pixel, AOV adds are per pixel).
#include "oracle.h"
#include "oracle_shading.h"

#include <cstdio>
#include <cstdlib>
#include <vector>
#include <omp.h>

void oracle_trace_one(const oracle_scene & s, float3 origin, float3 direction, uint32_t * hit4, oracle_trace_stats * stats);
bool oracle_trace_shadow_one(const oracle_scene & s, float3 origin, float3 direction, float max_distance, oracle_trace_stats * stats);
void oracle_svgf_taa(const oracle_scene & s, oracle_frame & f, int sample_index);

namespace {

constexpr unsigned FLAG_ALLOW_NEE     = 1u << 31; // Pathtracer.cu:27-30
constexpr unsigned FLAG_INSIDE_MEDIUM = 1u << 30;
constexpr unsigned FLAGS_ALL = FLAG_ALLOW_NEE | FLAG_INSIDE_MEDIUM;

struct RayHit { float t, u, v; int mesh_id, triangle_id; };

inline RayHit unpack_hit(const uint32_t * h) { // Buffers.h:34-48
	RayHit r;
	r.mesh_id = int(h[0]); r.triangle_id = int(h[1]);
	r.t = uint_as_float(h[2]);
	r.u = float(h[3] & 0xffff) / 65535.0f;
	r.v = float(h[3] >> 16)    / 65535.0f;
	return r;
}

// SoA queues (Pathtracer.cu:32-66), one entry per ray
struct TraceRay {
	float3 origin, direction;
	uint32_t hit[4];
	float cone_angle, cone_width;
	int medium;
	unsigned pixel_index_and_flags;
	float3 throughput;
	float last_pdf;
};
struct MaterialRay {
	float3 direction;
	uint32_t hit[4];
	float cone_angle, cone_width;
	int medium;
	unsigned pixel_index_and_flags;
	float3 throughput;
};
struct ShadowRay {
	float3 origin, direction;
	float max_distance;
	float3 illumination;
	int pixel_index;
};

struct Context {
	const oracle_scene & s;
	oracle_frame & f;
	Context(const oracle_scene & s, oracle_frame & f) : s(s), f(f) { }

	// AOV.h:15-33
	float4 aov_get(int aov, int pixel) const { const float * p = f.framebuffer[aov] + size_t(pixel) * 4; return make_float4(p[0], p[1], p[2], p[3]); }
	void aov_set(int aov, int pixel, float4 v) { if (f.framebuffer[aov]) { float * p = f.framebuffer[aov] + size_t(pixel) * 4; p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; } }
	void aov_add(int aov, int pixel, float4 v) { if (f.framebuffer[aov]) { float * p = f.framebuffer[aov] + size_t(pixel) * 4; p[0] += v.x; p[1] += v.y; p[2] += v.z; p[3] += v.w; } }

	float2 random(int dim, int pixel_index, int bounce, int sample_index) const { return oracle_random_sample(s, dim, uint32_t(pixel_index), uint32_t(bounce), uint32_t(sample_index)); }
};

// ---- triangles (Triangle.h) -----------------------------------------------------------------------
struct TrianglePosNorTex {
	float3 position_0, position_edge_1, position_edge_2;
	float3 normal_0, normal_edge_1, normal_edge_2;
	float2 tex_coord_0, tex_coord_edge_1, tex_coord_edge_2;
};
inline TrianglePosNorTex triangle_get(const oracle_scene & s, int index) {
	const float * t = s.triangles + size_t(index) * 24;
	TrianglePosNorTex r;
	r.position_0 = make_float3(t[0], t[1], t[2]); r.position_edge_1 = make_float3(t[3], t[4], t[5]); r.position_edge_2 = make_float3(t[6], t[7], t[8]);
	r.normal_0 = make_float3(t[9], t[10], t[11]); r.normal_edge_1 = make_float3(t[12], t[13], t[14]); r.normal_edge_2 = make_float3(t[15], t[16], t[17]);
	r.tex_coord_0 = make_float2(t[18], t[19]); r.tex_coord_edge_1 = make_float2(t[20], t[21]); r.tex_coord_edge_2 = make_float2(t[22], t[23]);
	return r;
}
inline float3 barycentric(float u, float v, float3 base, float3 e1, float3 e2) { return base + u * e1 + v * e2; } // Util.h:210-213
inline float2 barycentric(float u, float v, float2 base, float2 e1, float2 e2) { return base + u * e1 + v * e2; }

inline float triangle_get_lod(float double_area_world_inv, float2 te1, float2 te2) { // Triangle.h:105-115
	float area_texel = fabsf(te1.x * te2.y - te2.x * te1.y);
	return sqrtf(area_texel * double_area_world_inv);
}
inline float triangle_get_curvature(float3 pe1, float3 pe2, float3 ne1, float3 ne2) { // Triangle.h:117-131
	float3 ne0 = ne1 - ne2;
	float3 pe0 = pe1 - pe2;
	float k_01 = dot(ne1, pe1) / dot(pe1, pe1);
	float k_02 = dot(ne2, pe2) / dot(pe2, pe2);
	float k_12 = dot(ne0, pe0) / dot(pe0, pe0);
	return (k_01 + k_02 + k_12) * (1.0f / 3.0f);
}

// ---- Mesh.h ----------------------------------------------------------------------------------------
inline float3 m_position(const float * m, float3 p) {
	return make_float3(
		m[0] * p.x + m[1] * p.y + m[ 2] * p.z + m[ 3],
		m[4] * p.x + m[5] * p.y + m[ 6] * p.z + m[ 7],
		m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
inline float3 m_direction(const float * m, float3 d) {
	return make_float3(
		m[0] * d.x + m[1] * d.y + m[ 2] * d.z,
		m[4] * d.x + m[5] * d.y + m[ 6] * d.z,
		m[8] * d.x + m[9] * d.y + m[10] * d.z);
}
inline const float * mesh_transform(const oracle_scene & s, int mesh_id)      { return s.mesh_transforms      + size_t(mesh_id) * 12; }
inline const float * mesh_transform_prev(const oracle_scene & s, int mesh_id) { return s.mesh_transforms_prev + size_t(mesh_id) * 12; }
inline float mesh_get_scale(const oracle_scene & s, int mesh_id) { const float * m = mesh_transform(s, mesh_id); return length(make_float3(m[0], m[1], m[2])); }

// ---- RayCone.h ----------------------------------------------------------------------------------------
struct TextureLOD { float2 gradient_1, gradient_2; float lod; };

inline float3 material_get_albedo(const oracle_scene & s, float3 diffuse, int texture_id, float u, float v) {
	if (texture_id == RT_INVALID) return diffuse;
	return diffuse * make_float3(texture_get(s.textures[texture_id], u, v));
}
inline float3 sample_albedo(const oracle_scene & s, int bounce, float3 diffuse, int texture_id, float2 tex_coord, const TextureLOD & lod) { // RayCone.h:19-29
	if (s.config.enable_mipmapping && texture_id != RT_INVALID) {
		const oracle_texture & tex = s.textures[texture_id];
		if (bounce == 0) {
			return diffuse * make_float3(texture_get_grad(tex, tex_coord.x, tex_coord.y, lod.gradient_1, lod.gradient_2));
		} else {
			int lod_width  = tex.lod_width  > 0 ? tex.lod_width  : tex.width;
			int lod_height = tex.lod_height > 0 ? tex.lod_height : tex.height;
			float lod_bias = 0.5f * log2f(float(lod_width * lod_height)); // Integrator.cpp:95
			return diffuse * make_float3(texture_get_lod(tex, tex_coord.x, tex_coord.y, lod.lod + lod_bias));
		}
	}
	return material_get_albedo(s, diffuse, texture_id, tex_coord.x, tex_coord.y);
}
inline void ray_cone_get_ellipse_axes(float3 ray_direction, float3 geometric_normal, float cone_width, float3 & axis_1, float3 & axis_2) { // RayCone.h:32-44
	float3 h_1 = ray_direction - dot(geometric_normal, ray_direction) * geometric_normal;
	float3 h_2 = cross(geometric_normal, h_1);
	axis_1 = cone_width / fmaxf(0.0001f, length(h_1 - dot(ray_direction, h_1) * ray_direction)) * h_1;
	axis_2 = cone_width / fmaxf(0.0001f, length(h_2 - dot(ray_direction, h_2) * ray_direction)) * h_2;
}
inline float2 ray_cone_ellipse_axis_to_gradient(const TrianglePosNorTex & tri, float double_area_inv, float3 geometric_normal, float3 hit_point, float2 hit_tex_coord, float3 ellipse_axis) { // RayCone.h:47-61
	float3 e_p = hit_point + ellipse_axis - tri.position_0;
	float u = dot(geometric_normal, cross(e_p, tri.position_edge_2)) * double_area_inv;
	float v = dot(geometric_normal, cross(tri.position_edge_1, e_p)) * double_area_inv;
	return barycentric(u, v, tri.tex_coord_0, tri.tex_coord_edge_1, tri.tex_coord_edge_2) - hit_tex_coord;
}
inline float ray_cone_get_lod(float3 ray_direction, float3 geometric_normal, float cone_width) { return fabsf(cone_width / dot(ray_direction, geometric_normal)); }

// ---- BSDFs (BSDF.h) ----------------------------------------------------------------------------------------
struct BSDFBase {
	Context * c;
	int pixel_index, bounce, sample_index;
	float3 tangent, bitangent, normal, omega_i;
	const float * material; // 8 floats
};

struct BSDFDiffuse : BSDFBase { // BSDF.h:8-70
	static constexpr bool HAS_ALBEDO = true;
	float3 diffuse; int texture_id; float3 albedo;
	void init(bool /*entering*/) { diffuse = make_float3(material[0], material[1], material[2]); texture_id = float_as_int(material[3]); }
	void calc_albedo(float3 & throughput, float2 tex_coord, const TextureLOD & lod) {
		albedo = sample_albedo(c->s, bounce, diffuse, texture_id, tex_coord, lod);
		if (bounce == 0) c->aov_set(RT_AOV_ALBEDO, pixel_index, make_float4(albedo));
		if (!(c->s.config.enable_svgf && bounce == 0)) throughput *= albedo;
	}
	bool eval(float3 /*to_light*/, float cos_theta_o, float3 & bsdf, float & pdf) const {
		if (cos_theta_o <= 0.0f) return false;
		bsdf = make_float3(cos_theta_o * O_ONE_OVER_PI);
		pdf  = cos_theta_o * O_ONE_OVER_PI;
		return pdf_is_valid(pdf);
	}
	bool sample(float3 & /*throughput*/, int & /*medium_id*/, float3 & direction_out, float & pdf) const {
		float2 r = c->random(DIM_BSDF_0, pixel_index, bounce, sample_index);
		float3 omega_o = sample_cosine_weighted_direction(r.x, r.y);
		direction_out = local_to_world(omega_o, tangent, bitangent, normal);
		pdf = omega_o.z * O_ONE_OVER_PI;
		return pdf_is_valid(pdf);
	}
	bool has_texture() const { return texture_id != RT_INVALID; }
	bool allow_nee() const { return true; }
};

struct BSDFPlastic : BSDFBase { // BSDF.h:72-190
	static constexpr bool HAS_ALBEDO = true;
	static constexpr float IOR = 1.5f;
	static constexpr float ETA = 1.0f / IOR;
	float3 diffuse; int texture_id; float linear_roughness; float3 albedo;
	void init(bool) { diffuse = make_float3(material[0], material[1], material[2]); texture_id = float_as_int(material[3]); linear_roughness = material[4]; }
	void calc_albedo(float3 & /*throughput*/, float2 tex_coord, const TextureLOD & lod) {
		albedo = sample_albedo(c->s, bounce, diffuse, texture_id, tex_coord, lod);
		if (bounce == 0) c->aov_set(RT_AOV_ALBEDO, pixel_index, make_float4(albedo));
	}
	float3 diffuse_lobe(float F_i, float F_o, float cos_o) const {
		float F_avg = average_fresnel(IOR);
		float internal_scattering_factor = 1.0f - (1.0f - F_avg) * square(ETA);
		return ETA * ETA * (1.0f - F_i) * (1.0f - F_o) * albedo * O_ONE_OVER_PI / (1.0f - albedo * internal_scattering_factor) * cos_o;
	}
	bool eval(float3 to_light, float cos_theta_o, float3 & bsdf, float & pdf) const {
		if (cos_theta_o <= 0.0f) return false;
		float3 omega_o = world_to_local(to_light, tangent, bitangent, normal);
		float3 omega_m = normalize(omega_i + omega_o);
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		float F  = fresnel_dielectric(dot(omega_i, omega_m), ETA);
		float D  = ggx_D(omega_m, ax, ay);
		float G1 = ggx_G1(omega_i, ax, ay);
		float G2 = ggx_G2(omega_o, omega_i, omega_m, ax, ay);
		float3 brdf_specular = make_float3(F * G2 * D / (4.0f * omega_i.z));
		float F_i = fresnel_dielectric(omega_i.z, ETA);
		float F_o = fresnel_dielectric(omega_o.z, ETA);
		float3 brdf_diffuse = diffuse_lobe(F_i, F_o, omega_o.z);
		float pdf_specular = G1 * D / (4.0f * omega_i.z);
		float pdf_diffuse  = omega_o.z * O_ONE_OVER_PI;
		pdf  = lerp_ref(pdf_diffuse, pdf_specular, F_i);
		bsdf = brdf_specular + brdf_diffuse;
		return pdf_is_valid(pdf);
	}
	bool sample(float3 & throughput, int & /*medium_id*/, float3 & direction_out, float & pdf) const {
		float  rand_fresnel = c->random(DIM_BSDF_0, pixel_index, bounce, sample_index).x;
		float2 rand_brdf    = c->random(DIM_BSDF_1, pixel_index, bounce, sample_index);
		float F_i = fresnel_dielectric(omega_i.z, ETA);
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		float3 omega_m, omega_o;
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
		float G1 = ggx_G1(omega_i, ax, ay);
		float G2 = ggx_G2(omega_o, omega_i, omega_m, ax, ay);
		float3 brdf_specular = make_float3(F * G2 * D / (4.0f * omega_i.z));
		float F_o = fresnel_dielectric(omega_o.z, ETA);
		float3 brdf_diffuse = diffuse_lobe(F_i, F_o, omega_o.z);
		float pdf_specular = G1 * D / (4.0f * omega_i.z);
		float pdf_diffuse  = omega_o.z * O_ONE_OVER_PI;
		pdf = lerp_ref(pdf_diffuse, pdf_specular, F_i);
		throughput *= (brdf_specular + brdf_diffuse) / pdf;
		direction_out = local_to_world(omega_o, tangent, bitangent, normal);
		return pdf_is_valid(pdf);
	}
	bool has_texture() const { return texture_id != RT_INVALID; }
	bool allow_nee() const { return true; }
};

struct BSDFDielectric : BSDFBase { // BSDF.h:192-403
	static constexpr bool HAS_ALBEDO = false;
	int medium_id_material; float ior, linear_roughness, eta;
	void init(bool entering_material) {
		medium_id_material = float_as_int(material[0]); ior = material[1]; linear_roughness = material[2];
		eta = entering_material ? 1.0f / ior : ior;
	}
	void calc_albedo(float3 &, float2, const TextureLOD &) { }

	struct Lobes { float bsdf_single, bsdf_multi, pdf_single, pdf_multi; };
	Lobes lobes(bool reflected, bool entering_material, float3 omega_o, float3 omega_m, float F, float E_i, float ratio, float E_avg_enter, float E_avg_leave) const {
		const oracle_scene & s = c->s;
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
			float E_o   = dielectric_directional_albedo(s, ior, linear_roughness, omega_o.z, entering_material);
			float E_avg = entering_material ? E_avg_enter : E_avg_leave;
			l.bsdf_multi = (1.0f - ratio) * fabsf(omega_o.z) * kulla_conty_multiscatter_lobe(E_i, E_o, E_avg);
			l.pdf_multi  = (1.0f - ratio) * fabsf(omega_o.z) * O_ONE_OVER_PI;
		} else {
			l.bsdf_single = (1.0f - F) * G2 * D * i_dot_m * o_dot_m / (omega_i.z * square(eta * i_dot_m + o_dot_m) * square(eta));
			l.pdf_single  = (1.0f - F) * G1 * D * i_dot_m * o_dot_m / (omega_i.z * square(eta * i_dot_m + o_dot_m));
			float E_o   = dielectric_directional_albedo(s, ior, linear_roughness, omega_o.z, !entering_material);
			float E_avg = entering_material ? E_avg_leave : E_avg_enter; // inverted on purpose (BSDF.h:281)
			l.bsdf_multi = ratio * fabsf(omega_o.z) * kulla_conty_multiscatter_lobe(E_i, E_o, E_avg);
			l.pdf_multi  = ratio * fabsf(omega_o.z) * O_ONE_OVER_PI;
		}
		return l;
	}
	void common(bool & entering_material, float & E_i, float & ratio, float & E_avg_enter, float & E_avg_leave) const {
		const oracle_scene & s = c->s;
		entering_material = eta < 1.0f;
		E_i = dielectric_directional_albedo(s, ior, linear_roughness, omega_i.z, entering_material);
		float F_avg = average_fresnel(ior);
		if (!entering_material) F_avg = 1.0f - (1.0f - F_avg) / square(ior);
		E_avg_enter = dielectric_albedo(s, ior, linear_roughness, true);
		E_avg_leave = dielectric_albedo(s, ior, linear_roughness, false);
		float x = kulla_conty_dielectric_reciprocity_factor(E_avg_enter, E_avg_leave);
		ratio = (entering_material ? x : (1.0f - x)) * (1.0f - F_avg);
	}
	bool eval(float3 to_light, float /*cos_theta_o*/, float3 & bsdf, float & pdf) const {
		float3 omega_o = world_to_local(to_light, tangent, bitangent, normal);
		bool reflected = omega_o.z >= 0.0f;
		float3 omega_m = reflected ? normalize(omega_i + omega_o) : normalize(eta * omega_i + omega_o);
		omega_m *= sign_of(omega_m.z);
		float F = fresnel_dielectric(abs_dot(omega_i, omega_m), eta);
		bool entering_material; float E_i, ratio, E_avg_enter, E_avg_leave;
		common(entering_material, E_i, ratio, E_avg_enter, E_avg_leave);
		Lobes l = lobes(reflected, entering_material, omega_o, omega_m, F, E_i, ratio, E_avg_enter, E_avg_leave);
		bsdf = make_float3(l.bsdf_single + l.bsdf_multi);
		pdf = lerp_ref(l.pdf_multi, l.pdf_single, E_i);
		return pdf_is_valid(pdf);
	}
	bool sample(float3 & throughput, int & medium_id, float3 & direction_out, float & pdf) const {
		float2 r0 = c->random(DIM_BSDF_0, pixel_index, bounce, sample_index);
		float2 r1 = c->random(DIM_BSDF_1, pixel_index, bounce, sample_index);
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		bool entering_material; float E_i, ratio, E_avg_enter, E_avg_leave;
		common(entering_material, E_i, ratio, E_avg_enter, E_avg_leave);

		float F; bool reflected; float3 omega_m, omega_o;
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

		Lobes l = lobes(reflected, entering_material, omega_o, omega_m, F, E_i, ratio, E_avg_enter, E_avg_leave);
		if (!reflected) medium_id = entering_material ? medium_id_material : RT_INVALID;
		pdf = lerp_ref(l.pdf_multi, l.pdf_single, E_i);
		throughput *= (l.bsdf_single + l.bsdf_multi) / pdf;
		direction_out = local_to_world(omega_o, tangent, bitangent, normal);
		return pdf_is_valid(pdf);
	}
	bool has_texture() const { return false; }
	bool allow_nee() const { return linear_roughness >= ROUGHNESS_CUTOFF; }
};

struct BSDFConductor : BSDFBase { // BSDF.h:405-525
	static constexpr bool HAS_ALBEDO = false;
	float3 eta3, k3; float linear_roughness;
	void init(bool) { eta3 = make_float3(material[0], material[1], material[2]); linear_roughness = material[3]; k3 = make_float3(material[4], material[5], material[6]); }
	void calc_albedo(float3 &, float2, const TextureLOD &) { }
	void lobes(float3 omega_o, float3 omega_m, float o_dot_m, float E_i, float3 & brdf, float & pdf) const {
		const oracle_scene & s = c->s;
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		float3 F  = fresnel_conductor(o_dot_m, eta3, k3);
		float  D  = ggx_D(omega_m, ax, ay);
		float  G1 = ggx_G1(omega_i, ax, ay);
		float  G2 = ggx_G2(omega_o, omega_i, omega_m, ax, ay);
		float3 brdf_single = F * G2 * D / (4.0f * omega_i.z);
		float  pdf_single  =     G1 * D / (4.0f * omega_i.z);
		float E_o   = conductor_directional_albedo(s, linear_roughness, omega_o.z);
		float E_avg = conductor_albedo(s, linear_roughness);
		float3 F_avg = average_fresnel(eta3, k3);
		float3 F_ms  = fresnel_multiscatter(F_avg, E_avg);
		float3 brdf_multi = F_ms * kulla_conty_multiscatter_lobe(E_i, E_o, E_avg) * omega_o.z;
		float  pdf_multi  = omega_o.z * O_ONE_OVER_PI;
		brdf = brdf_single + brdf_multi;
		pdf = lerp_ref(pdf_multi, pdf_single, E_i);
	}
	bool eval(float3 to_light, float cos_theta_o, float3 & bsdf, float & pdf) const {
		if (cos_theta_o <= 0.0f) return false;
		float3 omega_o = world_to_local(to_light, tangent, bitangent, normal);
		float3 omega_m = normalize(omega_o + omega_i);
		float o_dot_m = dot(omega_o, omega_m);
		if (o_dot_m <= 0.0f) return false;
		float E_i = conductor_directional_albedo(c->s, linear_roughness, omega_i.z);
		lobes(omega_o, omega_m, o_dot_m, E_i, bsdf, pdf);
		return pdf_is_valid(pdf);
	}
	bool sample(float3 & throughput, int & /*medium_id*/, float3 & direction_out, float & pdf) const {
		float2 r0 = c->random(DIM_BSDF_0, pixel_index, bounce, sample_index);
		float2 r1 = c->random(DIM_BSDF_1, pixel_index, bounce, sample_index);
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		float E_i = conductor_directional_albedo(c->s, linear_roughness, omega_i.z);
		float3 omega_m, omega_o;
		if (r0.x < E_i) {
			omega_m = sample_visible_normals_ggx(omega_i, ax, ay, r1.x, r1.y);
			omega_o = reflect_direction(omega_i, omega_m);
		} else {
			omega_o = sample_cosine_weighted_direction(r1.x, r1.y);
			omega_m = normalize(omega_i + omega_o);
		}
		float o_dot_m = dot(omega_o, omega_m);
		if (o_dot_m <= 0.0f || omega_o.z < 0.0f) return false;
		float3 brdf;
		lobes(omega_o, omega_m, o_dot_m, E_i, brdf, pdf);
		throughput *= brdf / pdf;
		direction_out = local_to_world(omega_o, tangent, bitangent, normal);
		return pdf_is_valid(pdf);
	}
	bool has_texture() const { return false; }
	bool allow_nee() const { return linear_roughness >= ROUGHNESS_CUTOFF; }
};

// ---- SVGF g-buffers (SVGF.h:61-84), filled during shading ---------------------------------------------
inline float2 oct_encode_normal(float3 n) { // Util.h:238-248
	n /= (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
	if (n.z < 0.0f) {
		n.x = (1.0f - fabsf(n.y)) * (n.x >= 0.0f ? +1.0f : -1.0f);
		n.y = (1.0f - fabsf(n.x)) * (n.y >= 0.0f ? +1.0f : -1.0f);
	}
	return make_float2(0.5f + 0.5f * n.x, 0.5f + 0.5f * n.y);
}
inline float4 mat4_mul(const float * m, float4 v) {
	return make_float4(
		m[ 0] * v.x + m[ 1] * v.y + m[ 2] * v.z + m[ 3] * v.w,
		m[ 4] * v.x + m[ 5] * v.y + m[ 6] * v.z + m[ 7] * v.w,
		m[ 8] * v.x + m[ 9] * v.y + m[10] * v.z + m[11] * v.w,
		m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w);
}
inline void svgf_set_gbuffers(Context & c, int x, int y, const RayHit & hit, float3 hit_point, float3 normal, float3 hit_point_prev) {
	const oracle_scene & s = c.s;
	float4 u_curr = mat4_mul(s.view_projection,      make_float4(hit_point.x, hit_point.y, hit_point.z, 1.0f));
	float4 u_prev = mat4_mul(s.view_projection_prev, make_float4(hit_point_prev.x, hit_point_prev.y, hit_point_prev.z, 1.0f));
	float depth      = u_curr.z;
	float depth_prev = u_prev.z;
	// stored as NDC in [-1,1]; kernel_svgf_reproject / kernel_taa map it to [0,1] (SVGF.h:77-81)
	float2 screen_prev = make_float2(u_prev.x / u_prev.w, u_prev.y / u_prev.w);
	int idx = x + y * s.screen_pitch;
	float2 oct = oct_encode_normal(normal);
	float * nd = c.f.gbuffer_normal_and_depth + size_t(idx) * 4;
	nd[0] = oct.x; nd[1] = oct.y; nd[2] = depth; nd[3] = depth_prev;
	c.f.gbuffer_mesh_id_and_triangle_id[2 * idx + 0] = hit.mesh_id;
	c.f.gbuffer_mesh_id_and_triangle_id[2 * idx + 1] = hit.triangle_id;
	c.f.gbuffer_screen_position_prev[2 * idx + 0] = screen_prev.x;
	c.f.gbuffer_screen_position_prev[2 * idx + 1] = screen_prev.y;
}

// ---- the wavefront state ------------------------------------------------------------------------------
struct Wavefront {
	std::vector<TraceRay>    trace[2];
	std::vector<MaterialRay> material[4]; // diffuse, plastic, dielectric, conductor
	std::vector<ShadowRay>   shadow;
};

// The queue-processing kernels (sort, shade) run over their input in contiguous chunks, one per thread, each
// appending to a wavefront of its own; the pieces are then joined in chunk order, which IS the order a single
// thread would have produced. Paths of different queue entries never touch the same pixel within one kernel
// (one path per pixel and sample), so the AOV updates need no ordering either. Purely a speed-up of the checker:
// a 1080p sample has ~7 M queue entries to shade, and the GPU box has 100+ cores idling next to the test.
static int g_oracle_threads = 1;
template<typename Body>
static void run_in_chunks(size_t count, Wavefront & w, Body && body) {
	int chunks = count < 8192 ? 1 : g_oracle_threads;
	if (chunks <= 1) { body(size_t(0), count, w); return; }
	std::vector<Wavefront> pieces; pieces.resize(size_t(chunks));
	#pragma omp parallel for schedule(static, 1) num_threads(chunks)
	for (int t = 0; t < chunks; t++) body(count * size_t(t) / size_t(chunks), count * size_t(t + 1) / size_t(chunks), pieces[size_t(t)]);
	for (Wavefront & piece : pieces) {
		for (int k = 0; k < 2; k++) w.trace[k].insert(w.trace[k].end(), piece.trace[k].begin(), piece.trace[k].end());
		for (int k = 0; k < 4; k++) w.material[k].insert(w.material[k].end(), piece.material[k].begin(), piece.material[k].end());
		w.shadow.insert(w.shadow.end(), piece.shadow.begin(), piece.shadow.end());
	}
}

// Camera.h:20-62
inline void camera_generate_ray(Context & c, int pixel_index, int sample_index, int x, int y, float3 & origin, float3 & direction) {
	const oracle_scene & s = c.s;
	const rt_camera & cam = s.camera;
	float2 rand_filter   = c.random(DIM_FILTER,   pixel_index, 0, sample_index);
	float2 rand_aperture = c.random(DIM_APERTURE, pixel_index, 0, sample_index);

	float2 jitter;
	if (s.config.enable_svgf) {
		const float taa_halton_x[4] = { 0.3f, 0.7f, 0.2f, 0.8f };
		const float taa_halton_y[4] = { 0.2f, 0.8f, 0.7f, 0.3f };
		jitter.x = taa_halton_x[sample_index & 3];
		jitter.y = taa_halton_y[sample_index & 3];
	} else {
		switch (s.config.reconstruction_filter) {
			case RT_FILTER_BOX:  jitter = rand_filter; break;
			case RT_FILTER_TENT: jitter.x = sample_tent(rand_filter.x); jitter.y = sample_tent(rand_filter.y); break;
			default: {
				float2 g = sample_gaussian(rand_filter.x, rand_filter.y);
				jitter.x = 0.5f + 0.5f * g.x;
				jitter.y = 0.5f + 0.5f * g.y;
			}
		}
	}
	float x_jittered = float(x) + jitter.x;
	float y_jittered = float(y) + jitter.y;

	float3 blc = make_float3(cam.bottom_left_corner[0], cam.bottom_left_corner[1], cam.bottom_left_corner[2]);
	float3 xa  = make_float3(cam.x_axis[0], cam.x_axis[1], cam.x_axis[2]);
	float3 ya  = make_float3(cam.y_axis[0], cam.y_axis[1], cam.y_axis[2]);

	float3 focal_point = cam.focal_distance * normalize(blc + x_jittered * xa + y_jittered * ya);
	float2 lens_point  = cam.aperture_radius * sample_disk(rand_aperture.x, rand_aperture.y);

	float3 offset = xa * lens_point.x + ya * lens_point.y;
	direction = normalize(focal_point - offset);
	origin = make_float3(cam.position[0], cam.position[1], cam.position[2]) + offset;
}

// Pathtracer.cu:199-218
inline bool russian_roulette(Context & c, int pixel_index, int bounce, int sample_index, float3 & throughput) {
	const rt_gpu_config & cfg = c.s.config;
	if (bounce == cfg.num_bounces - 1) return true;
	if (cfg.enable_russian_roulette && bounce > 0) {
		float3 t = throughput;
		if (cfg.enable_svgf) t *= make_float3(c.aov_get(RT_AOV_ALBEDO, pixel_index));
		float survival_probability = saturate(fmaxf(fmaxf(t.x, t.y), t.z));
		float r = c.random(DIM_RUSSIAN_ROULETTE, pixel_index, bounce, sample_index).x;
		if (r > survival_probability) return true;
		throughput /= survival_probability;
	}
	return false;
}

inline void add_radiance(Context & c, int bounce, int pixel_index, float3 illumination, float3 bounce0_value) {
	// The common AOV update pattern of kernel_sort (Pathtracer.cu:330-343,380-392)
	if (bounce == 0) {
		c.aov_set(RT_AOV_ALBEDO,          pixel_index, make_float4(1.0f));
		c.aov_set(RT_AOV_RADIANCE,        pixel_index, make_float4(bounce0_value));
		c.aov_set(RT_AOV_RADIANCE_DIRECT, pixel_index, make_float4(bounce0_value));
	} else if (bounce == 1) {
		c.aov_add(RT_AOV_RADIANCE,        pixel_index, make_float4(illumination));
		c.aov_add(RT_AOV_RADIANCE_DIRECT, pixel_index, make_float4(illumination));
	} else {
		c.aov_add(RT_AOV_RADIANCE,          pixel_index, make_float4(illumination));
		c.aov_add(RT_AOV_RADIANCE_INDIRECT, pixel_index, make_float4(illumination));
	}
}

// kernel_sort, Pathtracer.cu:220-463
void kernel_sort(Context & c, const std::vector<TraceRay> & in, size_t begin, size_t end, Wavefront & w, int bounce, int sample_index) {
	const oracle_scene & s = c.s;
	const rt_gpu_config & cfg = s.config;
	std::vector<TraceRay> & out = w.trace[(bounce + 1) & 1];

	for (size_t index = begin; index < end; index++) {
		const TraceRay & r = in[index];
		float3 ray_direction = r.direction;
		RayHit hit = unpack_hit(r.hit);

		float ray_cone_angle = 0.0f, ray_cone_width = 0.0f;
		if (bounce > 0 && cfg.enable_mipmapping) { ray_cone_angle = r.cone_angle; ray_cone_width = r.cone_width; }

		unsigned pixel_index_and_flags = r.pixel_index_and_flags;
		int pixel_index = int(pixel_index_and_flags & ~FLAGS_ALL);
		int x = pixel_index % s.screen_pitch;
		int y = pixel_index / s.screen_pitch;

		bool allow_nee     = pixel_index_and_flags & FLAG_ALLOW_NEE;
		bool inside_medium = pixel_index_and_flags & FLAG_INSIDE_MEDIUM;

		float3 throughput = bounce == 0 ? make_float3(1.0f) : r.throughput;

		int medium_id = RT_INVALID;
		if (inside_medium) {
			medium_id = r.medium;
			HomogeneousMedium medium = medium_as_homogeneous(s, medium_id);
			bool medium_can_scatter = (medium.sigma_s.x + medium.sigma_s.y + medium.sigma_s.z) > 0.0f;
			if (medium_can_scatter) {
				float2 rand_scatter = c.random(DIM_BSDF_0, pixel_index, bounce, sample_index);
				float2 rand_phase   = c.random(DIM_BSDF_1, pixel_index, bounce, sample_index);
				float3 sigma_t = medium.sigma_a + medium.sigma_s;

				float  throughput_sum = throughput.x + throughput.y + throughput.z;
				float3 wavelength_pdf = throughput / throughput_sum;

				float sigma_t_used;
				if      (rand_scatter.x * throughput_sum < throughput.x)                sigma_t_used = sigma_t.x;
				else if (rand_scatter.x * throughput_sum < throughput.x + throughput.y) sigma_t_used = sigma_t.y;
				else                                                                   sigma_t_used = sigma_t.z;

				float scatter_distance = sample_exp(sigma_t_used, rand_scatter.y);
				float3 transmittance = beer_lambert(sigma_t, fminf(scatter_distance, hit.t));

				if (scatter_distance < hit.t) {
					float3 pdf = wavelength_pdf * sigma_t * transmittance;
					throughput *= medium.sigma_s * transmittance / (pdf.x + pdf.y + pdf.z);
					if (russian_roulette(c, pixel_index, bounce, sample_index, throughput)) continue;

					float3 direction_out = sample_henyey_greenstein(-ray_direction, medium.g, rand_phase.x, rand_phase.y);
					float3 origin_out = r.origin + scatter_distance * ray_direction;

					TraceRay n = { };
					n.origin = origin_out; n.direction = direction_out; n.medium = medium_id;
					if (cfg.enable_mipmapping) {
						if (bounce == 0) { ray_cone_angle = s.camera.pixel_spread_angle; ray_cone_width = s.camera.pixel_spread_angle * scatter_distance; }
						n.cone_angle = ray_cone_angle; n.cone_width = ray_cone_width;
					}
					n.pixel_index_and_flags = unsigned(pixel_index) | FLAG_INSIDE_MEDIUM;
					n.throughput = throughput;
					out.push_back(n);
					continue;
				} else {
					float3 pdf = wavelength_pdf * transmittance;
					throughput *= transmittance / (pdf.x + pdf.y + pdf.z);
				}
			} else {
				throughput *= beer_lambert(medium.sigma_a, hit.t);
			}
		}

		if (hit.triangle_id == RT_INVALID) { // miss: sky
			float3 illumination = throughput * sample_sky(s, ray_direction);
			add_radiance(c, bounce, pixel_index, illumination, illumination);
			continue;
		}

		int material_id = s.mesh_material_ids[hit.mesh_id];
		int material_type = s.material_types[material_id];

		if (material_type == RT_MATERIAL_LIGHT) {
			TrianglePosNorTex tri = triangle_get(s, hit.triangle_id);
			float3 light_point = barycentric(hit.u, hit.v, tri.position_0, tri.position_edge_1, tri.position_edge_2);
			float3 light_point_prev = light_point;
			float3 light_geometric_normal = cross(tri.position_edge_1, tri.position_edge_2);

			const float * world = mesh_transform(s, hit.mesh_id);
			light_point = m_position(world, light_point);
			light_geometric_normal = normalize(m_direction(world, light_geometric_normal));

			if (bounce == 0 && cfg.enable_svgf) {
				light_point_prev = m_position(mesh_transform_prev(s, hit.mesh_id), light_point_prev);
				svgf_set_gbuffers(c, x, y, hit, light_point, light_geometric_normal, light_point_prev);
			}

			const float * lm = s.materials + size_t(material_id) * 8;
			float3 emission = make_float3(lm[0], lm[1], lm[2]);

			bool count_light = cfg.enable_next_event_estimation ? !allow_nee : true;
			if (count_light) {
				add_radiance(c, bounce, pixel_index, throughput * emission, emission);
				continue;
			}
			if (cfg.enable_multiple_importance_sampling) {
				float cos_theta_light = abs_dot(ray_direction, light_geometric_normal);
				float distance_to_light_squared = hit.t * hit.t;
				float brdf_pdf = r.last_pdf;
				float light_power = luminance(emission.x, emission.y, emission.z);
				float light_pdf = light_power * distance_to_light_squared / (cos_theta_light * s.lights_total_weight);
				if (!pdf_is_valid(light_pdf)) continue;
				float mis_weight = power_heuristic(brdf_pdf, light_pdf);
				float3 illumination = throughput * emission * mis_weight;
				c.aov_add(RT_AOV_RADIANCE, pixel_index, make_float4(illumination));
				if (bounce == 1) c.aov_add(RT_AOV_RADIANCE_DIRECT,   pixel_index, make_float4(illumination));
				else             c.aov_add(RT_AOV_RADIANCE_INDIRECT, pixel_index, make_float4(illumination));
			}
			continue;
		}

		if (russian_roulette(c, pixel_index, bounce, sample_index, throughput)) continue;

		MaterialRay m = { };
		m.direction = ray_direction;
		m.medium = medium_id;
		if (bounce > 0 && cfg.enable_mipmapping) { m.cone_angle = ray_cone_angle; m.cone_width = ray_cone_width; }
		memcpy(m.hit, r.hit, sizeof(m.hit));
		unsigned flags = unsigned(medium_id != RT_INVALID) << 30;
		m.pixel_index_and_flags = unsigned(pixel_index) | flags;
		m.throughput = throughput;
		switch (material_type) {
			case RT_MATERIAL_DIFFUSE:    w.material[0].push_back(m); break;
			case RT_MATERIAL_PLASTIC:    w.material[1].push_back(m); break;
			case RT_MATERIAL_DIELECTRIC: w.material[2].push_back(m); break;
			case RT_MATERIAL_CONDUCTOR:  w.material[3].push_back(m); break;
		}
	}
}

// Sampling.h:180-190
inline int sample_light(const oracle_scene & s, float u1, float u2, int & transform_id) {
	int light_mesh_id = binary_search(s.light_mesh_cumulative_probability, 0, s.light_mesh_count - 1, u1);
	transform_id = s.light_mesh_transform_indices[light_mesh_id];
	int first = s.light_mesh_triangle_span[2 * light_mesh_id], last = s.light_mesh_triangle_span[2 * light_mesh_id + 1];
	int light_triangle_id = binary_search(s.light_triangle_cumulative_probability, first, last, u2);
	return s.light_triangle_indices[light_triangle_id];
}

// next_event_estimation, Pathtracer.cu:465-555
template<typename BSDF>
void next_event_estimation(Context & c, Wavefront & w, int pixel_index, int bounce, int sample_index, const BSDF & bsdf, float3 hit_point, float3 normal, float3 geometric_normal, float3 throughput) {
	const oracle_scene & s = c.s;
	float2 rand_light    = c.random(DIM_NEE_LIGHT,    pixel_index, bounce, sample_index);
	float2 rand_triangle = c.random(DIM_NEE_TRIANGLE, pixel_index, bounce, sample_index);

	int light_mesh_id;
	int light_triangle_id = sample_light(s, rand_light.x, rand_light.y, light_mesh_id);
	float2 light_uv = sample_triangle(rand_triangle.x, rand_triangle.y);

	TrianglePosNorTex tri = triangle_get(s, light_triangle_id);
	float3 light_point = barycentric(light_uv.x, light_uv.y, tri.position_0, tri.position_edge_1, tri.position_edge_2);
	float3 light_geometric_normal = cross(tri.position_edge_1, tri.position_edge_2);

	const float * light_world = mesh_transform(s, light_mesh_id);
	light_point = m_position(light_world, light_point);
	light_geometric_normal = normalize(m_direction(light_world, light_geometric_normal));

	hit_point   = ray_origin_epsilon_offset(hit_point,   light_point - hit_point, geometric_normal);
	light_point = ray_origin_epsilon_offset(light_point, hit_point - light_point, light_geometric_normal);

	float3 to_light = light_point - hit_point;
	float distance_to_light = length(to_light);
	to_light /= distance_to_light;

	float cos_theta_light = abs_dot(to_light, light_geometric_normal);
	float cos_theta_hit = dot(to_light, normal);

	int light_material_id = s.mesh_material_ids[light_mesh_id];
	const float * lm = s.materials + size_t(light_material_id) * 8;
	float3 emission = make_float3(lm[0], lm[1], lm[2]);

	float3 bsdf_value; float bsdf_pdf;
	if (!bsdf.eval(to_light, cos_theta_hit, bsdf_value, bsdf_pdf)) return;

	float light_power = luminance(emission.x, emission.y, emission.z);
	float light_pdf   = light_power * square(distance_to_light) / (cos_theta_light * s.lights_total_weight);
	if (!pdf_is_valid(light_pdf)) return;

	float mis_weight = s.config.enable_multiple_importance_sampling ? power_heuristic(light_pdf, bsdf_pdf) : 1.0f;
	float3 illumination = throughput * bsdf_value * emission * mis_weight / light_pdf;
	// (medium transmittance towards the light is commented out in the reference, Pathtracer.cu:534-541)

	ShadowRay sr;
	sr.origin = hit_point; sr.direction = to_light; sr.max_distance = distance_to_light;
	sr.illumination = illumination; sr.pixel_index = pixel_index;
	w.shadow.push_back(sr);
}

// shade_material<BSDF>, Pathtracer.cu:557-757
template<typename BSDF>
void shade_material(Context & c, Wavefront & w, const std::vector<MaterialRay> & queue, size_t begin, size_t end, int bounce, int sample_index) {
	const oracle_scene & s = c.s;
	const rt_gpu_config & cfg = s.config;
	std::vector<TraceRay> & out = w.trace[(bounce + 1) & 1];

	for (size_t index = begin; index < end; index++) {
		const MaterialRay & r = queue[index];
		float3 ray_direction = r.direction;
		RayHit hit = unpack_hit(r.hit);

		unsigned pixel_index_and_flags = r.pixel_index_and_flags;
		int pixel_index = int(pixel_index_and_flags & ~FLAGS_ALL);
		bool inside_medium = pixel_index_and_flags & FLAG_INSIDE_MEDIUM;
		int medium_id = inside_medium ? r.medium : RT_INVALID;

		float3 throughput = bounce == 0 ? make_float3(1.0f) : r.throughput;

		TrianglePosNorTex tri = triangle_get(s, hit.triangle_id);
		float3 hit_point = barycentric(hit.u, hit.v, tri.position_0, tri.position_edge_1, tri.position_edge_2);
		float3 normal    = barycentric(hit.u, hit.v, tri.normal_0,   tri.normal_edge_1,   tri.normal_edge_2);
		float2 tex_coord = barycentric(hit.u, hit.v, tri.tex_coord_0, tri.tex_coord_edge_1, tri.tex_coord_edge_2);
		float3 hit_point_local = hit_point;

		const float * world = mesh_transform(s, hit.mesh_id);
		hit_point = m_position(world, hit_point);
		normal = normalize(m_direction(world, normal));

		float mesh_scale_inv = 1.0f / mesh_get_scale(s, hit.mesh_id);

		float cone_angle = 0.0f, cone_width = 0.0f, curvature = 0.0f;
		if (cfg.enable_mipmapping) {
			if (bounce == 0) { cone_angle = s.camera.pixel_spread_angle; cone_width = cone_angle * hit.t; }
			else             { cone_angle = r.cone_angle; cone_width = r.cone_width + cone_angle * hit.t; }
			curvature = triangle_get_curvature(tri.position_edge_1, tri.position_edge_2, tri.normal_edge_1, tri.normal_edge_2) * mesh_scale_inv;
		}

		tri.position_edge_1 = m_direction(world, tri.position_edge_1);
		tri.position_edge_2 = m_direction(world, tri.position_edge_2);

		float3 geometric_normal = cross(tri.position_edge_1, tri.position_edge_2);
		float triangle_double_area_inv = 1.0f / length(geometric_normal);
		geometric_normal *= triangle_double_area_inv;

		bool entering_material = dot(ray_direction, geometric_normal) < 0.0f;
		if (!entering_material) { normal = -normal; curvature = -curvature; }

		float3 tangent, bitangent;
		orthonormal_basis(normal, tangent, bitangent);
		float3 omega_i = world_to_local(-ray_direction, tangent, bitangent, normal);
		if (omega_i.z <= 0.0f) continue;

		int material_id = s.mesh_material_ids[hit.mesh_id];

		BSDF bsdf;
		bsdf.c = &c;
		bsdf.pixel_index = pixel_index; bsdf.bounce = bounce; bsdf.sample_index = sample_index;
		bsdf.tangent = tangent; bsdf.bitangent = bitangent; bsdf.normal = normal; bsdf.omega_i = omega_i;
		bsdf.material = s.materials + size_t(material_id) * 8;
		bsdf.init(entering_material);

		if (BSDF::HAS_ALBEDO) {
			TextureLOD lod = { };
			if (cfg.enable_mipmapping && bsdf.has_texture()) {
				if (bounce == 0) {
					float3 axis_1, axis_2;
					ray_cone_get_ellipse_axes(ray_direction, geometric_normal, cone_width, axis_1, axis_2);
					lod.gradient_1 = ray_cone_ellipse_axis_to_gradient(tri, triangle_double_area_inv, geometric_normal, hit_point, tex_coord, axis_1);
					lod.gradient_2 = ray_cone_ellipse_axis_to_gradient(tri, triangle_double_area_inv, geometric_normal, hit_point, tex_coord, axis_2);
				} else {
					float lod_triangle = triangle_get_lod(triangle_double_area_inv, tri.tex_coord_edge_1, tri.tex_coord_edge_2);
					float lod_ray_cone = ray_cone_get_lod(ray_direction, geometric_normal, cone_width);
					lod.lod = log2f(lod_triangle * lod_ray_cone);
				}
			}
			bsdf.calc_albedo(throughput, tex_coord, lod);
		} else if (bounce == 0) {
			c.aov_set(RT_AOV_ALBEDO, pixel_index, make_float4(1.0f));
		}

		if (bounce == 0) {
			c.aov_set(RT_AOV_NORMAL,   pixel_index, make_float4(normal));
			c.aov_set(RT_AOV_POSITION, pixel_index, make_float4(hit_point));
		}

		if (cfg.enable_mipmapping) cone_angle -= 2.0f * curvature * fabsf(cone_width) / dot(normal, ray_direction);

		if (bounce == 0 && cfg.enable_svgf) {
			float3 hit_point_prev = m_position(mesh_transform_prev(s, hit.mesh_id), hit_point_local);
			int x = pixel_index % s.screen_pitch, y = pixel_index / s.screen_pitch;
			svgf_set_gbuffers(c, x, y, hit, hit_point, normal, hit_point_prev);
		}

		if (cfg.enable_next_event_estimation && s.lights_total_weight > 0.0f && bsdf.allow_nee()) {
			next_event_estimation(c, w, pixel_index, bounce, sample_index, bsdf, hit_point, normal, geometric_normal, throughput);
		}

		float3 direction_out; float pdf;
		if (!bsdf.sample(throughput, medium_id, direction_out, pdf)) continue;

		float3 origin_out = ray_origin_epsilon_offset(hit_point, direction_out, geometric_normal);

		TraceRay n = { };
		n.origin = origin_out; n.direction = direction_out;
		n.medium = medium_id;
		if (cfg.enable_mipmapping) { n.cone_angle = cone_angle; n.cone_width = cone_width; }
		bool allow_nee = bsdf.allow_nee();
		unsigned flags = 0;
		if (allow_nee)               flags |= FLAG_ALLOW_NEE;
		if (medium_id != RT_INVALID) flags |= FLAG_INSIDE_MEDIUM;
		n.pixel_index_and_flags = unsigned(pixel_index) | flags;
		n.throughput = throughput;
		if (allow_nee) n.last_pdf = pdf;
		out.push_back(n);
	}
}

// kernel_accumulate, Pathtracer.cu:775-796 + aov_accumulate, AOV.h:35-46
void kernel_accumulate(Context & c, float frames_accumulated, int pixel_offset, int pixel_count) {
	const oracle_scene & s = c.s;
	for (int i = 0; i < pixel_count; i++) {
		int idx = i + pixel_offset;
		int x = idx % s.screen_width, y = idx / s.screen_width;
		int pixel_index = x + y * s.screen_pitch;

		auto accumulate = [&](int aov) -> float4 {
			if (!c.f.framebuffer[aov]) return make_float4(0.0f);
			float * fb = c.f.framebuffer[aov] + size_t(pixel_index) * 4;
			float * acc = c.f.accumulator[aov] + size_t(pixel_index) * 4;
			for (int k = 0; k < 4; k++) {
				if (frames_accumulated > 0.0f) acc[k] += (fb[k] - acc[k]) / frames_accumulated; // online average
				else                           acc[k] = fb[k];
			}
			return make_float4(acc[0], acc[1], acc[2], acc[3]);
		};
		float4 colour = accumulate(RT_AOV_RADIANCE);
		accumulate(RT_AOV_ALBEDO);
		accumulate(RT_AOV_NORMAL);
		accumulate(RT_AOV_POSITION);

		if (!std::isfinite(colour.x + colour.y + colour.z)) colour = make_float4(1000.0f, 0.0f, 1000.0f, 1.0f);
		float * out = c.f.final_image + size_t(pixel_index) * 4;
		out[0] = colour.x; out[1] = colour.y; out[2] = colour.z; out[3] = colour.w;
	}
}

} // namespace

extern "C" {

void oracle_generate(const oracle_scene * scene, int sample_index, int pixel_offset, int pixel_count,
                     float * ox, float * oy, float * oz, float * dx, float * dy, float * dz, uint32_t * pixel_index_and_flags) {
	oracle_frame dummy = { };
	Context c(*scene, dummy);
	for (int index = 0; index < pixel_count; index++) { // kernel_generate, Pathtracer.cu:122-139
		int index_offset = index + pixel_offset;
		int x = index_offset % scene->screen_width;
		int y = index_offset / scene->screen_width;
		int pixel_index = x + y * scene->screen_pitch;
		float3 o, d;
		camera_generate_ray(c, pixel_index, sample_index, x, y, o, d);
		ox[index] = o.x; oy[index] = o.y; oz[index] = o.z;
		dx[index] = d.x; dy[index] = d.y; dz[index] = d.z;
		pixel_index_and_flags[index] = uint32_t(pixel_index);
	}
}

void oracle_random(const oracle_scene * scene, int dimension, const uint32_t * pixel_indices, size_t count,
                   uint32_t bounce, uint32_t sample_index, float * out_xy) {
	for (size_t i = 0; i < count; i++) {
		float2 r = oracle_random_sample(*scene, dimension, pixel_indices[i], bounce, sample_index);
		out_xy[2 * i] = r.x; out_xy[2 * i + 1] = r.y;
	}
}

// Ambient-occlusion integrator: AO::render (Integrators/AO.cpp:148-200) with the kernels of
// CUDA/AO.cu -- generate (:48-63), trace, kernel_ambient_occlusion (:103-159), shadow trace whose
// miss lambda sets RADIANCE to 1 (:77-101), kernel_accumulate (:161-183: RADIANCE, NORMAL, POSITION).
void oracle_render_ao_sample(const oracle_scene * scene, oracle_frame * frame, int sample_index, float ao_radius,
                             int range_offset, int range_count, oracle_counters * counters, int threads) {
	const oracle_scene & s = *scene;
	if (threads <= 0) threads = oracle_default_threads();
	Context c(s, *frame);
	oracle_counters local; memset(&local, 0, sizeof(local));

	int pixels_left = range_count;
	int batch_size  = range_count < RT_BATCH_SIZE ? range_count : RT_BATCH_SIZE;
	while (pixels_left > 0) {
		int pixel_offset = range_offset + (range_count - pixels_left);
		int pixel_count  = batch_size < pixels_left ? batch_size : pixels_left;

		std::vector<TraceRay> rays;
		rays.resize(size_t(pixel_count));
		for (int index = 0; index < pixel_count; index++) { // kernel_generate
			int index_offset = index + pixel_offset;
			int x = index_offset % s.screen_width, y = index_offset / s.screen_width;
			int pixel_index = x + y * s.screen_pitch;
			camera_generate_ray(c, pixel_index, sample_index, x, y, rays[index].origin, rays[index].direction);
			rays[index].pixel_index_and_flags = unsigned(pixel_index);
		}
		local.trace[0] += pixel_count;

		#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
		for (long long i = 0; i < (long long)rays.size(); i++) oracle_trace_one(s, rays[i].origin, rays[i].direction, rays[i].hit, nullptr);

		std::vector<ShadowRay> shadow;
		for (const TraceRay & r : rays) { // kernel_ambient_occlusion
			RayHit hit = unpack_hit(r.hit);
			int pixel_index = int(r.pixel_index_and_flags);
			if (hit.triangle_id == RT_INVALID) continue;

			TrianglePosNorTex tri = triangle_get(s, hit.triangle_id);
			float3 geometric_normal = normalize(cross(tri.position_edge_1, tri.position_edge_2)); // object space, as in AO.cu:123
			float3 hit_point  = barycentric(hit.u, hit.v, tri.position_0, tri.position_edge_1, tri.position_edge_2);
			float3 hit_normal = barycentric(hit.u, hit.v, tri.normal_0,   tri.normal_edge_1,   tri.normal_edge_2);

			const float * world = mesh_transform(s, hit.mesh_id);
			hit_point  = m_position (world, hit_point);
			hit_normal = normalize(m_direction(world, hit_normal));
			if (dot(r.direction, hit_normal) > 0.0f) hit_normal = -hit_normal;

			c.aov_set(RT_AOV_NORMAL,   pixel_index, make_float4(hit_normal));
			c.aov_set(RT_AOV_POSITION, pixel_index, make_float4(hit_point));

			float3 tangent, bitangent;
			orthonormal_basis(hit_normal, tangent, bitangent);
			float2 rand_brdf = c.random(DIM_BSDF_0, pixel_index, 0, sample_index);
			float3 omega_o = sample_cosine_weighted_direction(rand_brdf.x, rand_brdf.y);
			float3 direction_out = local_to_world(omega_o, tangent, bitangent, hit_normal);
			float pdf = omega_o.z * O_ONE_OVER_PI;
			if (!pdf_is_valid(pdf)) continue;

			ShadowRay sr;
			sr.origin = ray_origin_epsilon_offset(hit_point, direction_out, geometric_normal);
			sr.direction = direction_out;
			sr.max_distance = ao_radius;
			sr.illumination = make_float3(1.0f);
			sr.pixel_index = pixel_index;
			shadow.push_back(sr);
		}
		local.shadow[0] += int(shadow.size());

		std::vector<unsigned char> occluded(shadow.size());
		#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
		for (long long i = 0; i < (long long)shadow.size(); i++) occluded[i] = oracle_trace_shadow_one(s, shadow[i].origin, shadow[i].direction, shadow[i].max_distance, nullptr);
		for (size_t i = 0; i < shadow.size(); i++) if (!occluded[i]) c.aov_set(RT_AOV_RADIANCE, shadow[i].pixel_index, make_float4(1.0f));

		pixels_left -= batch_size;
	}

	for (int i = 0; i < range_count; i++) { // kernel_accumulate of AO.cu: no ALBEDO
		int idx = i + range_offset;
		int x = idx % s.screen_width, y = idx / s.screen_width;
		int pixel_index = x + y * s.screen_pitch;
		float frames_accumulated = float(sample_index);
		auto accumulate = [&](int aov) -> float4 {
			if (!c.f.framebuffer[aov]) return make_float4(0.0f);
			float * fb = c.f.framebuffer[aov] + size_t(pixel_index) * 4;
			float * acc = c.f.accumulator[aov] + size_t(pixel_index) * 4;
			for (int k = 0; k < 4; k++) {
				if (frames_accumulated > 0.0f) acc[k] += (fb[k] - acc[k]) / frames_accumulated;
				else                           acc[k] = fb[k];
			}
			return make_float4(acc[0], acc[1], acc[2], acc[3]);
		};
		float4 colour = accumulate(RT_AOV_RADIANCE);
		accumulate(RT_AOV_NORMAL);
		accumulate(RT_AOV_POSITION);
		if (!std::isfinite(colour.x + colour.y + colour.z)) colour = make_float4(1000.0f, 0.0f, 1000.0f, 1.0f);
		float * out = c.f.final_image + size_t(pixel_index) * 4;
		out[0] = colour.x; out[1] = colour.y; out[2] = colour.z; out[3] = colour.w;
	}
	// aovs_clear_to_zero (AO.cpp:193)
	for (int i = 0; i < RT_AOV_COUNT; i++) if (c.f.framebuffer[i]) memset(c.f.framebuffer[i], 0, size_t(s.screen_pitch) * s.screen_height * 16);
	if (counters) *counters = local;
}

static void render_sample_impl(const oracle_scene * scene, oracle_frame * frame, int sample_index, int range_offset, int range_count, oracle_counters * counters, int threads, bool finish);

void oracle_render_sample(const oracle_scene * scene, oracle_frame * frame, int sample_index,
                          int range_offset, int range_count, oracle_counters * counters, int threads) {
	render_sample_impl(scene, frame, sample_index, range_offset, range_count, counters, threads, true);
}

// The two halves of an SVGF frame for the multi-GPU tile split: every rank path-traces its own pixels and
// leaves the per-frame AOVs and g-buffers in place; once the ranks have exchanged them, each filters the whole frame.
void oracle_render_sample_unfiltered(const oracle_scene * scene, oracle_frame * frame, int sample_index,
                                     int range_offset, int range_count, oracle_counters * counters, int threads) {
	render_sample_impl(scene, frame, sample_index, range_offset, range_count, counters, threads, false);
}
void oracle_filter_frame(const oracle_scene * scene, oracle_frame * frame, int sample_index) {
	oracle_svgf_taa(*scene, *frame, sample_index);
	for (int a = 0; a < RT_AOV_COUNT; a++) if (frame->framebuffer[a]) memset(frame->framebuffer[a], 0, size_t(scene->screen_pitch) * scene->screen_height * 4 * sizeof(float));
}

static void render_sample_impl(const oracle_scene * scene, oracle_frame * frame, int sample_index,
                               int range_offset, int range_count, oracle_counters * counters, int threads, bool finish) {
	const oracle_scene & s = *scene;
	if (threads <= 0) threads = oracle_default_threads();
	g_oracle_threads = threads < 32 ? threads : 32;   // chunks of the sort / shade kernels: joining many small pieces costs more than it saves
	if (const char * e = getenv("ORACLE_CHUNKS")) g_oracle_threads = atoi(e) > 0 ? atoi(e) : 1;
	const bool profile = getenv("ORACLE_PROFILE") != nullptr;
	double t_trace = 0.0, t_sort = 0.0, t_shade = 0.0, t_shadow = 0.0, t_rest = 0.0, t_mark = omp_get_wtime();
	auto lap = [&](double & bucket) { double now = omp_get_wtime(); bucket += now - t_mark; t_mark = now; };
	Context c(s, *frame);
	Wavefront w;
	oracle_counters local; memset(&local, 0, sizeof(local));

	bool has[4] = { false, false, false, false };
	bool has_lights = false;
	for (int i = 0; i < s.material_count; i++) {
		switch (s.material_types[i]) {
			case RT_MATERIAL_DIFFUSE: has[0] = true; break;
			case RT_MATERIAL_PLASTIC: has[1] = true; break;
			case RT_MATERIAL_DIELECTRIC: has[2] = true; break;
			case RT_MATERIAL_CONDUCTOR: has[3] = true; break;
			case RT_MATERIAL_LIGHT: { const float * m = s.materials + size_t(i) * 8; has_lights |= (m[0] * m[0] + m[1] * m[1] + m[2] * m[2]) > 0.0f; break; }
		}
	}

	if ((has[2] && (!s.lut_dielectric_directional_albedo_enter || !s.lut_dielectric_albedo_enter)) || (has[3] && (!s.lut_conductor_directional_albedo || !s.lut_conductor_albedo))) {
		fprintf(stderr, "oracle_render_sample: the scene has dielectric / conductor materials but no Kulla-Conty LUTs were supplied (SceneView(..., luts=...))\n");
		abort();
	}

	int pixels_left = range_count;
	int batch_size  = range_count < RT_BATCH_SIZE ? range_count : RT_BATCH_SIZE;
	while (pixels_left > 0) { // Pathtracer.cpp:746-796
		int pixel_offset = range_offset + (range_count - pixels_left);
		int pixel_count  = batch_size < pixels_left ? batch_size : pixels_left;

		w.trace[0].assign(size_t(pixel_count), TraceRay());
		#pragma omp parallel for schedule(static) num_threads(threads)
		for (int index = 0; index < pixel_count; index++) { // kernel_generate
			int index_offset = index + pixel_offset;
			int x = index_offset % s.screen_width, y = index_offset / s.screen_width;
			int pixel_index = x + y * s.screen_pitch;
			TraceRay & r = w.trace[0][index];
			camera_generate_ray(c, pixel_index, sample_index, x, y, r.origin, r.direction);
			r.pixel_index_and_flags = unsigned(pixel_index);
		}

		for (int bounce = 0; bounce < s.config.num_bounces; bounce++) {
			std::vector<TraceRay> & rays = w.trace[bounce & 1];
			w.trace[(bounce + 1) & 1].clear();
			for (auto & q : w.material) q.clear();
			w.shadow.clear();
			local.trace[bounce] += int(rays.size());

			// kernel_trace_bvhN
			#pragma omp parallel num_threads(threads)
			{
				oracle_trace_stats st = { 0, 0, 0, 0, 0 };
				#pragma omp for schedule(dynamic, 1024)
				for (long long i = 0; i < (long long)rays.size(); i++) oracle_trace_one(s, rays[i].origin, rays[i].direction, rays[i].hit, &st);
				#pragma omp critical
				{ local.trace_stats.nodes += st.nodes; local.trace_stats.triangles += st.triangles; local.trace_stats.instances_transformed += st.instances_transformed; local.trace_stats.instances_identity += st.instances_identity; local.trace_stats.rays += st.rays; }
			}
			lap(t_trace);

			{	// kernel_sort: the input queue is read in place, outputs are appended chunk by chunk
				const std::vector<TraceRay> & in = w.trace[bounce & 1];
				run_in_chunks(in.size(), w, [&](size_t begin, size_t end, Wavefront & piece) { kernel_sort(c, in, begin, end, piece, bounce, sample_index); });
			}
			lap(t_sort);
			local.diffuse[bounce] += int(w.material[0].size()); local.plastic[bounce] += int(w.material[1].size());
			local.dielectric[bounce] += int(w.material[2].size()); local.conductor[bounce] += int(w.material[3].size());

			if (has[0]) { const std::vector<MaterialRay> queue = w.material[0]; run_in_chunks(queue.size(), w, [&](size_t begin, size_t end, Wavefront & piece) { shade_material<BSDFDiffuse>(c, piece, queue, begin, end, bounce, sample_index); }); }
			if (has[1]) { const std::vector<MaterialRay> queue = w.material[1]; run_in_chunks(queue.size(), w, [&](size_t begin, size_t end, Wavefront & piece) { shade_material<BSDFPlastic>(c, piece, queue, begin, end, bounce, sample_index); }); }
			if (has[2]) { const std::vector<MaterialRay> queue = w.material[2]; run_in_chunks(queue.size(), w, [&](size_t begin, size_t end, Wavefront & piece) { shade_material<BSDFDielectric>(c, piece, queue, begin, end, bounce, sample_index); }); }
			if (has[3]) { const std::vector<MaterialRay> queue = w.material[3]; run_in_chunks(queue.size(), w, [&](size_t begin, size_t end, Wavefront & piece) { shade_material<BSDFConductor>(c, piece, queue, begin, end, bounce, sample_index); }); }

			lap(t_shade);
			if (has_lights && s.config.enable_next_event_estimation) { // kernel_trace_shadow_bvhN + miss lambda (Pathtracer.cu:183-196)
				local.shadow[bounce] += int(w.shadow.size());
				std::vector<uint8_t> occluded(w.shadow.size());
				#pragma omp parallel num_threads(threads)
				{
					oracle_trace_stats st = { 0, 0, 0, 0, 0 };
					#pragma omp for schedule(dynamic, 1024)
					for (long long i = 0; i < (long long)w.shadow.size(); i++) occluded[i] = oracle_trace_shadow_one(s, w.shadow[i].origin, w.shadow[i].direction, w.shadow[i].max_distance, &st);
					#pragma omp critical
					{ local.shadow_stats.nodes += st.nodes; local.shadow_stats.triangles += st.triangles; local.shadow_stats.instances_transformed += st.instances_transformed; local.shadow_stats.instances_identity += st.instances_identity; local.shadow_stats.rays += st.rays; }
				}
				for (size_t i = 0; i < w.shadow.size(); i++) {
					if (occluded[i]) continue;
					const ShadowRay & sr = w.shadow[i];
					c.aov_add(RT_AOV_RADIANCE, sr.pixel_index, make_float4(sr.illumination));
					if (bounce == 0) c.aov_set(RT_AOV_RADIANCE_DIRECT,   sr.pixel_index, make_float4(sr.illumination));
					else             c.aov_add(RT_AOV_RADIANCE_INDIRECT, sr.pixel_index, make_float4(sr.illumination));
				}
			}
			lap(t_shadow);
		}
		pixels_left -= batch_size;
	}

	if (finish) {
		if (s.config.enable_svgf) oracle_svgf_taa(s, *frame, sample_index);
		else kernel_accumulate(c, float(sample_index), range_offset, range_count);

		// aovs_clear_to_zero (Integrator.cpp:379-385): framebuffers are zeroed for the next frame
		for (int a = 0; a < RT_AOV_COUNT; a++) {
			if (frame->framebuffer[a]) memset(frame->framebuffer[a], 0, size_t(s.screen_pitch) * s.screen_height * 4 * sizeof(float));
		}
	}
	if (counters) *counters = local;
	lap(t_rest);
	if (profile) fprintf(stderr, "[oracle] sample %d: trace %.2f s, sort %.2f s, shade %.2f s, shadow %.2f s, generate + accumulate / filter %.2f s (%d threads, %d chunks)\n", sample_index, t_trace, t_sort, t_shade, t_shadow, t_rest, threads, g_oracle_threads);
}

} // extern "C"
