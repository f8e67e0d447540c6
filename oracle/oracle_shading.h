// oracle_shading.h -- materials, textures, BSDFs, sky, media. TEST INFRASTRUCTURE ONLY.
// Restates CUDA/Material.h, CUDA/BSDF.h, CUDA/KullaConty.h:12-81, CUDA/RayCone.h, CUDA/Sky.h,
// CUDA/Medium.h, CUDA/Raytracing/Ray.h:16-28, CUDA/Raytracing/Triangle.h:105-146.
//
// Texture unit emulation: the reference samples through NVIDIA texture hardware
// (tex2D / tex2DLod / tex2DGrad on wrap-addressed, linearly filtered, mip-mapped arrays,
// Integrator.cpp:70-93; clamp-addressed linear LUT/sky textures, CUDAMemory.cpp:116-132),
// whose exact filter arithmetic is unspecified ("parity unpinned").  The rules
// below are this project's software definition of those fetches; the HIP shade kernel
// implements the same rules:
//   texel centre convention x = s*W - 0.5, wrap (textures) or clamp (sky, LUTs) addressing,
//   bilinear weights in full fp32, trilinear between floor(lod) and floor(lod)+1 with lod
//   clamped to [0, levels-1], anisotropic footprints approximated by N <= 16 trilinear
//   probes along the major axis at lod = log2(major / N) (EXT_texture_filter_anisotropic).
#pragma once
#include "oracle_sampling.h"

#define O_EPSILON 0.0001f
#define ROUGHNESS_CUTOFF 0.05f
#define LUT_DIELECTRIC_MIN_IOR 1.0001f
#define LUT_DIELECTRIC_MAX_IOR 2.5f

static inline float roughness_to_alpha(float r) { return fmaxf(1e-6f, square(r)); } // Material.h:7-9
static inline float sign_of(float x) { return copysignf(1.0f, x); }
static inline float abs_dot(float3 a, float3 b) { return fabsf(dot(a, b)); }
static inline float remap(float v, float a0, float a1, float b0, float b1) { return b0 + (v - a0) / (a1 - a0) * (b1 - b0); } // Util.h:171-173

// ---- Ray.h:16-28 ----------------------------------------------------------------------------
static inline float3 ray_origin_epsilon_offset(float3 origin, float3 direction, float3 geometric_normal) {
	return origin + sign_of(dot(direction, geometric_normal)) * O_EPSILON * geometric_normal;
}
static inline float3 reflect_direction(float3 direction, float3 normal) { return 2.0f * dot(direction, normal) * normal - direction; }
static inline float3 refract_direction(float3 direction, float3 normal, float eta) {
	float cos_theta = dot(direction, normal);
	float k = 1.0f - eta * eta * (1.0f - square(cos_theta));
	return (eta * cos_theta - safe_sqrt(k)) * normal - eta * direction;
}

// ---- textures -----------------------------------------------------------------------------------
static inline int wrap_index(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }
static inline float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }
static inline float4 lerp4(float4 a, float4 b, float t) { return make_float4(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t)); }

static inline float4 texture_texel(const oracle_texture & tex, int level, int x, int y) {
	int w = tex.width >> level;  if (w < 1) w = 1;
	int h = tex.height >> level; if (h < 1) h = 1;
	size_t offset = 0;
	for (int l = 0; l < level; l++) { int lw = tex.width >> l; if (lw < 1) lw = 1; int lh = tex.height >> l; if (lh < 1) lh = 1; offset += size_t(lw) * lh; }
	const uint8_t * p = tex.texels + (offset + size_t(wrap_index(x, w)) + size_t(wrap_index(y, h)) * w) * 4;
	return make_float4(float(p[0]) * (1.0f / 255.0f), float(p[1]) * (1.0f / 255.0f), float(p[2]) * (1.0f / 255.0f), float(p[3]) * (1.0f / 255.0f));
}
static inline float4 texture_bilinear(const oracle_texture & tex, int level, float s, float t) {
	int w = tex.width >> level;  if (w < 1) w = 1;
	int h = tex.height >> level; if (h < 1) h = 1;
	float x = s * float(w) - 0.5f, y = t * float(h) - 0.5f;
	float x0f = floorf(x), y0f = floorf(y);
	float fx = x - x0f, fy = y - y0f;
	int x0 = int(x0f), y0 = int(y0f);
	float4 c00 = texture_texel(tex, level, x0, y0),     c10 = texture_texel(tex, level, x0 + 1, y0);
	float4 c01 = texture_texel(tex, level, x0, y0 + 1), c11 = texture_texel(tex, level, x0 + 1, y0 + 1);
	return lerp4(lerp4(c00, c10, fx), lerp4(c01, c11, fx), fy);
}
static inline float4 texture_get(const oracle_texture & tex, float s, float t) { return texture_bilinear(tex, 0, s, t); } // tex2D
static inline float4 texture_get_lod(const oracle_texture & tex, float s, float t, float lod) {                         // tex2DLod
	float max_level = float(tex.mip_levels - 1);
	lod = fminf(fmaxf(lod, 0.0f), max_level);
	float l0f = floorf(lod);
	int l0 = int(l0f), l1 = l0 + 1 < tex.mip_levels ? l0 + 1 : l0;
	float fl = lod - l0f;
	float4 a = texture_bilinear(tex, l0, s, t);
	if (fl == 0.0f || l1 == l0) return a;
	return lerp4(a, texture_bilinear(tex, l1, s, t), fl);
}
static inline float4 texture_get_grad(const oracle_texture & tex, float s, float t, float2 dx, float2 dy) {              // tex2DGrad
	float w = float(tex.width), h = float(tex.height);
	float px = sqrtf(square(dx.x * w) + square(dx.y * h));
	float py = sqrtf(square(dy.x * w) + square(dy.y * h));
	float p_max = fmaxf(px, py), p_min = fminf(px, py);
	float2 major = px >= py ? dx : dy;
	float n_f = fminf(ceilf(p_max / fmaxf(p_min, 1e-12f)), 16.0f);
	if (!(n_f >= 1.0f)) n_f = 1.0f; // NaN / degenerate footprints
	int n = int(n_f);
	float lod = log2f(fmaxf(p_max / n_f, 1e-12f));
	float4 sum = make_float4(0.0f);
	for (int i = 0; i < n; i++) {
		float o = (float(i) + 0.5f) / n_f - 0.5f;
		sum += texture_get_lod(tex, s + major.x * o, t + major.y * o, lod);
	}
	return sum * (1.0f / n_f);
}

// Clamp-addressed linear fetch of a float table (LUTs), normalised coordinates.
static inline void clamp_taps(float coord, int n, int & i0, int & i1, float & f) {
	float x = coord * float(n) - 0.5f;
	float x0 = floorf(x);
	f = x - x0;
	i0 = int(x0); i1 = i0 + 1;
	i0 = i0 < 0 ? 0 : (i0 > n - 1 ? n - 1 : i0);
	i1 = i1 < 0 ? 0 : (i1 > n - 1 ? n - 1 : i1);
}
static inline float lut_get_1d(const float * lut, int nx, float s) {
	int a, b; float f; clamp_taps(s, nx, a, b, f);
	return lerpf(lut[a], lut[b], f);
}
static inline float lut_get_2d(const float * lut, int nx, int ny, float s, float t) {
	int x0, x1, y0, y1; float fx, fy;
	clamp_taps(s, nx, x0, x1, fx); clamp_taps(t, ny, y0, y1, fy);
	float r0 = lerpf(lut[x0 + y0 * nx], lut[x1 + y0 * nx], fx);
	float r1 = lerpf(lut[x0 + y1 * nx], lut[x1 + y1 * nx], fx);
	return lerpf(r0, r1, fy);
}
static inline float lut_get_3d(const float * lut, int nx, int ny, int nz, float s, float t, float r) {
	int z0, z1; float fz; clamp_taps(r, nz, z0, z1, fz);
	float a = lut_get_2d(lut + size_t(z0) * nx * ny, nx, ny, s, t);
	float b = lut_get_2d(lut + size_t(z1) * nx * ny, nx, ny, s, t);
	return lerpf(a, b, fz);
}

// ---- Sky.h:7-16 ------------------------------------------------------------------------------
static inline float3 sample_sky(const oracle_scene & s, float3 direction) {
	float phi   = atan2f(-direction.z, direction.x);
	float theta = acosf(clampf(direction.y, -1.0f, 1.0f));
	float u = phi   * O_ONE_OVER_TWO_PI + 0.5f;
	float v = theta * O_ONE_OVER_PI;

	int x0, x1, y0, y1; float fx, fy;
	clamp_taps(u, s.sky_width, x0, x1, fx); clamp_taps(v, s.sky_height, y0, y1, fy);
	auto texel = [&](int x, int y) { const float * p = s.sky + (size_t(x) + size_t(y) * s.sky_width) * 4; return make_float4(p[0], p[1], p[2], p[3]); };
	float4 c = lerp4(lerp4(texel(x0, y0), texel(x1, y0), fx), lerp4(texel(x0, y1), texel(x1, y1), fx), fy);
	return s.sky_scale * make_float3(c);
}

// ---- Medium.h ----------------------------------------------------------------------------------
struct HomogeneousMedium { float3 sigma_a, sigma_s; float g; };
static inline HomogeneousMedium medium_as_homogeneous(const oracle_scene & s, int medium_id) {
	const float * m = s.media + size_t(medium_id) * 8;
	return { make_float3(m[0], m[1], m[2]), make_float3(m[4], m[5], m[6]), m[3] };
}
static inline float3 beer_lambert(float3 sigma_t, float distance) {
	return make_float3(expf(-sigma_t.x * distance), expf(-sigma_t.y * distance), expf(-sigma_t.z * distance));
}

// ---- Material.h:145-222 ---------------------------------------------------------------------------
static inline float divide_difference_by_sum(float a, float b) { return (a - b) / (a + b); }
static inline float3 divide_difference_by_sum(float3 a, float3 b) { return (a - b) / (a + b); }

static inline float fresnel_dielectric(float cos_theta_i, float eta) {
	float sin_theta_o2 = eta * eta * (1.0f - square(cos_theta_i));
	if (sin_theta_o2 >= 1.0f) return 1.0f; // total internal reflection
	float cos_theta_o = safe_sqrt(1.0f - sin_theta_o2);
	float p = divide_difference_by_sum(eta * cos_theta_i, cos_theta_o);
	float s = divide_difference_by_sum(cos_theta_i, eta * cos_theta_o);
	return 0.5f * (p * p + s * s);
}
static inline float3 safe_sqrt3(float3 v) { return make_float3(safe_sqrt(v.x), safe_sqrt(v.y), safe_sqrt(v.z)); }
static inline float3 fresnel_conductor(float cos_theta_i, float3 eta, float3 k) {
	float cos_theta_i2 = square(cos_theta_i);
	float sin_theta_i2 = 1.0f - cos_theta_i2;
	float3 inner      = eta * eta - k * k - sin_theta_i2;
	float3 a2_plus_b2 = safe_sqrt3(inner * inner + 4.0f * k * k * eta * eta);
	float3 a          = safe_sqrt3(0.5f * (a2_plus_b2 + inner));
	float3 s2 = divide_difference_by_sum(a2_plus_b2 + cos_theta_i2, 2.0f * a * cos_theta_i);
	float3 p2 = divide_difference_by_sum(a2_plus_b2 * cos_theta_i2 + square(sin_theta_i2), 2.0f * a * cos_theta_i * sin_theta_i2) * s2;
	return 0.5f * (p2 + s2);
}
static inline float average_fresnel(float ior) { return (ior - 1.0f) / (4.08567f + 1.00071f * ior); }
static inline float3 average_fresnel(float3 eta, float3 k) {
	float3 numerator   = eta * (133.736f - 98.9833f * eta) + k * (eta * (59.5617f - 3.98288f * eta) - 182.37f) + ((0.30818f * eta - 13.1093f) * eta - 62.5919f) * k * k - 8.21474f;
	float3 denominator = k * (eta * (94.6517f - 15.8558f * eta) - 187.166f) + (-78.476f * eta - 395.268f) * eta + (eta * (eta - 15.4387f) - 62.0752f) * k * k;
	return numerator / denominator;
}
static inline float ggx_D(float3 m, float ax, float ay) {
	if (m.z < 1e-6f) return 0.0f;
	float sx = -m.x / (m.z * ax);
	float sy = -m.y / (m.z * ay);
	float sl = 1.0f + sx * sx + sy * sy;
	float cos_theta_2 = m.z * m.z;
	float cos_theta_4 = cos_theta_2 * cos_theta_2;
	return 1.0f / (sl * sl * O_PI * ax * ay * cos_theta_4);
}
static inline float ggx_lambda(float3 w, float ax, float ay) { return 0.5f * (sqrtf(1.0f + (square(ax * w.x) + square(ay * w.y)) / square(w.z)) - 1.0f); }
static inline float ggx_G1(float3 w, float ax, float ay) { return 1.0f / (1.0f + ggx_lambda(w, ax, ay)); }
static inline float ggx_G2(float3 wo, float3 wi, float3 wm, float ax, float ay) {
	bool i_back = dot(wi, wm) * wi.z <= 0.0f;
	bool o_back = dot(wo, wm) * wo.z <= 0.0f;
	if (i_back || o_back) return 0.0f;
	return 1.0f / (1.0f + ggx_lambda(wo, ax, ay) + ggx_lambda(wi, ax, ay));
}

// ---- KullaConty.h:12-81 ---------------------------------------------------------------------------
static inline float3 fresnel_multiscatter(float3 F_avg, float E_avg) { return F_avg * F_avg * E_avg / (make_float3(1.0f) - F_avg * (1.0f - E_avg)); }
static inline float dielectric_directional_albedo(const oracle_scene & s, float ior, float roughness, float cos_theta, bool entering) {
	ior = remap(ior, LUT_DIELECTRIC_MIN_IOR, LUT_DIELECTRIC_MAX_IOR, 0.0f, 1.0f);
	cos_theta = fabsf(cos_theta);
	return lut_get_3d(entering ? s.lut_dielectric_directional_albedo_enter : s.lut_dielectric_directional_albedo_leave, 16, 16, 16, ior, roughness, cos_theta);
}
static inline float dielectric_albedo(const oracle_scene & s, float ior, float roughness, bool entering) {
	ior = remap(ior, LUT_DIELECTRIC_MIN_IOR, LUT_DIELECTRIC_MAX_IOR, 0.0f, 1.0f);
	return lut_get_2d(entering ? s.lut_dielectric_albedo_enter : s.lut_dielectric_albedo_leave, 16, 16, ior, roughness);
}
static inline float conductor_directional_albedo(const oracle_scene & s, float roughness, float cos_theta) {
	return lut_get_2d(s.lut_conductor_directional_albedo, 32, 32, roughness, fabsf(cos_theta));
}
static inline float conductor_albedo(const oracle_scene & s, float roughness) { return lut_get_1d(s.lut_conductor_albedo, 32, roughness); }
static inline float kulla_conty_multiscatter_lobe(float E_i, float E_o, float E_avg) { return (1.0f - E_i) * (1.0f - E_o) / fmaxf(0.0001f, O_PI * (1.0f - E_avg)); }
static inline float kulla_conty_dielectric_reciprocity_factor(float E_avg_enter, float E_avg_leave) { return (1.0f - E_avg_leave) / fmaxf(0.0001f, 2.0f - E_avg_enter - E_avg_leave); }
