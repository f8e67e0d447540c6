// oracle_sampling.h -- hashes, the random<Dim>() sample generator and the warps built on it.
// TEST INFRASTRUCTURE ONLY. Restates CUDA/Util.h:87-149 and CUDA/Sampling.h:18-190.
#pragma once
#include "oracle.h"
#include "oracle_math.h"

#define O_PI          3.14159265359f
#define O_ONE_OVER_PI 0.31830988618f
#define O_TWO_PI          6.28318530718f
#define O_ONE_OVER_TWO_PI 0.15915494309f

enum SampleDimension { DIM_FILTER = 0, DIM_APERTURE, DIM_RUSSIAN_ROULETTE, DIM_NEE_LIGHT, DIM_NEE_TRIANGLE, DIM_BSDF_0, DIM_BSDF_1, DIM_NUM_DIMENSIONS, DIM_NUM_BOUNCE = 5 };

// Util.h:104-108
static inline uint32_t pcg_hash(uint32_t seed) {
	uint32_t state = seed * 747796405u + 2891336453u;
	uint32_t word  = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
	return (word >> 22u) ^ word;
}
// Util.h:114-121 (Wang hash)
static inline uint32_t hash_with(uint32_t seed, uint32_t hash) {
	seed = (seed ^ 61) ^ hash;
	seed += seed << 3;
	seed ^= seed >> 4;
	seed *= 0x27d4eb2d;
	return seed;
}
// Util.h:124-149 (pbrt-v4 PermutationElement, length is a power of two)
static inline uint32_t permute(uint32_t index, uint32_t length, uint32_t seed) {
	uint32_t mask = length - 1;
	index ^= seed;
	index *= 0xe170893d;
	index ^= seed >> 16;
	index ^= (index & mask) >> 4;
	index ^= seed >> 8;
	index *= 0x0929eb3f;
	index ^= seed >> 23;
	index ^= (index & mask) >> 1;
	index *= 1 | seed >> 27;
	index *= 0x6935fa69;
	index ^= (index & mask) >> 11;
	index *= 0x74dcb303;
	index ^= (index & mask) >> 2;
	index *= 0x9e501cc3;
	index ^= (index & mask) >> 2;
	index *= 0xc860a3df;
	index &= mask;
	index ^= index >> 5;
	return (index + seed) & mask;
}

// Sampling.h:44-84
static inline float2 oracle_random_sample(const oracle_scene & s, int Dim, uint32_t pixel_index, uint32_t bounce, uint32_t sample_index) {
	uint32_t hash = pcg_hash((pixel_index * uint32_t(DIM_NUM_DIMENSIONS) + uint32_t(Dim)) * RT_MAX_BOUNCES + bounce);

	if (sample_index >= RT_PMJ_NUM_SAMPLES_PER_SEQUENCE) { // out of PMJ samples: hashed random
		const float one_over_max_unsigned = uint_as_float(0x2f7fffff);
		float x = float(hash_with(sample_index,               hash)) * one_over_max_unsigned;
		float y = float(hash_with(sample_index + 0xdeadbeefu, hash)) * one_over_max_unsigned;
		return make_float2(x, y);
	}

	uint32_t dim = uint32_t(Dim) + uint32_t(DIM_NUM_BOUNCE) * bounce;
	if (dim >= RT_PMJ_NUM_SEQUENCES) sample_index = permute(sample_index, RT_PMJ_NUM_SAMPLES_PER_SEQUENCE, hash);

	const float * seq = s.pmj_samples + size_t(dim % RT_PMJ_NUM_SEQUENCES) * RT_PMJ_NUM_SAMPLES_PER_SEQUENCE * 2;
	float2 sample = make_float2(seq[2 * sample_index], seq[2 * sample_index + 1]);

	// Cranley-Patterson rotation by a blue-noise tile
	const uint8_t * tile = s.blue_noise + size_t(dim % RT_BLUE_NOISE_NUM_TEXTURES) * (RT_BLUE_NOISE_TEXTURE_DIM * RT_BLUE_NOISE_TEXTURE_DIM) * 2;
	int x = int((pixel_index % uint32_t(s.screen_pitch)) % RT_BLUE_NOISE_TEXTURE_DIM);
	int y = int((pixel_index / uint32_t(s.screen_pitch)) % RT_BLUE_NOISE_TEXTURE_DIM);
	const uint8_t * bn = tile + size_t(x + y * RT_BLUE_NOISE_TEXTURE_DIM) * 2;
	sample.x = sample.x + float(bn[0]) * (1.0f / 255.0f);
	sample.y = sample.y + float(bn[1]) * (1.0f / 255.0f);
	if (sample.x >= 1.0f) sample.x -= 1.0f;
	if (sample.y >= 1.0f) sample.y -= 1.0f;
	return sample;
}

static inline float square(float x) { return x * x; }
static inline float safe_sqrt(float x) { return sqrtf(fmaxf(0.0f, x)); }
static inline float2 sincos_pair(float x) { return make_float2(sinf(x), cosf(x)); } // Util.h:191-195 (__sincosf)

static inline float sample_tent(float u) { // Sampling.h:86-92
	if (u < 0.5f) return safe_sqrt(2.0f * u) - 1.0f;
	return 1.0f - safe_sqrt(2.0f - 2.0f * u);
}
static inline float2 sample_gaussian(float u1, float u2) { // Sampling.h:95-99 (Box-Muller)
	float f = sqrtf(-2.0f * logf(u1));
	float a = O_TWO_PI * u2;
	return f * sincos_pair(a);
}
static inline float sample_exp(float lambda, float u) { return -logf(u) / lambda; } // Sampling.h:101-103
static inline float2 sample_triangle(float u1, float u2) { // Sampling.h:106-115
	if (u2 > u1) { u1 *= 0.5f; u2 -= u1; } else { u2 *= 0.5f; u1 -= u2; }
	return make_float2(u1, u2);
}
static inline float2 sample_disk(float u1, float u2) { // Sampling.h:118-132 (concentric map)
	float a = 2.0f * u1 - 1.0f;
	float b = 2.0f * u2 - 1.0f;
	float phi, r;
	if (a * a > b * b) { r = a; phi = 0.25f * O_PI * (b / a); }
	else               { r = b; phi = 0.5f * O_PI - 0.25f * O_PI * (a / b); }
	return r * sincos_pair(phi);
}
static inline float3 sample_cosine_weighted_direction(float u1, float u2) { // Sampling.h:134-137
	float2 d = sample_disk(u1, u2);
	return make_float3(d.x, d.y, safe_sqrt(1.0f - dot(d, d)));
}

// Util.h:212-219
static inline void orthonormal_basis(float3 normal, float3 & tangent, float3 & binormal) {
	float sign = copysignf(1.0f, normal.z);
	float a = -1.0f / (sign + normal.z);
	float b = normal.x * normal.y * a;
	tangent  = make_float3(1.0f + sign * normal.x * normal.x * a, sign * b, -sign * normal.x);
	binormal = make_float3(b, sign + normal.y * normal.y * a, -normal.y);
}
static inline float3 local_to_world(float3 v, float3 t, float3 b, float3 n) { // Util.h:221-227
	return make_float3(
		t.x * v.x + b.x * v.y + n.x * v.z,
		t.y * v.x + b.y * v.y + n.y * v.z,
		t.z * v.x + b.z * v.y + n.z * v.z);
}
static inline float3 world_to_local(float3 v, float3 t, float3 b, float3 n) { return make_float3(dot(t, v), dot(b, v), dot(n, v)); }

// Sampling.h:140-156 (PBRT v3)
static inline float3 sample_henyey_greenstein(float3 omega, float g, float u1, float u2) {
	float cos_theta;
	if (fabsf(g) < 1e-3f) cos_theta = 1.0f - 2.0f * u1;
	else cos_theta = -(1.0f + g * g - square((1.0f - g * g) / (1.0f + g - 2.0f * g * u1))) / (2.0f * g);
	float sin_theta = safe_sqrt(1.0f - square(cos_theta));
	float2 sc = sincos_pair(O_TWO_PI * u2);
	float3 direction = make_float3(sin_theta * sc.x, sin_theta * sc.y, cos_theta);
	float3 v1, v2;
	orthonormal_basis(omega, v1, v2);
	return local_to_world(direction, v1, v2, omega);
}

template<typename T> static inline T lerp_ref(const T & a, const T & b, float t) { return (1.0f - t) * a + t * b; } // Util.h:205-208

// Sampling.h:159-178 (Heitz 2018)
static inline float3 sample_visible_normals_ggx(float3 omega, float alpha_x, float alpha_y, float u1, float u2) {
	float3 v = normalize(make_float3(alpha_x * omega.x, alpha_y * omega.y, omega.z));
	float length_squared = v.x * v.x + v.y * v.y;
	float3 axis_1 = length_squared > 0.0f ? make_float3(-v.y, v.x, 0.0f) / sqrtf(length_squared) : make_float3(1.0f, 0.0f, 0.0f);
	float3 axis_2 = cross(v, axis_1);
	float2 d = sample_disk(u1, u2);
	float t1 = d.x;
	float t2 = lerp_ref(safe_sqrt(1.0f - t1 * t1), d.y, 0.5f + 0.5f * v.z);
	float3 n_h = t1 * axis_1 + t2 * axis_2 + safe_sqrt(1.0f - t1 * t1 - t2 * t2) * v;
	return normalize(make_float3(alpha_x * n_h.x, alpha_y * n_h.y, n_h.z));
}

// Util.h:87-102: first index whose cumulative value is >= value
static inline int binary_search(const float * cdf, int index_first, int index_last, float value) {
	int left = index_first, right = index_last;
	while (true) {
		int middle = (left + right) / 2;
		if (middle > index_first && value <= cdf[middle - 1]) right = middle - 1;
		else if (value > cdf[middle]) left = middle + 1;
		else return middle;
	}
}

static inline bool pdf_is_valid(float pdf) { return std::isfinite(pdf) && pdf > 1e-4f; } // Sampling.h:18-20
static inline float power_heuristic(float f, float g) { return (f * f) / (f * f + g * g); } // Sampling.h:26-28
static inline float luminance(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; } // Util.h:66-68
