// rt_shading.h -- device-side sampling, texturing and BSDF code for the shade kernels.
//
// Replaces CUDA/Sampling.h, CUDA/Material.h, CUDA/BSDF.h, CUDA/KullaConty.h:12-81,
// CUDA/RayCone.h, CUDA/Sky.h, CUDA/Medium.h and the hash / basis helpers of CUDA/Util.h.
// MI355X has no texture units: albedo / sky / LUT fetches are software-filtered buffer loads
// (wrap or clamp addressing, bilinear, trilinear, probe-based anisotropic -- rules in DESIGN.md).
#pragma once
#include "rt_math.h"

#define RT_PI          3.14159265359f
#define RT_ONE_OVER_PI 0.31830988618f
#define RT_TWO_PI          6.28318530718f
#define RT_ONE_OVER_TWO_PI 0.15915494309f
#define RT_EPSILON 0.0001f
#define RT_ROUGHNESS_CUTOFF 0.05f
#define RT_LUT_DIELECTRIC_MIN_IOR 1.0001f
#define RT_LUT_DIELECTRIC_MAX_IOR 2.5f

enum { DIM_FILTER = 0, DIM_APERTURE, DIM_RUSSIAN_ROULETTE, DIM_NEE_LIGHT, DIM_NEE_TRIANGLE, DIM_BSDF_0, DIM_BSDF_1, DIM_NUM_DIMENSIONS, DIM_NUM_BOUNCE = 5 };

// ---- integer hashing: bit-exact with the reference (CUDA/Util.h:104-149) ----------------------
RT_DEV unsigned pcg_hash(unsigned seed) {
	unsigned state = seed * 747796405u + 2891336453u;
	unsigned word  = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
	return (word >> 22u) ^ word;
}
RT_DEV unsigned hash_with(unsigned seed, unsigned hash) {
	seed = (seed ^ 61u) ^ hash;
	seed += seed << 3;
	seed ^= seed >> 4;
	seed *= 0x27d4eb2du;
	return seed;
}
RT_DEV unsigned permute(unsigned index, unsigned length, unsigned seed) {
	unsigned mask = length - 1;
	index ^= seed;
	index *= 0xe170893du;
	index ^= seed >> 16;
	index ^= (index & mask) >> 4;
	index ^= seed >> 8;
	index *= 0x0929eb3fu;
	index ^= seed >> 23;
	index ^= (index & mask) >> 1;
	index *= 1u | seed >> 27;
	index *= 0x6935fa69u;
	index ^= (index & mask) >> 11;
	index *= 0x74dcb303u;
	index ^= (index & mask) >> 2;
	index *= 0x9e501cc3u;
	index ^= (index & mask) >> 2;
	index *= 0xc860a3dfu;
	index &= mask;
	index ^= index >> 5;
	return (index + seed) & mask;
}

// random<Dim>() of the reference (CUDA/Sampling.h:44-84), in two steps: what depends on the path alone -- the pixel behind the
// virtual index, its cell of the blue-noise tile (two divisions by the pitch), the sample number -- is computed once per hit
// (random_path), a shade kernel then draws its four pairs of numbers from it. Same integers, same floats as the one-step form.
struct RandomPath { unsigned pixel, sample_index, blue_noise_cell; };
RT_DEV RandomPath random_path(const RtParams & p, unsigned pixel_index, unsigned sample_index) {
	// callers pass the virtual pixel index of the path and the first sample of the batch (rt_types.h)
	RandomPath r;
	unsigned sample_in_batch;
	r.pixel = rt_split_virtual_pixel(p, pixel_index, sample_in_batch);
	r.sample_index = sample_index + sample_in_batch;
	unsigned x = (r.pixel % unsigned(p.screen_pitch)) % RT_BLUE_NOISE_TEXTURE_DIM;
	unsigned y = (r.pixel / unsigned(p.screen_pitch)) % RT_BLUE_NOISE_TEXTURE_DIM;
	r.blue_noise_cell = x + y * RT_BLUE_NOISE_TEXTURE_DIM;
	return r;
}
RT_DEV f2 random_sample(const RtParams & p, const RandomPath & path, int dimension, unsigned bounce) {
	unsigned sample_index = path.sample_index;
	unsigned hash = pcg_hash((path.pixel * unsigned(DIM_NUM_DIMENSIONS) + unsigned(dimension)) * RT_MAX_BOUNCES + bounce);

	if (sample_index >= RT_PMJ_NUM_SAMPLES_PER_SEQUENCE) {
		const float one_over_max_unsigned = __uint_as_float(0x2f7fffffu);
		float x = float(hash_with(sample_index,               hash)) * one_over_max_unsigned;
		float y = float(hash_with(sample_index + 0xdeadbeefu, hash)) * one_over_max_unsigned;
		return mk2(x, y);
	}

	unsigned dim = unsigned(dimension) + unsigned(DIM_NUM_BOUNCE) * bounce;
	if (dim >= RT_PMJ_NUM_SEQUENCES) sample_index = permute(sample_index, RT_PMJ_NUM_SAMPLES_PER_SEQUENCE, hash);

	float2 s = p.pmj_samples[(dim % RT_PMJ_NUM_SEQUENCES) * RT_PMJ_NUM_SAMPLES_PER_SEQUENCE + sample_index];
	uchar2 bn = p.blue_noise[(dim % RT_BLUE_NOISE_NUM_TEXTURES) * (RT_BLUE_NOISE_TEXTURE_DIM * RT_BLUE_NOISE_TEXTURE_DIM) + path.blue_noise_cell];

	f2 sample = mk2(s.x + float(bn.x) * (1.0f / 255.0f), s.y + float(bn.y) * (1.0f / 255.0f));
	if (sample.x >= 1.0f) sample.x -= 1.0f;
	if (sample.y >= 1.0f) sample.y -= 1.0f;
	return sample;
}
RT_DEV f2 random_sample(const RtParams & p, int dimension, unsigned pixel_index, unsigned bounce, unsigned sample_index) {
	return random_sample(p, random_path(p, pixel_index, sample_index), dimension, bounce);
}

// ---- warps (CUDA/Sampling.h:86-178) ---------------------------------------------------------------
// sinf and cosf of this library each reduce the argument and evaluate BOTH polynomials before they pick one by quadrant (ocml
// sincosred): sincosf does that once and returns the same two floats (the reference calls the hardware approximation __sincosf here,
// Util.h:191-195; the oracle calls libm)
RT_DEV f2 sincos_pair(float x) { float s, c; sincosf(x, &s, &c); return mk2(s, c); }

RT_DEV float sample_tent(float u) {
	if (u < 0.5f) return safe_sqrt(2.0f * u) - 1.0f;
	return 1.0f - safe_sqrt(2.0f - 2.0f * u);
}
RT_DEV f2 sample_gaussian(float u1, float u2) {
	float f = sqrtf(-2.0f * logf(u1));
	float a = RT_TWO_PI * u2;
	return f * sincos_pair(a);
}
RT_DEV float sample_exp(float lambda, float u) { return -logf(u) / lambda; }
RT_DEV f2 sample_triangle(float u1, float u2) {
	if (u2 > u1) { u1 *= 0.5f; u2 -= u1; } else { u2 *= 0.5f; u1 -= u2; }
	return mk2(u1, u2);
}
RT_DEV f2 sample_disk(float u1, float u2) {
	float a = 2.0f * u1 - 1.0f;
	float b = 2.0f * u2 - 1.0f;
	float phi, r;
	if (a * a > b * b) { r = a; phi = 0.25f * RT_PI * (b / a); }
	else               { r = b; phi = 0.5f * RT_PI - 0.25f * RT_PI * (a / b); }
	return r * sincos_pair(phi);
}
RT_DEV f3 sample_cosine_weighted_direction(float u1, float u2) {
	f2 d = sample_disk(u1, u2);
	return mk3(d.x, d.y, safe_sqrt(1.0f - dot(d, d)));
}
RT_DEV void orthonormal_basis(f3 normal, f3 & tangent, f3 & binormal) {
	float sign = copysignf(1.0f, normal.z);
	float a = -1.0f / (sign + normal.z);
	float b = normal.x * normal.y * a;
	tangent  = mk3(1.0f + sign * normal.x * normal.x * a, sign * b, -sign * normal.x);
	binormal = mk3(b, sign + normal.y * normal.y * a, -normal.y);
}
RT_DEV f3 local_to_world(f3 v, f3 t, f3 b, f3 n) {
	return mk3(t.x * v.x + b.x * v.y + n.x * v.z, t.y * v.x + b.y * v.y + n.y * v.z, t.z * v.x + b.z * v.y + n.z * v.z);
}
RT_DEV f3 world_to_local(f3 v, f3 t, f3 b, f3 n) { return mk3(dot(t, v), dot(b, v), dot(n, v)); }

RT_DEV f3 sample_henyey_greenstein(f3 omega, float g, float u1, float u2) {
	float cos_theta;
	if (fabsf(g) < 1e-3f) cos_theta = 1.0f - 2.0f * u1;
	else cos_theta = -(1.0f + g * g - square((1.0f - g * g) / (1.0f + g - 2.0f * g * u1))) / (2.0f * g);
	float sin_theta = safe_sqrt(1.0f - square(cos_theta));
	f2 sc = sincos_pair(RT_TWO_PI * u2);
	f3 direction = mk3(sin_theta * sc.x, sin_theta * sc.y, cos_theta);
	f3 v1, v2;
	orthonormal_basis(omega, v1, v2);
	return local_to_world(direction, v1, v2, omega);
}

RT_DEV float lerp_ref(float a, float b, float t) { return (1.0f - t) * a + t * b; }
RT_DEV f3 lerp_ref(f3 a, f3 b, float t) { return (1.0f - t) * a + t * b; }
RT_DEV f4 lerp_ref(f4 a, f4 b, float t) { return (1.0f - t) * a + t * b; }

RT_DEV f3 sample_visible_normals_ggx(f3 omega, float alpha_x, float alpha_y, float u1, float u2) {
	f3 v = normalize(mk3(alpha_x * omega.x, alpha_y * omega.y, omega.z));
	float length_squared = v.x * v.x + v.y * v.y;
	f3 axis_1 = length_squared > 0.0f ? mk3(-v.y, v.x, 0.0f) / sqrtf(length_squared) : mk3(1.0f, 0.0f, 0.0f);
	f3 axis_2 = cross(v, axis_1);
	f2 d = sample_disk(u1, u2);
	float t1 = d.x;
	float t2 = lerp_ref(safe_sqrt(1.0f - t1 * t1), d.y, 0.5f + 0.5f * v.z);
	f3 n_h = t1 * axis_1 + t2 * axis_2 + safe_sqrt(1.0f - t1 * t1 - t2 * t2) * v;
	return normalize(mk3(alpha_x * n_h.x, alpha_y * n_h.y, n_h.z));
}

template<typename Table>
RT_DEV int binary_search(const Table cdf, int index_first, int index_last, float value) {
	int left = index_first, right = index_last;
	while (true) {
		int middle = (left + right) / 2;
		if (middle > index_first && value <= cdf[middle - 1]) right = middle - 1;
		else if (value > cdf[middle]) left = middle + 1;
		else return middle;
	}
}

RT_DEV bool pdf_is_valid(float pdf) { return isfinite(pdf) && pdf > 1e-4f; }
RT_DEV float power_heuristic(float f, float g) { return (f * f) / (f * f + g * g); }
RT_DEV float luminance(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }
RT_DEV float roughness_to_alpha(float r) { return fmaxf(1e-6f, square(r)); }
RT_DEV float sign_of(float x) { return copysignf(1.0f, x); }
RT_DEV float abs_dot(f3 a, f3 b) { return fabsf(dot(a, b)); }
RT_DEV float remap(float v, float a0, float a1, float b0, float b1) { return b0 + (v - a0) / (a1 - a0) * (b1 - b0); }

RT_DEV f3 ray_origin_epsilon_offset(f3 origin, f3 direction, f3 geometric_normal) {
	return origin + sign_of(dot(direction, geometric_normal)) * RT_EPSILON * geometric_normal;
}
RT_DEV f3 reflect_direction(f3 direction, f3 normal) { return 2.0f * dot(direction, normal) * normal - direction; }
RT_DEV f3 refract_direction(f3 direction, f3 normal, float eta) {
	float cos_theta = dot(direction, normal);
	float k = 1.0f - eta * eta * (1.0f - square(cos_theta));
	return (eta * cos_theta - safe_sqrt(k)) * normal - eta * direction;
}

// ---- software texture unit -----------------------------------------------------------------------
RT_DEV int wrap_index(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }
RT_DEV float lerpf(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }
RT_DEV f4 lerp4(f4 a, f4 b, float t) { return mk4(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t), lerpf(a.w, b.w, t)); }

// One texel of a BC1 block (the texture unit's decode, done here: CDNA compute has none). D3D rules, the same
// integer arithmetic as the host's BlockCompression::decode_bc1_block: end points expand by bit replication,
// c0 > c1: the two thirds rounded to nearest, else the half (rounded down) and transparent black.
RT_DEV uchar4 bc1_texel(uint2 block, int x_in_block, int y_in_block) {
	unsigned c0 = block.x & 0xffffu, c1 = block.x >> 16;
	unsigned index = (block.y >> (2 * (y_in_block * 4 + x_in_block))) & 3u;
	unsigned r0 = c0 >> 11, g0 = (c0 >> 5) & 63u, b0 = c0 & 31u;
	unsigned r1 = c1 >> 11, g1 = (c1 >> 5) & 63u, b1 = c1 & 31u;
	r0 = (r0 << 3) | (r0 >> 2); g0 = (g0 << 2) | (g0 >> 4); b0 = (b0 << 3) | (b0 >> 2);
	r1 = (r1 << 3) | (r1 >> 2); g1 = (g1 << 2) | (g1 >> 4); b1 = (b1 << 3) | (b1 >> 2);
	if (index == 0u) return make_uchar4((unsigned char)r0, (unsigned char)g0, (unsigned char)b0, 255);
	if (index == 1u) return make_uchar4((unsigned char)r1, (unsigned char)g1, (unsigned char)b1, 255);
	if (c0 > c1) { // (2 a + b + 1) / 3 with a the nearer end point
		unsigned ra = index == 2u ? r0 : r1, rb = index == 2u ? r1 : r0;
		unsigned ga = index == 2u ? g0 : g1, gb = index == 2u ? g1 : g0;
		unsigned ba = index == 2u ? b0 : b1, bb = index == 2u ? b1 : b0;
		return make_uchar4((unsigned char)((2u * ra + rb + 1u) / 3u), (unsigned char)((2u * ga + gb + 1u) / 3u), (unsigned char)((2u * ba + bb + 1u) / 3u), 255);
	}
	if (index == 2u) return make_uchar4((unsigned char)((r0 + r1) / 2u), (unsigned char)((g0 + g1) / 2u), (unsigned char)((b0 + b1) / 2u), 255);
	return make_uchar4(0, 0, 0, 0);
}

// A mip level as the filter sees it: where it starts (in texels for RGBA8, in blocks for BC1 and expanded BC1) and its size.
struct RtTextureLevel { size_t offset; int w, h; };
RT_DEV RtTextureLevel texture_level(const RtTexture & tex, int level) {
	RtTextureLevel r; r.offset = 0;
	for (int l = 0; l < level; l++) {
		int lw = max(tex.width >> l, 1), lh = max(tex.height >> l, 1);
		r.offset += tex.format != RT_TEXTURE_RGBA8 ? size_t((lw + 3) >> 2) * ((lh + 3) >> 2) : size_t(lw) * lh;
	}
	r.w = max(tex.width >> level, 1); r.h = max(tex.height >> level, 1);
	return r;
}
RT_DEV RtTextureLevel texture_level_after(const RtTexture & tex, const RtTextureLevel & at, int level) {   // level + 1, from `level`
	RtTextureLevel r;
	r.offset = at.offset + (tex.format != RT_TEXTURE_RGBA8 ? size_t((at.w + 3) >> 2) * ((at.h + 3) >> 2) : size_t(at.w) * at.h);
	r.w = max(tex.width >> (level + 1), 1); r.h = max(tex.height >> (level + 1), 1);
	return r;
}
RT_DEV f4 texel_to_float(uchar4 c) { return mk4(float(c.x) * (1.0f / 255.0f), float(c.y) * (1.0f / 255.0f), float(c.z) * (1.0f / 255.0f), float(c.w) * (1.0f / 255.0f)); }
typedef const __attribute__((address_space(1))) unsigned * GlobalWords;
RT_DEV unsigned texel_word(uchar4 c) { return unsigned(c.x) | (unsigned(c.y) << 8) | (unsigned(c.z) << 16) | (unsigned(c.w) << 24); }
RT_DEV f4 texel_to_float(unsigned c) { return mk4(float(c & 0xffu) * (1.0f / 255.0f), float((c >> 8) & 0xffu) * (1.0f / 255.0f), float((c >> 16) & 0xffu) * (1.0f / 255.0f), float(c >> 24) * (1.0f / 255.0f)); }

// One bilinear footprint: the four texels around (s, t) of one level, wrap addressing, weights in full fp32.
// The two columns and the two rows are wrapped ONCE each -- with a mask where the size is a power of two (every BC1 level
// is), a remainder otherwise: the integer remainders of a tap-by-tap formulation (eight per footprint, ~20 instructions each
// without a divide unit) were most of what a texture lookup cost. Same texels, same arithmetic on them.
// COMPRESSED = false: the caller knows that no texture on the device holds BC1 blocks (RtParams::textures_compressed == 0, the
// default: rt_set_texture_expansion) -- the per-fetch decode is compiled out, and with it 1 300 instructions and the registers it pins.
template<bool COMPRESSED = true>
RT_DEV f4 texture_bilinear(const RtTexture & tex, const RtTextureLevel & lv, float s, float t) {
	const int w = lv.w, h = lv.h;
	float x = s * float(w) - 0.5f, y = t * float(h) - 0.5f;
	float x0f = floorf(x), y0f = floorf(y);
	float fx = x - x0f, fy = y - y0f;
	int x0 = int(x0f), y0 = int(y0f), x1 = x0 + 1, y1 = y0 + 1;
	if ((w & (w - 1)) == 0) { x0 &= w - 1; x1 &= w - 1; } else { x0 = wrap_index(x0, w); x1 = x0 + 1 == w ? 0 : x0 + 1; }
	if ((h & (h - 1)) == 0) { y0 &= h - 1; y1 &= h - 1; } else { y0 = wrap_index(y0, h); y1 = y0 + 1 == h ? 0 : y0 + 1; }
	// Texels travel as one 32-bit word each (r | g << 8 | b << 16 | a << 24) and are read through pointers that carry the global
	// address space: the texel pointer comes out of a table in memory, and a pointer of unknown provenance is dereferenced with FLAT
	// loads (the LDS aperture check in the address path, LDS and memory counters tied together).
	unsigned c00, c10, c01, c11;
	if (!COMPRESSED && tex.format == RT_TEXTURE_BC1_EXPANDED) {   // (a context holds compressed blocks OR expanded ones, never both: rt_upload_textures)
		// the blocks of the BC1 chain, decoded at upload: texel (x, y) is entry (y & 3) * 4 + (x & 3) of block (y >> 2, x >> 2)
		const GlobalWords texels = (GlobalWords)tex.texels + (lv.offset << 4);
		const unsigned blocks_per_row = unsigned(w + 3) >> 2;
		const unsigned row0 = (unsigned(y0) >> 2) * blocks_per_row, row1 = (unsigned(y1) >> 2) * blocks_per_row;   // (32 bits: a level has far fewer than 2^28 blocks)
		const unsigned in0 = (unsigned(y0) & 3u) << 2, in1 = (unsigned(y1) & 3u) << 2;
		const unsigned col0 = ((unsigned(x0) >> 2) << 4) | (unsigned(x0) & 3u), col1 = ((unsigned(x1) >> 2) << 4) | (unsigned(x1) & 3u);
		c00 = texels[(row0 << 4) + in0 + col0]; c10 = texels[(row0 << 4) + in0 + col1];
		c01 = texels[(row1 << 4) + in1 + col0]; c11 = texels[(row1 << 4) + in1 + col1];
	} else if (COMPRESSED && tex.format == RT_TEXTURE_BC1) {
		const GlobalWords blocks = (GlobalWords)tex.texels + (lv.offset << 1);   // 8-byte blocks: two words each
		const int blocks_per_row = (w + 3) >> 2;
		const size_t row0 = size_t(y0 >> 2) * blocks_per_row, row1 = size_t(y1 >> 2) * blocks_per_row;
		const size_t i00 = (row0 + (x0 >> 2)) << 1, i10 = (row0 + (x1 >> 2)) << 1, i01 = (row1 + (x0 >> 2)) << 1, i11 = (row1 + (x1 >> 2)) << 1;
		uint2 b00 = make_uint2(blocks[i00], blocks[i00 + 1]), b10 = make_uint2(blocks[i10], blocks[i10 + 1]), b01 = make_uint2(blocks[i01], blocks[i01 + 1]), b11 = make_uint2(blocks[i11], blocks[i11 + 1]);
		c00 = texel_word(bc1_texel(b00, x0 & 3, y0 & 3)); c10 = texel_word(bc1_texel(b10, x1 & 3, y0 & 3));
		c01 = texel_word(bc1_texel(b01, x0 & 3, y1 & 3)); c11 = texel_word(bc1_texel(b11, x1 & 3, y1 & 3));
	} else {
		const GlobalWords texels = (GlobalWords)tex.texels + lv.offset;
		const size_t row0 = size_t(y0) * w, row1 = size_t(y1) * w;
		c00 = texels[row0 + x0]; c10 = texels[row0 + x1]; c01 = texels[row1 + x0]; c11 = texels[row1 + x1];
	}
	return lerp4(lerp4(texel_to_float(c00), texel_to_float(c10), fx), lerp4(texel_to_float(c01), texel_to_float(c11), fx), fy);
}
template<bool COMPRESSED = true>
RT_DEV f4 texture_get(const RtTexture & tex, float s, float t) { return texture_bilinear<COMPRESSED>(tex, texture_level(tex, 0), s, t); }

// Trilinear between floor(lod) and the next level; the levels are looked up once per filtered fetch, not once per probe.
struct RtTrilinear { RtTextureLevel l0, l1; float fl; bool single; };
RT_DEV RtTrilinear texture_trilinear_levels(const RtTexture & tex, float lod) {
	float max_level = float(tex.mip_levels - 1);
	lod = fminf(fmaxf(lod, 0.0f), max_level);
	float l0f = floorf(lod);
	int l0 = int(l0f), l1 = l0 + 1 < tex.mip_levels ? l0 + 1 : l0;
	RtTrilinear r;
	r.fl = lod - l0f;
	r.single = r.fl == 0.0f || l1 == l0;
	r.l0 = texture_level(tex, l0);
	r.l1 = r.single ? r.l0 : texture_level_after(tex, r.l0, l0);
	return r;
}
template<bool COMPRESSED = true>
RT_DEV f4 texture_trilinear(const RtTexture & tex, const RtTrilinear & tri, float s, float t) {
	f4 a = texture_bilinear<COMPRESSED>(tex, tri.l0, s, t);
	if (tri.single) return a;
	return lerp4(a, texture_bilinear<COMPRESSED>(tex, tri.l1, s, t), tri.fl);
}
template<bool COMPRESSED = true>
RT_DEV f4 texture_get_lod(const RtTexture & tex, float s, float t, float lod) { return texture_trilinear<COMPRESSED>(tex, texture_trilinear_levels(tex, lod), s, t); }
template<bool COMPRESSED = true>
RT_DEV f4 texture_get_grad(const RtTexture & tex, float s, float t, f2 dx, f2 dy) {
	float w = float(tex.width), h = float(tex.height);
	float px = sqrtf(square(dx.x * w) + square(dx.y * h));
	float py = sqrtf(square(dy.x * w) + square(dy.y * h));
	float p_max = fmaxf(px, py), p_min = fminf(px, py);
	f2 major = px >= py ? dx : dy;
	float n_f = fminf(ceilf(p_max / fmaxf(p_min, 1e-12f)), 16.0f);
	if (!(n_f >= 1.0f)) n_f = 1.0f;
	int n = int(n_f);
	float lod = log2f(fmaxf(p_max / n_f, 1e-12f));
	const RtTrilinear tri = texture_trilinear_levels(tex, lod);   // every probe of the footprint filters the same two levels
	f4 sum = mk4(0.0f);
	for (int i = 0; i < n; i++) {
		float o = (float(i) + 0.5f) / n_f - 0.5f;
		sum += texture_trilinear<COMPRESSED>(tex, tri, s + major.x * o, t + major.y * o);
	}
	return sum * (1.0f / n_f);
}

RT_DEV void clamp_taps(float coord, int n, int & i0, int & i1, float & f) {
	float x = coord * float(n) - 0.5f;
	float x0 = floorf(x);
	f = x - x0;
	i0 = int(x0); i1 = i0 + 1;
	i0 = min(max(i0, 0), n - 1);
	i1 = min(max(i1, 0), n - 1);
}
RT_DEV float lut_get_1d(const float * __restrict__ lut, int nx, float s) {
	int a, b; float f; clamp_taps(s, nx, a, b, f);
	return lerpf(lut[a], lut[b], f);
}
RT_DEV float lut_get_2d(const float * __restrict__ lut, int nx, int ny, float s, float t) {
	int x0, x1, y0, y1; float fx, fy;
	clamp_taps(s, nx, x0, x1, fx); clamp_taps(t, ny, y0, y1, fy);
	float r0 = lerpf(lut[x0 + y0 * nx], lut[x1 + y0 * nx], fx);
	float r1 = lerpf(lut[x0 + y1 * nx], lut[x1 + y1 * nx], fx);
	return lerpf(r0, r1, fy);
}
RT_DEV float lut_get_3d(const float * __restrict__ lut, int nx, int ny, int nz, float s, float t, float r) {
	int z0, z1; float fz; clamp_taps(r, nz, z0, z1, fz);
	float a = lut_get_2d(lut + size_t(z0) * nx * ny, nx, ny, s, t);
	float b = lut_get_2d(lut + size_t(z1) * nx * ny, nx, ny, s, t);
	return lerpf(a, b, fz);
}

// CUDA/Sky.h:7-16
RT_DEV f3 sample_sky(const RtParams & p, f3 direction) {
	float phi   = atan2f(-direction.z, direction.x);
	float theta = acosf(clampf(direction.y, -1.0f, 1.0f));
	float u = phi   * RT_ONE_OVER_TWO_PI + 0.5f;
	float v = theta * RT_ONE_OVER_PI;
	int x0, x1, y0, y1; float fx, fy;
	clamp_taps(u, p.sky_width, x0, x1, fx); clamp_taps(v, p.sky_height, y0, y1, fy);
	f4 c00 = mk4(p.sky[x0 + size_t(y0) * p.sky_width]), c10 = mk4(p.sky[x1 + size_t(y0) * p.sky_width]);
	f4 c01 = mk4(p.sky[x0 + size_t(y1) * p.sky_width]), c11 = mk4(p.sky[x1 + size_t(y1) * p.sky_width]);
	f4 c = lerp4(lerp4(c00, c10, fx), lerp4(c01, c11, fx), fy);
	return p.sky_scale * mk3(c);
}

// CUDA/Medium.h
struct HomogeneousMedium { f3 sigma_a, sigma_s; float g; };
RT_DEV HomogeneousMedium medium_as_homogeneous(const RtParams & p, int medium_id) {
	float4 a = p.media[2 * medium_id], s = p.media[2 * medium_id + 1];
	return { mk3(a.x, a.y, a.z), mk3(s.x, s.y, s.z), a.w };
}
RT_DEV f3 beer_lambert(f3 sigma_t, float distance) { return mk3(expf(-sigma_t.x * distance), expf(-sigma_t.y * distance), expf(-sigma_t.z * distance)); }

// ---- microfacet helpers (CUDA/Material.h:145-222) ---------------------------------------------------
RT_DEV float divide_difference_by_sum(float a, float b) { return (a - b) / (a + b); }
RT_DEV f3 divide_difference_by_sum(f3 a, f3 b) { return (a - b) / (a + b); }
RT_DEV float fresnel_dielectric(float cos_theta_i, float eta) {
	float sin_theta_o2 = eta * eta * (1.0f - square(cos_theta_i));
	if (sin_theta_o2 >= 1.0f) return 1.0f;
	float cos_theta_o = safe_sqrt(1.0f - sin_theta_o2);
	float pp = divide_difference_by_sum(eta * cos_theta_i, cos_theta_o);
	float ss = divide_difference_by_sum(cos_theta_i, eta * cos_theta_o);
	return 0.5f * (pp * pp + ss * ss);
}
RT_DEV f3 safe_sqrt3(f3 v) { return mk3(safe_sqrt(v.x), safe_sqrt(v.y), safe_sqrt(v.z)); }
RT_DEV f3 fresnel_conductor(float cos_theta_i, f3 eta, f3 k) {
	float cos_theta_i2 = square(cos_theta_i);
	float sin_theta_i2 = 1.0f - cos_theta_i2;
	f3 inner      = eta * eta - k * k - sin_theta_i2;
	f3 a2_plus_b2 = safe_sqrt3(inner * inner + 4.0f * k * k * eta * eta);
	f3 a          = safe_sqrt3(0.5f * (a2_plus_b2 + inner));
	f3 s2 = divide_difference_by_sum(a2_plus_b2 + cos_theta_i2, 2.0f * a * cos_theta_i);
	f3 p2 = divide_difference_by_sum(a2_plus_b2 * cos_theta_i2 + square(sin_theta_i2), 2.0f * a * cos_theta_i * sin_theta_i2) * s2;
	return 0.5f * (p2 + s2);
}
RT_DEV float average_fresnel(float ior) { return (ior - 1.0f) / (4.08567f + 1.00071f * ior); }
RT_DEV f3 average_fresnel(f3 eta, f3 k) {
	f3 numerator   = eta * (133.736f - 98.9833f * eta) + k * (eta * (59.5617f - 3.98288f * eta) - 182.37f) + ((0.30818f * eta - 13.1093f) * eta - 62.5919f) * k * k - 8.21474f;
	f3 denominator = k * (eta * (94.6517f - 15.8558f * eta) - 187.166f) + (-78.476f * eta - 395.268f) * eta + (eta * (eta - 15.4387f) - 62.0752f) * k * k;
	return numerator / denominator;
}
RT_DEV float ggx_D(f3 m, float ax, float ay) {
	if (m.z < 1e-6f) return 0.0f;
	float sx = -m.x / (m.z * ax);
	float sy = -m.y / (m.z * ay);
	float sl = 1.0f + sx * sx + sy * sy;
	float cos_theta_2 = m.z * m.z;
	float cos_theta_4 = cos_theta_2 * cos_theta_2;
	return 1.0f / (sl * sl * RT_PI * ax * ay * cos_theta_4);
}
RT_DEV float ggx_lambda(f3 w, float ax, float ay) { return 0.5f * (sqrtf(1.0f + (square(ax * w.x) + square(ay * w.y)) / square(w.z)) - 1.0f); }
RT_DEV float ggx_G1(f3 w, float ax, float ay) { return 1.0f / (1.0f + ggx_lambda(w, ax, ay)); }
RT_DEV float ggx_G2(f3 wo, f3 wi, f3 wm, float ax, float ay) {
	bool i_back = dot(wi, wm) * wi.z <= 0.0f;
	bool o_back = dot(wo, wm) * wo.z <= 0.0f;
	if (i_back || o_back) return 0.0f;
	return 1.0f / (1.0f + ggx_lambda(wo, ax, ay) + ggx_lambda(wi, ax, ay));
}

// ---- Kulla-Conty lookups (CUDA/KullaConty.h:12-81) -----------------------------------------------------
RT_DEV f3 fresnel_multiscatter(f3 F_avg, float E_avg) { return F_avg * F_avg * E_avg / (mk3(1.0f) - F_avg * (1.0f - E_avg)); }
RT_DEV float dielectric_directional_albedo(const RtParams & p, float ior, float roughness, float cos_theta, bool entering) {
	ior = remap(ior, RT_LUT_DIELECTRIC_MIN_IOR, RT_LUT_DIELECTRIC_MAX_IOR, 0.0f, 1.0f);
	cos_theta = fabsf(cos_theta);
	return lut_get_3d(entering ? p.lut_dielectric_directional_albedo_enter : p.lut_dielectric_directional_albedo_leave, 16, 16, 16, ior, roughness, cos_theta);
}
RT_DEV float dielectric_albedo(const RtParams & p, float ior, float roughness, bool entering) {
	ior = remap(ior, RT_LUT_DIELECTRIC_MIN_IOR, RT_LUT_DIELECTRIC_MAX_IOR, 0.0f, 1.0f);
	return lut_get_2d(entering ? p.lut_dielectric_albedo_enter : p.lut_dielectric_albedo_leave, 16, 16, ior, roughness);
}
RT_DEV float conductor_directional_albedo(const RtParams & p, float roughness, float cos_theta) { return lut_get_2d(p.lut_conductor_directional_albedo, 32, 32, roughness, fabsf(cos_theta)); }
RT_DEV float conductor_albedo(const RtParams & p, float roughness) { return lut_get_1d(p.lut_conductor_albedo, 32, roughness); }
RT_DEV float kulla_conty_multiscatter_lobe(float E_i, float E_o, float E_avg) { return (1.0f - E_i) * (1.0f - E_o) / fmaxf(0.0001f, RT_PI * (1.0f - E_avg)); }
RT_DEV float kulla_conty_dielectric_reciprocity_factor(float E_avg_enter, float E_avg_leave) { return (1.0f - E_avg_leave) / fmaxf(0.0001f, 2.0f - E_avg_enter - E_avg_leave); }

// ---- AOV access (CUDA/AOV.h:15-33) ---------------------------------------------------------------------
RT_DEV f4 aov_get(const RtParams & p, int aov, int pixel) { return mk4(p.aovs[aov].framebuffer[pixel]); }
RT_DEV void aov_set(const RtParams & p, int aov, int pixel, f4 v) { if (p.aovs[aov].framebuffer) p.aovs[aov].framebuffer[pixel] = to_float4(v); }
RT_DEV void aov_add(const RtParams & p, int aov, int pixel, f4 v) {
	if (p.aovs[aov].framebuffer) { f4 c = mk4(p.aovs[aov].framebuffer[pixel]); p.aovs[aov].framebuffer[pixel] = to_float4(c + v); }
}
