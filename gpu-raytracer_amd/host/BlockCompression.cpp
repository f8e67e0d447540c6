// BC1 (DXT1) block compression of textures, as the reference applies it to every power-of-two texture
// by default (Src/Assets/TextureLoader.cpp:208-262, cpu_config.enable_block_compression) through the
// stb_dxt.h v1.12 it vendors, STB_DXT_HIGHQUAL, no dithering, no alpha.
//
// The encoder below restates that algorithm (range fit along the principal axis found by 4 power iterations,
// then up to two least-squares refinements of the end points) so that it produces the same 8 bytes per
// block; tests/test_loaders.py compares it with stb_compress_dxt_block compiled verbatim into oracle/_ref.
// Its two constant tables are generated instead of transcribed:
//   * single-colour match: for a target value t the 5/6-bit pair (max, min) that minimises
//     100 * |(2 * e(max) + e(min)) / 3 - t| + 3 * |e(max) - e(min)|, first minimum in (min, max) order
//   * quantisation midpoints: (e(q) + e(q+1)) / 510 rounded to 6 decimals
// with e() the bit-replicating expansion to 8 bits.
// CDNA has no texture unit, so compressed textures are immediately decoded back to RGBA8
// (ImageDecoders: the D3D interpolation rules); what reaches the device is the quantised image.
#include "BlockCompression.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace {

inline int expand5(int v) { return (v * 33) >> 2; }
inline int expand6(int v) { return (v * 65) >> 4; }
inline int mul_8bit(int a, int b) { int t = a * b + 128; return (t + (t >> 8)) >> 8; }
inline int third(int a, int b) { return (2 * a + b) / 3; } // the colour one third of the way from a to b

struct Tables {
	unsigned char match5[256][2], match6[256][2]; // [target] -> { max, min } quantised end points
	float midpoint5[32], midpoint6[64];

	Tables() {
		build_match(match5, 32, expand5);
		build_match(match6, 64, expand6);
		build_midpoints(midpoint5, 32, expand5);
		build_midpoints(midpoint6, 64, expand6);
	}
	static void build_match(unsigned char table[256][2], int size, int (*expand)(int)) {
		for (int target = 0; target < 256; target++) {
			int best = 1 << 30;
			for (int lo = 0; lo < size; lo++) {
				for (int hi = 0; hi < size; hi++) {
					int e_lo = expand(lo), e_hi = expand(hi);
					int error = abs(third(e_hi, e_lo) - target) * 100 + abs(e_hi - e_lo) * 3;
					if (error < best) { best = error; table[target][0] = (unsigned char)hi; table[target][1] = (unsigned char)lo; }
				}
			}
		}
	}
	static void build_midpoints(float * table, int size, int (*expand)(int)) {
		for (int q = 0; q + 1 < size; q++) table[q] = float(std::round((expand(q) + expand(q + 1)) / 510.0 * 1e6) / 1e6);
		table[size - 1] = 1.0f;
	}
};
const Tables & tables() { static const Tables t; return t; }

inline uint16_t pack_565(int r, int g, int b) { return uint16_t((mul_8bit(r, 31) << 11) + (mul_8bit(g, 63) << 5) + mul_8bit(b, 31)); }

// The four colours of a block with end points c0, c1 (4-colour mode): c0, c1, 2/3 c0 + 1/3 c1, 1/3 c0 + 2/3 c1
void palette_of(uint16_t c0, uint16_t c1, int palette[4][3]) {
	const uint16_t end[2] = { c0, c1 };
	for (int k = 0; k < 2; k++) {
		palette[k][0] = expand5((end[k] >> 11) & 31);
		palette[k][1] = expand6((end[k] >> 5) & 63);
		palette[k][2] = expand5(end[k] & 31);
	}
	for (int c = 0; c < 3; c++) {
		palette[2][c] = third(palette[0][c], palette[1][c]);
		palette[3][c] = third(palette[1][c], palette[0][c]);
	}
}

// Index of the nearest palette colour for each texel, by projecting onto the c0 - c1 axis (2 bits per texel,
// texel 0 in the low bits). Index order along the axis is 1, 3, 2, 0.
uint32_t match_indices(const unsigned char * block, const int palette[4][3]) {
	int dir[3] = { palette[0][0] - palette[1][0], palette[0][1] - palette[1][1], palette[0][2] - palette[1][2] };
	int stop[4];
	for (int k = 0; k < 4; k++) stop[k] = palette[k][0] * dir[0] + palette[k][1] * dir[1] + palette[k][2] * dir[2];
	int c0_point = stop[1] + stop[3], half_point = stop[3] + stop[2], c3_point = stop[2] + stop[0];

	uint32_t mask = 0;
	for (int i = 15; i >= 0; i--) {
		int dot = 2 * (block[i * 4] * dir[0] + block[i * 4 + 1] * dir[1] + block[i * 4 + 2] * dir[2]);
		mask <<= 2;
		if (dot < half_point) mask |= dot < c0_point ? 1u : 3u;
		else                  mask |= dot < c3_point ? 2u : 0u;
	}
	return mask;
}

// Initial end points: the texels that are extreme along the principal axis of the colour distribution
void fit_range(const unsigned char * block, uint16_t & max16, uint16_t & min16) {
	int mean[3], lo[3], hi[3];
	for (int c = 0; c < 3; c++) {
		int sum = 0; lo[c] = 255; hi[c] = 0;
		for (int i = 0; i < 16; i++) { int v = block[i * 4 + c]; sum += v; lo[c] = v < lo[c] ? v : lo[c]; hi[c] = v > hi[c] ? v : hi[c]; }
		mean[c] = (sum + 8) >> 4;
	}
	int cov[6] = { 0, 0, 0, 0, 0, 0 }; // rr rg rb gg gb bb
	for (int i = 0; i < 16; i++) {
		int r = block[i * 4] - mean[0], g = block[i * 4 + 1] - mean[1], b = block[i * 4 + 2] - mean[2];
		cov[0] += r * r; cov[1] += r * g; cov[2] += r * b; cov[3] += g * g; cov[4] += g * b; cov[5] += b * b;
	}
	float covf[6];
	for (int i = 0; i < 6; i++) covf[i] = cov[i] / 255.0f;

	float axis[3] = { float(hi[0] - lo[0]), float(hi[1] - lo[1]), float(hi[2] - lo[2]) };
	for (int iteration = 0; iteration < 4; iteration++) {
		float r = axis[0] * covf[0] + axis[1] * covf[1] + axis[2] * covf[2];
		float g = axis[0] * covf[1] + axis[1] * covf[3] + axis[2] * covf[4];
		float b = axis[0] * covf[2] + axis[1] * covf[4] + axis[2] * covf[5];
		axis[0] = r; axis[1] = g; axis[2] = b;
	}
	double magnitude = fabs(axis[0]);
	if (fabs(axis[1]) > magnitude) magnitude = fabs(axis[1]);
	if (fabs(axis[2]) > magnitude) magnitude = fabs(axis[2]);

	int weight[3];
	if (magnitude < 4.0f) { // no direction to speak of: luma
		weight[0] = 299; weight[1] = 587; weight[2] = 114;
	} else {
		magnitude = 512.0 / magnitude;
		for (int c = 0; c < 3; c++) weight[c] = int(axis[c] * magnitude);
	}
	int min_dot = block[0] * weight[0] + block[1] * weight[1] + block[2] * weight[2], max_dot = min_dot;
	const unsigned char * min_texel = block, * max_texel = block;
	for (int i = 1; i < 16; i++) {
		int dot = block[i * 4] * weight[0] + block[i * 4 + 1] * weight[1] + block[i * 4 + 2] * weight[2];
		if (dot < min_dot) { min_dot = dot; min_texel = block + i * 4; }
		if (dot > max_dot) { max_dot = dot; max_texel = block + i * 4; }
	}
	max16 = pack_565(max_texel[0], max_texel[1], max_texel[2]);
	min16 = pack_565(min_texel[0], min_texel[1], min_texel[2]);
}

inline uint16_t quantise(float x, int levels, const float * midpoint) {
	x = x < 0 ? 0 : x > 1 ? 1 : x;
	uint16_t q = uint16_t(x * float(levels));
	q = uint16_t(q + (x > midpoint[q]));
	return q;
}

// Least-squares end points for the current index assignment (normal equations, Cramer's rule).
// Returns whether the end points changed.
bool refine(const unsigned char * block, uint16_t & max16, uint16_t & min16, uint32_t mask) {
	const Tables & t = tables();
	static const int WEIGHT_OF_C0[4] = { 3, 0, 2, 1 };                              // thirds of c0 in palette entry 0..3
	static const int PRODUCTS[4] = { 0x090000, 0x000900, 0x040102, 0x010402 };      // w0^2 << 16 | w1^2 << 8 | w0 * w1
	uint16_t old_min = min16, old_max = max16;

	if ((mask ^ (mask << 2)) < 4) { // every texel has the same index: fit the average colour with the single-colour tables
		int r = 8, g = 8, b = 8;
		for (int i = 0; i < 16; i++) { r += block[i * 4]; g += block[i * 4 + 1]; b += block[i * 4 + 2]; }
		r >>= 4; g >>= 4; b >>= 4;
		max16 = uint16_t((t.match5[r][0] << 11) | (t.match6[g][0] << 5) | t.match5[b][0]);
		min16 = uint16_t((t.match5[r][1] << 11) | (t.match6[g][1] << 5) | t.match5[b][1]);
	} else {
		int at1[3] = { 0, 0, 0 }, at2[3] = { 0, 0, 0 }, packed = 0;
		uint32_t m = mask;
		for (int i = 0; i < 16; i++, m >>= 2) {
			int step = int(m & 3), w = WEIGHT_OF_C0[step];
			packed += PRODUCTS[step];
			for (int c = 0; c < 3; c++) { at1[c] += w * block[i * 4 + c]; at2[c] += block[i * 4 + c]; }
		}
		for (int c = 0; c < 3; c++) at2[c] = 3 * at2[c] - at1[c];
		int xx = packed >> 16, yy = (packed >> 8) & 0xff, xy = packed & 0xff;
		float f = 3.0f / 255.0f / float(xx * yy - xy * xy);

		max16 = uint16_t(quantise(float(at1[0] * yy - at2[0] * xy) * f, 31, t.midpoint5) << 11);
		max16 = uint16_t(max16 | quantise(float(at1[1] * yy - at2[1] * xy) * f, 63, t.midpoint6) << 5);
		max16 = uint16_t(max16 | quantise(float(at1[2] * yy - at2[2] * xy) * f, 31, t.midpoint5));
		min16 = uint16_t(quantise(float(at2[0] * xx - at1[0] * xy) * f, 31, t.midpoint5) << 11);
		min16 = uint16_t(min16 | quantise(float(at2[1] * xx - at1[1] * xy) * f, 63, t.midpoint6) << 5);
		min16 = uint16_t(min16 | quantise(float(at2[2] * xx - at1[2] * xy) * f, 31, t.midpoint5));
	}
	return old_min != min16 || old_max != max16;
}

} // namespace

void BlockCompression::compress_bc1_block(const unsigned char rgba[64], unsigned char dst[8]) {
	const Tables & t = tables();
	uint16_t max16, min16;
	uint32_t mask;
	int palette[4][3];

	bool constant = true;
	for (int i = 1; i < 16 && constant; i++) constant = memcmp(rgba + 4 * i, rgba, 4) == 0; // alpha takes part in this test
	if (constant) {
		int r = rgba[0], g = rgba[1], b = rgba[2];
		mask  = 0xaaaaaaaau;
		max16 = uint16_t((t.match5[r][0] << 11) | (t.match6[g][0] << 5) | t.match5[b][0]);
		min16 = uint16_t((t.match5[r][1] << 11) | (t.match6[g][1] << 5) | t.match5[b][1]);
	} else {
		fit_range(rgba, max16, min16);
		if (max16 != min16) { palette_of(max16, min16, palette); mask = match_indices(rgba, palette); }
		else mask = 0;

		for (int pass = 0; pass < 2; pass++) { // STB_DXT_HIGHQUAL: two refinement passes
			uint32_t last_mask = mask;
			if (refine(rgba, max16, min16, mask)) {
				if (max16 != min16) { palette_of(max16, min16, palette); mask = match_indices(rgba, palette); }
				else { mask = 0; break; }
			}
			if (mask == last_mask) break;
		}
	}
	if (max16 < min16) { // c0 > c1 selects the 4-colour mode: swap the ends and with them indices 0 <-> 1, 2 <-> 3
		std::swap(max16, min16);
		mask ^= 0x55555555u;
	}
	dst[0] = (unsigned char)max16; dst[1] = (unsigned char)(max16 >> 8);
	dst[2] = (unsigned char)min16; dst[3] = (unsigned char)(min16 >> 8);
	dst[4] = (unsigned char)mask;  dst[5] = (unsigned char)(mask >> 8); dst[6] = (unsigned char)(mask >> 16); dst[7] = (unsigned char)(mask >> 24);
}

void BlockCompression::quantise_level_bc1(unsigned char * rgba, int width, int height, std::vector<unsigned char> * blocks) {
	for (int by = 0; by < (height + 3) / 4; by++) {
		for (int bx = 0; bx < (width + 3) / 4; bx++) {
			unsigned char block[64] = { }; // texels beyond the level's edge stay zero, as in the reference
			for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
				int x = bx * 4 + i, y = by * 4 + j;
				if (x < width && y < height) memcpy(block + 4 * (j * 4 + i), rgba + 4 * (size_t(y) * width + x), 4);
			}
			unsigned char compressed[8], decoded[16][4];
			compress_bc1_block(block, compressed);
			if (blocks) blocks->insert(blocks->end(), compressed, compressed + 8);
			decode_bc1_block(compressed, decoded);
			for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
				int x = bx * 4 + i, y = by * 4 + j;
				if (x < width && y < height) memcpy(rgba + 4 * (size_t(y) * width + x), decoded[j * 4 + i], 4);
			}
		}
	}
}
