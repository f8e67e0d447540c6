// JPEG decoder for textures: baseline and progressive Huffman DCT, 8 bit, 1 / 3 / 4 components, any
// sampling factors, restart intervals, JFIF (YCbCr), Adobe RGB / CMYK / YCCK.
//
// The reference decodes JPEGs with the stb_image v2.19 it vendors (Src/Assets/TextureLoader.cpp:129,
// e.g. the textures of Data/instancing). What a decoder outputs is fixed by the standard only up to the
// inverse DCT, the chroma upsampling filter and the colour conversion, so those three follow stb_image's
// integer arithmetic exactly (stb_image.h: stbi__idct_block, stbi__resample_row_*, stbi__YCbCr_to_RGB_row)
// and tests/test_loaders.py compares the decoded bytes with that header compiled verbatim (oracle/_ref).
// Everything else -- marker parsing, Huffman decoding, progressive refinement -- is ITU T.81.
#include "ImageDecoders.h"

#include <algorithm>
#include <cstdint>
#include <cstring>

namespace {

const unsigned char ZIGZAG[64] = {
	 0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
	12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
	58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct HuffmanTable {
	bool defined = false;
	// canonical code: for each length 1..16 the first code, the index of its first symbol, and the count
	int first_code[17], first_symbol[17], count[17];
	unsigned char symbols[256];

	bool build(const unsigned char counts[16], const unsigned char * values, int value_count) {
		int code = 0, index = 0;
		for (int length = 1; length <= 16; length++) {
			count[length] = counts[length - 1];
			first_code[length]   = code;
			first_symbol[length] = index;
			code  += count[length];
			index += count[length];
			if (code > (1 << length)) return false; // over-subscribed
			code <<= 1;
		}
		if (index != value_count || index > 256) return false;
		memcpy(symbols, values, size_t(index));
		defined = true;
		return true;
	}
};

// Entropy-coded segment reader: removes stuffed zero bytes, stops at any marker (which it remembers) and then
// delivers zero bits, as every JPEG decoder does for truncated data.
struct BitReader {
	const unsigned char * data;
	size_t size, pos;
	uint32_t buffer = 0;
	int      bits   = 0;
	int      marker = -1;

	void reset() { buffer = 0; bits = 0; marker = -1; }

	void fill() {
		while (bits <= 24) {
			int byte = 0;
			if (marker < 0 && pos < size) {
				byte = data[pos++];
				if (byte == 0xff) {
					int next = pos < size ? data[pos] : 0xd9;
					while (next == 0xff && pos + 1 < size) next = data[++pos]; // fill bytes
					if (next == 0) { pos++; }
					else { marker = next; pos++; byte = 0; }
				}
			}
			buffer |= uint32_t(byte) << (24 - bits);
			bits += 8;
		}
	}
	int get_bit() {
		if (bits < 1) fill();
		int bit = int(buffer >> 31);
		buffer <<= 1; bits--;
		return bit;
	}
	int get_bits(int n) {
		if (n == 0) return 0;
		if (bits < n) fill();
		int value = int(buffer >> (32 - n));
		buffer <<= n; bits -= n;
		return value;
	}
	// T.81 F.2.2.1 EXTEND: n received bits -> signed value
	int receive_extend(int n) {
		if (n == 0) return 0;
		int value = get_bits(n);
		return value < (1 << (n - 1)) ? value - (1 << n) + 1 : value;
	}
	int decode(const HuffmanTable & table) {
		if (bits < 16) fill();
		int code = 0;
		for (int length = 1; length <= 16; length++) {
			code = (code << 1) | int((buffer >> (32 - length)) & 1);
			int offset = code - table.first_code[length];
			if (offset >= 0 && offset < table.count[length]) {
				buffer <<= length; bits -= length;
				return table.symbols[table.first_symbol[length] + offset];
			}
		}
		return -1;
	}
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0;
	int dc_table = 0, ac_table = 0;
	int dc_pred = 0;
	int width = 0, height = 0;         // samples that belong to the image
	int padded_w = 0, padded_h = 0;    // whole MCUs
	std::vector<unsigned char> samples;      // padded_w x padded_h after the inverse DCT
	std::vector<int16_t>       coefficients; // progressive only: 64 per block, blocks in raster order
};

inline unsigned char clamp_u8(long long x) { return (unsigned char)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// jidctint-style 13-bit fixed point butterflies in the form stb_image uses (constants scaled by 4096).
// 64-bit intermediates: identical results for every valid stream, and no signed overflow on corrupt ones.
typedef long long wide;
#define F2F(x) (wide((x) * 4096 + 0.5))
#define IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7) \
	wide t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3; \
	p2 = s2; p3 = s6; \
	p1 = (p2 + p3) * F2F(0.5411961f); \
	t2 = p1 + p3 * F2F(-1.847759065f); \
	t3 = p1 + p2 * F2F( 0.765366865f); \
	p2 = s0; p3 = s4; \
	t0 = (p2 + p3) * 4096; \
	t1 = (p2 - p3) * 4096; \
	x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2; \
	t0 = s7; t1 = s5; t2 = s3; t3 = s1; \
	p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2; \
	p5 = (p3 + p4) * F2F(1.175875602f); \
	t0 = t0 * F2F(0.298631336f); \
	t1 = t1 * F2F(2.053119869f); \
	t2 = t2 * F2F(3.072711026f); \
	t3 = t3 * F2F(1.501321110f); \
	p1 = p5 + p1 * F2F(-0.899976223f); \
	p2 = p5 + p2 * F2F(-2.562915447f); \
	p3 = p3 * F2F(-1.961570560f); \
	p4 = p4 * F2F(-0.390180644f); \
	t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

// Dequantised coefficients (row-major) -> 8x8 samples. Columns first, keeping 2 extra bits; rows second
// with the rounding bias and the +128 level shift folded into one constant.
void inverse_dct(const int16_t block[64], unsigned char * out, int stride) {
	wide column_pass[64];
	for (int i = 0; i < 8; i++) {
		const int16_t * d = block + i;
		wide * v = column_pass + i;
		IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
		x0 += 512; x1 += 512; x2 += 512; x3 += 512;
		v[ 0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
		v[ 8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
		v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
		v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
	}
	for (int i = 0; i < 8; i++) {
		const wide * v = column_pass + 8 * i;
		unsigned char * o = out + size_t(i) * stride;
		IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
		const wide bias = 65536 + (128 << 17);
		x0 += bias; x1 += bias; x2 += bias; x3 += bias;
		o[0] = clamp_u8((x0 + t3) >> 17); o[7] = clamp_u8((x0 - t3) >> 17);
		o[1] = clamp_u8((x1 + t2) >> 17); o[6] = clamp_u8((x1 - t2) >> 17);
		o[2] = clamp_u8((x2 + t1) >> 17); o[5] = clamp_u8((x2 - t1) >> 17);
		o[3] = clamp_u8((x3 + t0) >> 17); o[4] = clamp_u8((x3 - t0) >> 17);
	}
}
#undef IDCT_1D
#undef F2F

struct Decoder {
	const unsigned char * data;
	size_t size, pos = 0;

	bool progressive = false;
	int  width = 0, height = 0;
	int  h_max = 1, v_max = 1, mcus_x = 0, mcus_y = 0;
	int  restart_interval = 0;
	bool jfif = false;
	int  adobe_transform = -1;
	int  rgb_ids = 0;

	std::vector<Component> components;
	uint16_t     quant[4][64] = { };
	HuffmanTable dc_tables[4], ac_tables[4];

	// current scan
	std::vector<int> scan_order;
	int spectral_start = 0, spectral_end = 63, approx_high = 0, approx_low = 0;
	int eob_run = 0;
	BitReader reader;

	int  u8()  { return pos < size ? data[pos++] : -1; }
	int  u16() { int a = u8(), b = u8(); return (a < 0 || b < 0) ? -1 : (a << 8) | b; }

	// next marker code, skipping fill bytes; -1 at the end of the data
	int next_marker() {
		while (pos < size) {
			if (data[pos++] != 0xff) continue;
			while (pos < size && data[pos] == 0xff) pos++;
			if (pos < size && data[pos] != 0) return data[pos++];
		}
		return -1;
	}

	bool read_quant_tables() {
		int length = u16() - 2;
		while (length > 0) {
			int q = u8();
			int precision = q >> 4, table = q & 15;
			if (q < 0 || precision > 1 || table > 3) return false;
			for (int i = 0; i < 64; i++) {
				int value = precision ? u16() : u8();
				if (value < 0) return false;
				quant[table][ZIGZAG[i]] = uint16_t(value);
			}
			length -= precision ? 129 : 65;
		}
		return length == 0;
	}

	bool read_huffman_tables() {
		int length = u16() - 2;
		while (length > 0) {
			int q = u8();
			int kind = q >> 4, table = q & 15;
			if (q < 0 || kind > 1 || table > 3) return false;
			unsigned char counts[16], values[256];
			int total = 0;
			for (int i = 0; i < 16; i++) { int c = u8(); if (c < 0) return false; counts[i] = (unsigned char)c; total += c; }
			if (total > 256) return false;
			for (int i = 0; i < total; i++) { int v = u8(); if (v < 0) return false; values[i] = (unsigned char)v; }
			if (!(kind == 0 ? dc_tables : ac_tables)[table].build(counts, values, total)) return false;
			length -= 17 + total;
		}
		return length == 0;
	}

	bool read_application_segment(int marker) {
		int length = u16();
		if (length < 2) return false;
		size_t end = pos + size_t(length - 2);
		if (end > size) return false;
		if (marker == 0xe0 && length - 2 >= 5 && memcmp(data + pos, "JFIF\0", 5) == 0) jfif = true;
		if (marker == 0xee && length - 2 >= 12 && memcmp(data + pos, "Adobe\0", 6) == 0) adobe_transform = data[pos + 11];
		pos = end;
		return true;
	}

	bool read_frame_header() {
		int length = u16();
		if (length < 11 || u8() != 8) return false; // 8-bit samples only
		height = u16(); width = u16();
		int count = u8();
		if (width <= 0 || height <= 0 || width > (1 << 15) || height > (1 << 15) || size_t(width) * height > (size_t(1) << 28)) return false;
		if (!(count == 1 || count == 3 || count == 4) || length != 8 + 3 * count) return false;
		components.assign(count, Component());
		for (int i = 0; i < count; i++) {
			Component & c = components[i];
			c.id = u8();
			int q = u8();
			c.h = q >> 4; c.v = q & 15; c.tq = u8();
			if (q < 0 || c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq < 0 || c.tq > 3) return false;
			if (count == 3 && c.id == "RGB"[i]) rgb_ids++;
			h_max = std::max(h_max, c.h); v_max = std::max(v_max, c.v);
		}
		mcus_x = (width  + 8 * h_max - 1) / (8 * h_max);
		mcus_y = (height + 8 * v_max - 1) / (8 * v_max);
		for (Component & c : components) {
			c.width    = (width  * c.h + h_max - 1) / h_max;
			c.height   = (height * c.v + v_max - 1) / v_max;
			c.padded_w = mcus_x * c.h * 8;
			c.padded_h = mcus_y * c.v * 8;
			c.samples.assign(size_t(c.padded_w) * c.padded_h, 0);
			if (progressive) c.coefficients.assign(size_t(c.padded_w) * c.padded_h, 0);
		}
		return true;
	}

	bool read_scan_header() {
		int length = u16();
		int count  = u8();
		if (count < 1 || count > 4 || count > int(components.size()) || length != 6 + 2 * count) return false;
		scan_order.clear();
		for (int i = 0; i < count; i++) {
			int id = u8(), q = u8();
			int which = -1;
			for (size_t c = 0; c < components.size(); c++) if (components[c].id == id) { which = int(c); break; }
			if (which < 0 || q < 0 || (q >> 4) > 3 || (q & 15) > 3) return false;
			components[which].dc_table = q >> 4;
			components[which].ac_table = q & 15;
			scan_order.push_back(which);
		}
		spectral_start = u8(); spectral_end = u8();
		int approx = u8();
		if (approx < 0) return false;
		approx_high = approx >> 4; approx_low = approx & 15;
		if (progressive) {
			if (spectral_start > 63 || spectral_end > 63 || spectral_start > spectral_end || approx_high > 13 || approx_low > 13) return false;
			if (spectral_start == 0 && spectral_end != 0) return false; // DC and AC never share a progressive scan
			if (spectral_start != 0 && count != 1) return false;
		} else {
			if (spectral_start != 0 || approx_high != 0 || approx_low != 0) return false;
			spectral_end = 63;
		}
		return true;
	}

	// ---- block decoding ---------------------------------------------------------------------

	bool decode_baseline_block(Component & c, int16_t block[64]) {
		const HuffmanTable & dc = dc_tables[c.dc_table], & ac = ac_tables[c.ac_table];
		if (!dc.defined || !ac.defined) return false;
		memset(block, 0, 64 * sizeof(int16_t));
		int t = reader.decode(dc);
		if (t < 0 || t > 15) return false;
		c.dc_pred = int(uint32_t(c.dc_pred) + uint32_t(reader.receive_extend(t))); // wraps instead of overflowing on hostile streams
		block[0] = int16_t(wide(c.dc_pred) * quant[c.tq][0]);
		for (int k = 1; k < 64;) {
			int rs = reader.decode(ac);
			if (rs < 0) return false;
			int run = rs >> 4, bits = rs & 15;
			if (bits == 0) {
				if (run != 15) break; // end of block
				k += 16;
				continue;
			}
			k += run;
			if (k > 63) return false;
			int zig = ZIGZAG[k++];
			block[zig] = int16_t(reader.receive_extend(bits) * quant[c.tq][zig]);
		}
		return true;
	}

	bool decode_progressive_dc(Component & c, int16_t * block) {
		if (approx_high == 0) {
			const HuffmanTable & dc = dc_tables[c.dc_table];
			if (!dc.defined) return false;
			int t = reader.decode(dc);
			if (t < 0 || t > 15) return false;
			c.dc_pred = int(uint32_t(c.dc_pred) + uint32_t(reader.receive_extend(t)));
			block[0] = int16_t(wide(c.dc_pred) * (1 << approx_low));
		} else if (reader.get_bit()) {
			block[0] = int16_t(block[0] + (1 << approx_low));
		}
		return true;
	}

	// refinement of an already non-zero coefficient: one correction bit moves it away from zero
	void refine(int16_t & coefficient, int16_t bit) {
		if (reader.get_bit() && (coefficient & bit) == 0) coefficient = int16_t(coefficient > 0 ? coefficient + bit : coefficient - bit);
	}

	bool decode_progressive_ac(Component & c, int16_t * block) {
		const HuffmanTable & ac = ac_tables[c.ac_table];
		if (!ac.defined) return false;
		if (approx_high == 0) { // first pass over this band
			if (eob_run) { eob_run--; return true; }
			for (int k = spectral_start; k <= spectral_end;) {
				int rs = reader.decode(ac);
				if (rs < 0) return false;
				int run = rs >> 4, bits = rs & 15;
				if (bits == 0) {
					if (run < 15) { // end of band for 2^run + extra blocks, this one included
						eob_run = (1 << run) - 1;
						if (run) eob_run += reader.get_bits(run);
						break;
					}
					k += 16;
					continue;
				}
				k += run;
				if (k > 63) return false;
				block[ZIGZAG[k++]] = int16_t(reader.receive_extend(bits) * (1 << approx_low));
			}
			return true;
		}
		// refinement pass (T.81 G.1.2.3)
		int16_t bit = int16_t(1 << approx_low);
		if (eob_run) {
			eob_run--;
			for (int k = spectral_start; k <= spectral_end; k++) {
				int16_t & coefficient = block[ZIGZAG[k]];
				if (coefficient != 0) refine(coefficient, bit);
			}
			return true;
		}
		for (int k = spectral_start; k <= spectral_end;) {
			int rs = reader.decode(ac);
			if (rs < 0) return false;
			int run = rs >> 4, bits = rs & 15;
			int new_value = 0;
			if (bits == 0) {
				if (run < 15) {
					eob_run = (1 << run) - 1;
					if (run) eob_run += reader.get_bits(run);
					run = 64; // refine the rest of the band, place nothing
				}
			} else {
				if (bits != 1) return false;
				new_value = reader.get_bit() ? bit : -bit;
			}
			while (k <= spectral_end) { // skip `run` zero-history coefficients, refining the others on the way
				int16_t & coefficient = block[ZIGZAG[k++]];
				if (coefficient != 0) {
					refine(coefficient, bit);
				} else {
					if (run == 0) { coefficient = int16_t(new_value); break; }
					run--;
				}
			}
		}
		return true;
	}

	// ---- scans --------------------------------------------------------------------------------

	void restart() {
		reader.reset();
		for (Component & c : components) c.dc_pred = 0;
		eob_run = 0;
	}

	// One coded block of component c at block coordinates (bx, by)
	bool decode_block_at(Component & c, int bx, int by) {
		if (progressive) {
			int16_t * block = c.coefficients.data() + 64 * (size_t(bx) + size_t(by) * (c.padded_w / 8));
			return spectral_start == 0 ? decode_progressive_dc(c, block) : decode_progressive_ac(c, block);
		}
		int16_t block[64];
		if (!decode_baseline_block(c, block)) return false;
		inverse_dct(block, c.samples.data() + size_t(by) * 8 * c.padded_w + size_t(bx) * 8, c.padded_w);
		return true;
	}

	bool decode_scan() {
		reader.data = data; reader.size = size; reader.pos = pos;
		restart();
		int todo = restart_interval ? restart_interval : 0x7fffffff;
		// after every MCU: count down the restart interval and resynchronise at RSTn
		auto mcu_done = [&]() -> int { // 1: go on, 0: stop decoding this scan (no restart marker where one is due)
			if (--todo > 0) return 1;
			if (reader.bits < 24) reader.fill();
			if (reader.marker < 0xd0 || reader.marker > 0xd7) return 0;
			restart();
			todo = restart_interval ? restart_interval : 0x7fffffff;
			return 1;
		};
		bool stop = false;
		if (scan_order.size() == 1) { // not interleaved: the component's own blocks in raster order
			Component & c = components[scan_order[0]];
			int blocks_x = (c.width + 7) >> 3, blocks_y = (c.height + 7) >> 3;
			for (int by = 0; by < blocks_y && !stop; by++) {
				for (int bx = 0; bx < blocks_x && !stop; bx++) {
					if (!decode_block_at(c, bx, by)) return false;
					if (!mcu_done()) stop = true;
				}
			}
		} else {
			for (int my = 0; my < mcus_y && !stop; my++) {
				for (int mx = 0; mx < mcus_x && !stop; mx++) {
					for (int index : scan_order) {
						Component & c = components[index];
						for (int y = 0; y < c.v; y++) for (int x = 0; x < c.h; x++) {
							if (!decode_block_at(c, mx * c.h + x, my * c.v + y)) return false;
						}
					}
					if (!mcu_done()) stop = true;
				}
			}
		}
		pos = reader.pos;
		if (reader.marker >= 0) pos -= 2; // hand the marker that ended the scan back to the segment parser
		return true;
	}

	void finish_progressive() {
		for (Component & c : components) {
			int blocks_x = (c.width + 7) >> 3, blocks_y = (c.height + 7) >> 3;
			for (int by = 0; by < blocks_y; by++) {
				for (int bx = 0; bx < blocks_x; bx++) {
					int16_t * block = c.coefficients.data() + 64 * (size_t(bx) + size_t(by) * (c.padded_w / 8));
					for (int i = 0; i < 64; i++) block[i] = int16_t(block[i] * quant[c.tq][i]);
					inverse_dct(block, c.samples.data() + size_t(by) * 8 * c.padded_w + size_t(bx) * 8, c.padded_w);
				}
			}
		}
	}

	bool decode_image() {
		if (next_marker() != 0xd8) return false; // SOI
		bool have_frame = false, have_scan = false;
		while (true) {
			int marker = next_marker();
			if (marker < 0) return have_scan; // ran out of data: keep what was decoded
			if (marker == 0xd9) break;        // EOI
			switch (marker) {
				case 0xdb: if (!read_quant_tables())   return false; break;
				case 0xc4: if (!read_huffman_tables()) return false; break;
				case 0xdd: if (u16() != 4) return false; restart_interval = u16(); if (restart_interval < 0) return false; break;
				case 0xc0: case 0xc1: case 0xc2:
					if (have_frame) return false;
					progressive = marker == 0xc2;
					if (!read_frame_header()) return false;
					have_frame = true;
					break;
				case 0xda:
					if (!have_frame || !read_scan_header() || !decode_scan()) return false;
					have_scan = true;
					break;
				case 0xdc: { int l = u16(), lines = u16(); if (l != 4 || lines != height) return false; break; } // DNL
				default:
					if ((marker >= 0xe0 && marker <= 0xef) || marker == 0xfe) { if (!read_application_segment(marker)) return false; }
					else if (marker >= 0xd0 && marker <= 0xd7) { /* stray restart marker */ }
					else return false; // arithmetic coding, lossless, hierarchical, 12 bit: not supported
			}
		}
		if (!have_scan) return false;
		if (progressive) finish_progressive();
		return true;
	}
};

// ---- chroma upsampling (stb_image's "fancy" triangle filters) ---------------------------------

void upsample_row(const Component & c, int hs, int vs, const unsigned char * near, const unsigned char * far, int w, unsigned char * out) {
	auto div4  = [](int x) { return (unsigned char)(x >> 2); };
	auto div16 = [](int x) { return (unsigned char)(x >> 4); };
	(void)c;
	if (hs == 1 && vs == 1) {
		memcpy(out, near, size_t(w));
	} else if (hs == 1 && vs == 2) {
		for (int i = 0; i < w; i++) out[i] = div4(3 * near[i] + far[i] + 2);
	} else if (hs == 2 && vs == 1) {
		if (w == 1) { out[0] = out[1] = near[0]; return; }
		out[0] = near[0];
		out[1] = div4(near[0] * 3 + near[1] + 2);
		int i;
		for (i = 1; i < w - 1; i++) {
			int n = 3 * near[i] + 2;
			out[i * 2]     = div4(n + near[i - 1]);
			out[i * 2 + 1] = div4(n + near[i + 1]);
		}
		out[i * 2]     = div4(near[w - 2] * 3 + near[w - 1] + 2);
		out[i * 2 + 1] = near[w - 1];
	} else if (hs == 2 && vs == 2) {
		if (w == 1) { out[0] = out[1] = div4(3 * near[0] + far[0] + 2); return; }
		int t1 = 3 * near[0] + far[0];
		out[0] = div4(t1 + 2);
		for (int i = 1; i < w; i++) {
			int t0 = t1;
			t1 = 3 * near[i] + far[i];
			out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
			out[i * 2]     = div16(3 * t1 + t0 + 8);
		}
		out[w * 2 - 1] = div4(t1 + 2);
	} else { // any other ratio: nearest neighbour
		for (int i = 0; i < w; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
	}
}

inline unsigned char multiply_8x8(unsigned char a, unsigned char b) { // a * b / 255, rounded
	unsigned t = unsigned(a) * b + 128;
	return (unsigned char)((t + (t >> 8)) >> 8);
}

} // namespace

bool ImageDecoders::decode_jpeg(const std::vector<unsigned char> & file, int & width, int & height, std::vector<unsigned char> & rgba) {
	if (file.size() < 4 || file[0] != 0xff || file[1] != 0xd8) return false;
	Decoder d;
	d.data = file.data(); d.size = file.size();
	if (!d.decode_image()) return false;

	width = d.width; height = d.height;
	int count = int(d.components.size());
	bool is_rgb = count == 3 && (d.rgb_ids == 3 || (d.adobe_transform == 0 && !d.jfif));

	// Per component: which two source rows feed the current output row. The pair advances every vs
	// output rows, and the nearer of the two alternates half way -- stb_image's resampler state machine.
	struct RowState { int hs, vs, ystep, ypos, line0, line1, w_lores; };
	std::vector<RowState> state(count);
	std::vector<std::vector<unsigned char>> line(count, std::vector<unsigned char>(size_t(width) + 8));
	for (int k = 0; k < count; k++) {
		const Component & c = d.components[k];
		state[k] = { d.h_max / c.h, d.v_max / c.v, (d.v_max / c.v) >> 1, 0, 0, 0, 0 };
		state[k].w_lores = (width + state[k].hs - 1) / state[k].hs;
		line[k].resize(size_t(state[k].w_lores) * state[k].hs + 8);
	}

	rgba.resize(size_t(width) * height * 4);
	for (int y = 0; y < height; y++) {
		for (int k = 0; k < count; k++) {
			const Component & c = d.components[k];
			RowState & s = state[k];
			bool near_is_line1 = s.ystep >= (s.vs >> 1);
			const unsigned char * row0 = c.samples.data() + size_t(s.line0) * c.padded_w;
			const unsigned char * row1 = c.samples.data() + size_t(s.line1) * c.padded_w;
			upsample_row(c, s.hs, s.vs, near_is_line1 ? row1 : row0, near_is_line1 ? row0 : row1, s.w_lores, line[k].data());
			if (++s.ystep >= s.vs) {
				s.ystep = 0;
				s.line0 = s.line1;
				if (++s.ypos < c.height) s.line1++;
			}
		}
		unsigned char * out = &rgba[size_t(y) * width * 4];
		for (int x = 0; x < width; x++, out += 4) {
			out[3] = 255;
			if (count == 1) {
				out[0] = out[1] = out[2] = line[0][x];
				continue;
			}
			if (is_rgb || (count == 4 && d.adobe_transform == 0)) { // stored as they are (RGB, or CMYK below)
				out[0] = line[0][x]; out[1] = line[1][x]; out[2] = line[2][x];
			} else { // YCbCr -> RGB, 12-bit coefficients with the low 8 bits cleared: the same in scalar and SIMD stb builds
				int luma = (line[0][x] << 20) + (1 << 19);
				int cb = line[1][x] - 128, cr = line[2][x] - 128;
				auto fixed = [](float v) { return int(v * 4096.0f + 0.5f) << 8; };
				int r = luma + cr * fixed(1.40200f);
				int g = luma + cr * -fixed(0.71414f) + int((cb * -fixed(0.34414f)) & 0xffff0000);
				int b = luma + cb * fixed(1.77200f);
				out[0] = clamp_u8(r >> 20); out[1] = clamp_u8(g >> 20); out[2] = clamp_u8(b >> 20);
			}
			if (count == 4) {
				unsigned char k = line[3][x];
				if (d.adobe_transform == 0) { // CMYK (inverted, as Adobe writes it)
					out[0] = multiply_8x8(out[0], k); out[1] = multiply_8x8(out[1], k); out[2] = multiply_8x8(out[2], k);
				} else if (d.adobe_transform == 2) { // YCCK
					out[0] = multiply_8x8(255 - out[0], k); out[1] = multiply_8x8(255 - out[1], k); out[2] = multiply_8x8(255 - out[2], k);
				}
			}
		}
	}
	return true;
}
