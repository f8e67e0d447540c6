// Image file decoders for textures, beyond TGA / PPM (TextureLoader.cpp): PNG, BMP and
// DXT1/3/5 DDS -- the reference reads these through stb_image and its own DDS reader
// (Src/Assets/TextureLoader.cpp:19-106,129-206). All produce 8-bit RGBA, row 0 = top of the image.
//   PNG: every colour type and bit depth, palette and colour-key transparency, Adam7 interlace;
//        16-bit samples keep their high byte (as stb_image does). Decompression via zlib.
//   BMP: uncompressed 8-bit palettised, 24- and 32-bit (BI_RGB, BI_BITFIELDS), either row order.
//   DDS: block-compressed mip chains are decoded to RGBA8 on the host, because CDNA has no
//        texture unit to hand the blocks to; the values are used as they are (the reference does
//        not gamma-convert DDS data either).
#include "ImageDecoders.h"
#include "BlockCompression.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <zlib.h>

namespace {

uint32_t be32(const unsigned char * p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
uint32_t le32(const unsigned char * p) { return p[0] | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
uint16_t le16(const unsigned char * p) { return uint16_t(p[0] | (p[1] << 8)); }

int paeth(int a, int b, int c) {
	int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undoes the per-scanline filters of one (sub-)image in place; returns false on a bad filter type.
// `rows` scanlines of `stride` bytes, each preceded by its filter byte; bpp = bytes per complete pixel (>= 1).
bool png_unfilter(unsigned char * data, int rows, size_t stride, int bpp) {
	std::vector<unsigned char> zero(stride, 0);
	const unsigned char * prior = zero.data();
	for (int y = 0; y < rows; y++) {
		unsigned char * line = data + size_t(y) * (stride + 1);
		int filter = line[0];
		unsigned char * cur = line + 1;
		for (size_t i = 0; i < stride; i++) {
			int a = i >= size_t(bpp) ? cur[i - bpp] : 0;
			int b = prior[i];
			int c = i >= size_t(bpp) ? prior[i - bpp] : 0;
			int predicted;
			switch (filter) {
				case 0: predicted = 0; break;
				case 1: predicted = a; break;
				case 2: predicted = b; break;
				case 3: predicted = (a + b) >> 1; break;
				case 4: predicted = paeth(a, b, c); break;
				default: return false;
			}
			cur[i] = (unsigned char)(cur[i] + predicted);
		}
		prior = cur;
	}
	return true;
}

struct PNGInfo {
	int width = 0, height = 0, depth = 0, colour_type = 0, interlace = 0;
	int channels() const { return colour_type == 0 ? 1 : colour_type == 2 ? 3 : colour_type == 3 ? 1 : colour_type == 4 ? 2 : 4; }
};

// Sample `index` of a scanline whose samples are `depth` bits wide (big-endian bit order); 16-bit -> high byte
inline int png_sample(const unsigned char * line, size_t index, int depth) {
	switch (depth) {
		case 8:  return line[index];
		case 16: return line[index * 2];
		default: {
			size_t bit = index * depth;
			return (line[bit >> 3] >> (8 - depth - int(bit & 7))) & ((1 << depth) - 1);
		}
	}
}

} // namespace

bool ImageDecoders::decode_png(const std::vector<unsigned char> & file, int & width, int & height, std::vector<unsigned char> & rgba) {
	static const unsigned char SIGNATURE[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
	if (file.size() < 8 || memcmp(file.data(), SIGNATURE, 8) != 0) return false;

	PNGInfo info;
	std::vector<unsigned char> idat, palette, palette_alpha;
	int  key[3] = { -1, -1, -1 }; // colour-key transparency for grey / rgb images (raw sample values)
	bool have_header = false;
	size_t pos = 8;
	while (pos + 12 <= file.size()) {
		uint32_t length = be32(&file[pos]);
		const unsigned char * type = &file[pos + 4];
		const unsigned char * data = &file[pos + 8];
		if (length > file.size() - pos - 12) return false;
		if (memcmp(type, "IHDR", 4) == 0) {
			if (length < 13) return false;
			info.width = int(be32(data)); info.height = int(be32(data + 4));
			info.depth = data[8]; info.colour_type = data[9]; info.interlace = data[12];
			if (data[10] != 0 || data[11] != 0 || info.interlace > 1) return false;
			have_header = true;
		} else if (memcmp(type, "PLTE", 4) == 0) {
			palette.assign(data, data + length);
		} else if (memcmp(type, "tRNS", 4) == 0) {
			if (info.colour_type == 3) palette_alpha.assign(data, data + length);
			else if (info.colour_type == 0 && length >= 2) key[0] = (data[0] << 8) | data[1];
			else if (info.colour_type == 2 && length >= 6) for (int c = 0; c < 3; c++) key[c] = (data[2 * c] << 8) | data[2 * c + 1];
		} else if (memcmp(type, "IDAT", 4) == 0) {
			idat.insert(idat.end(), data, data + length);
		} else if (memcmp(type, "IEND", 4) == 0) {
			break;
		}
		pos += size_t(length) + 12;
	}
	if (!have_header || info.width <= 0 || info.height <= 0 || info.width > (1 << 15) || info.height > (1 << 15)) return false;
	if (size_t(info.width) * info.height > (size_t(1) << 28)) return false; // a quarter gigapixel is beyond any texture
	static const int VALID_DEPTHS[7][6] = { { 1, 2, 4, 8, 16, 0 }, { 0 }, { 8, 16, 0 }, { 1, 2, 4, 8, 0 }, { 8, 16, 0 }, { 0 }, { 8, 16, 0 } };
	if (info.colour_type > 6) return false;
	bool depth_ok = false;
	for (int d : VALID_DEPTHS[info.colour_type]) if (d && d == info.depth) depth_ok = true;
	if (!depth_ok) return false;
	if (info.colour_type == 3 && palette.size() < 3) return false;

	int channels   = info.channels();
	int pixel_bits = channels * info.depth;
	int bpp        = pixel_bits >= 8 ? pixel_bits / 8 : 1;
	auto stride_of = [&](int w) { return (size_t(w) * pixel_bits + 7) / 8; };

	// Sub-images: one for a plain PNG, seven for Adam7
	struct Pass { int x0, y0, dx, dy; };
	static const Pass ADAM7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
	static const Pass WHOLE = { 0, 0, 1, 1 };
	int pass_count = info.interlace ? 7 : 1;
	size_t raw_size = 0;
	for (int p = 0; p < pass_count; p++) {
		const Pass & pass = info.interlace ? ADAM7[p] : WHOLE;
		int w = (info.width - pass.x0 + pass.dx - 1) / pass.dx, h = (info.height - pass.y0 + pass.dy - 1) / pass.dy;
		if (w > 0 && h > 0) raw_size += (stride_of(w) + 1) * h;
	}
	std::vector<unsigned char> raw(raw_size);
	uLongf produced = uLongf(raw_size);
	int status = uncompress(raw.data(), &produced, idat.data(), uLong(idat.size()));
	if ((status != Z_OK && status != Z_BUF_ERROR) || produced != raw_size) return false; // Z_BUF_ERROR: trailing data beyond what the image needs

	width = info.width; height = info.height;
	rgba.assign(size_t(width) * height * 4, 0);
	int max_sample = (1 << (info.depth > 8 ? 8 : info.depth)) - 1;
	size_t offset = 0;
	for (int p = 0; p < pass_count; p++) {
		const Pass & pass = info.interlace ? ADAM7[p] : WHOLE;
		int w = (info.width - pass.x0 + pass.dx - 1) / pass.dx, h = (info.height - pass.y0 + pass.dy - 1) / pass.dy;
		if (w <= 0 || h <= 0) continue;
		size_t stride = stride_of(w);
		if (!png_unfilter(raw.data() + offset, h, stride, bpp)) return false;
		for (int y = 0; y < h; y++) {
			const unsigned char * line = raw.data() + offset + size_t(y) * (stride + 1) + 1;
			for (int x = 0; x < w; x++) {
				unsigned char * out = &rgba[(size_t(pass.y0 + y * pass.dy) * width + (pass.x0 + x * pass.dx)) * 4];
				auto sample = [&](int c) { return png_sample(line, size_t(x) * channels + c, info.depth); };
				auto sample16 = [&](int c) { return info.depth == 16 ? (line[(size_t(x) * channels + c) * 2] << 8) | line[(size_t(x) * channels + c) * 2 + 1] : sample(c); };
				switch (info.colour_type) {
					case 0: {
						int g = sample(0);
						out[0] = out[1] = out[2] = (unsigned char)(info.depth < 8 ? g * 255 / max_sample : g);
						out[3] = (key[0] >= 0 && sample16(0) == key[0]) ? 0 : 255;
						break;
					}
					case 2:
						for (int c = 0; c < 3; c++) out[c] = (unsigned char)sample(c);
						out[3] = (key[0] >= 0 && sample16(0) == key[0] && sample16(1) == key[1] && sample16(2) == key[2]) ? 0 : 255;
						break;
					case 3: {
						size_t index = size_t(sample(0));
						if (index * 3 + 2 >= palette.size()) index = 0;
						for (int c = 0; c < 3; c++) out[c] = palette[index * 3 + c];
						out[3] = index < palette_alpha.size() ? palette_alpha[index] : 255;
						break;
					}
					case 4:
						out[0] = out[1] = out[2] = (unsigned char)sample(0);
						out[3] = (unsigned char)sample(1);
						break;
					default:
						for (int c = 0; c < 4; c++) out[c] = (unsigned char)sample(c);
				}
			}
		}
		offset += (stride + 1) * h;
	}
	return true;
}

bool ImageDecoders::decode_bmp(const std::vector<unsigned char> & file, int & width, int & height, std::vector<unsigned char> & rgba) {
	if (file.size() < 54 || file[0] != 'B' || file[1] != 'M') return false;
	uint32_t data_offset = le32(&file[10]);
	uint32_t header_size = le32(&file[14]);
	if (header_size < 40 || 14 + size_t(header_size) > file.size()) return false;
	int32_t w = int32_t(le32(&file[18])), h = int32_t(le32(&file[22]));
	int bits = le16(&file[28]);
	uint32_t compression = le32(&file[30]);
	bool top_down = h < 0;
	if (top_down) h = -h;
	if (w <= 0 || h <= 0 || w > (1 << 15) || h > (1 << 15) || size_t(w) * h > (size_t(1) << 28)) return false;
	if (!(bits == 8 || bits == 24 || bits == 32)) return false;
	if (!(compression == 0 || (compression == 3 && bits == 32))) return false; // BI_RGB, or BI_BITFIELDS for 32 bit

	// r g b a of BI_RGB; like stb_image, the fourth byte of a 32-bit pixel is alpha unless it is zero everywhere
	uint32_t mask[4] = { 0x00ff0000u, 0x0000ff00u, 0x000000ffu, bits == 32 ? 0xff000000u : 0u };
	if (compression == 3) {
		size_t mask_pos = header_size >= 52 ? 14 + 40 : 14 + size_t(header_size);
		if (mask_pos + 12 > file.size()) return false;
		for (int c = 0; c < 3; c++) mask[c] = le32(&file[mask_pos + 4 * c]);
		if (header_size >= 56) mask[3] = le32(&file[14 + 52]);
	}
	auto extract = [](uint32_t v, uint32_t m) -> unsigned char {
		if (!m) return 255;
		int shift = 0; while (!((m >> shift) & 1)) shift++;
		uint32_t field = (v & m) >> shift, max = m >> shift;
		return (unsigned char)(max == 255 ? field : field * 255 / max);
	};

	// The palette is copied into a full 256-entry table: a pixel byte may name any entry, whatever biClrUsed says
	// (entries the file does not define read as black instead of running past the buffer).
	unsigned char palette[256 * 4] = { };
	if (bits == 8) {
		uint32_t colours = le32(&file[46]); if (colours == 0) colours = 256;
		if (colours > 256) return false;
		if (14 + size_t(header_size) + size_t(colours) * 4 > file.size()) return false;
		memcpy(palette, &file[14 + header_size], size_t(colours) * 4);
	}
	if (size_t(data_offset) < 14 + size_t(header_size)) return false; // pixel data cannot start inside the headers
	size_t row_bytes = ((size_t(w) * bits + 31) / 32) * 4;
	if (size_t(data_offset) + row_bytes * h > file.size()) return false;

	width = w; height = h;
	rgba.resize(size_t(w) * h * 4);
	for (int y = 0; y < h; y++) {
		const unsigned char * row = &file[data_offset + row_bytes * size_t(top_down ? y : h - 1 - y)];
		for (int x = 0; x < w; x++) {
			unsigned char * out = &rgba[(size_t(y) * w + x) * 4];
			if (bits == 8)       { const unsigned char * c = palette + size_t(row[x]) * 4; out[0] = c[2]; out[1] = c[1]; out[2] = c[0]; out[3] = 255; }
			else if (bits == 24) { out[0] = row[3 * x + 2]; out[1] = row[3 * x + 1]; out[2] = row[3 * x]; out[3] = 255; }
			else {
				uint32_t v = le32(&row[4 * x]);
				for (int c = 0; c < 4; c++) out[c] = extract(v, mask[c]);
			}
		}
	}
	bool any_alpha = false;
	for (size_t i = 3; i < rgba.size(); i += 4) any_alpha |= rgba[i] != 0;
	if (!any_alpha) for (size_t i = 3; i < rgba.size(); i += 4) rgba[i] = 255;
	return true;
}

namespace {
// One 4x4 block of BC1 colour data -> 16 RGBA texels. `opaque_only`: BC2 / BC3 colour blocks never use the
// 3-colour + transparent mode. Endpoints expand by bit replication; the two interpolated colours are the
// exact thirds rounded to nearest (the D3D definition; GPUs differ from it by at most one step).
void decode_colour_block(const unsigned char * block, bool opaque_only, unsigned char out[16][4]) {
	uint16_t c0 = le16(block), c1 = le16(block + 2);
	int colour[4][4];
	auto expand = [](uint16_t c, int rgb[4]) {
		int r = (c >> 11) & 31, g = (c >> 5) & 63, b = c & 31;
		rgb[0] = (r << 3) | (r >> 2); rgb[1] = (g << 2) | (g >> 4); rgb[2] = (b << 3) | (b >> 2); rgb[3] = 255;
	};
	expand(c0, colour[0]);
	expand(c1, colour[1]);
	if (c0 > c1 || opaque_only) {
		for (int c = 0; c < 3; c++) {
			colour[2][c] = (2 * colour[0][c] + colour[1][c] + 1) / 3;
			colour[3][c] = (colour[0][c] + 2 * colour[1][c] + 1) / 3;
		}
		colour[2][3] = colour[3][3] = 255;
	} else {
		for (int c = 0; c < 3; c++) { colour[2][c] = (colour[0][c] + colour[1][c]) / 2; colour[3][c] = 0; }
		colour[2][3] = 255; colour[3][3] = 0;
	}
	uint32_t indices = le32(block + 4);
	for (int i = 0; i < 16; i++) {
		const int * c = colour[(indices >> (2 * i)) & 3];
		for (int k = 0; k < 4; k++) out[i][k] = (unsigned char)c[k];
	}
}
}

void BlockCompression::decode_bc1_block(const unsigned char block[8], unsigned char rgba[16][4]) {
	decode_colour_block(block, false, rgba);
}

bool ImageDecoders::decode_dds(const std::vector<unsigned char> & file, int & width, int & height, std::vector<std::vector<unsigned char>> & mip_levels) {
	if (file.size() < 128 || memcmp(file.data(), "DDS ", 4) != 0) return false;
	uint32_t h = le32(&file[12]), w = le32(&file[16]), mip_count = le32(&file[28]);
	const unsigned char * four_cc = &file[84];
	if (memcmp(four_cc, "DXT", 3) != 0) return false;
	int kind = four_cc[3] == '1' ? 1 : four_cc[3] == '3' ? 3 : four_cc[3] == '5' ? 5 : 0;
	if (!kind || w == 0 || h == 0 || w > (1u << 15) || h > (1u << 15)) return false;
	if (mip_count == 0) mip_count = 1;
	size_t block_bytes = kind == 1 ? 8 : 16;

	width = int(w); height = int(h);
	mip_levels.clear();
	size_t pos = 128;
	for (uint32_t level = 0; level < mip_count; level++) {
		int lw = std::max(1, int(w >> level)), lh = std::max(1, int(h >> level));
		int bw = (lw + 3) / 4, bh = (lh + 3) / 4;
		if (pos + size_t(bw) * bh * block_bytes > file.size()) break; // fewer levels than announced: keep what is there
		std::vector<unsigned char> rgba(size_t(lw) * lh * 4);
		for (int by = 0; by < bh; by++) {
			for (int bx = 0; bx < bw; bx++, pos += block_bytes) {
				const unsigned char * block = &file[pos];
				unsigned char texel[16][4];
				decode_colour_block(block + (kind == 1 ? 0 : 8), kind != 1, texel);
				if (kind == 3) { // explicit 4-bit alpha
					for (int i = 0; i < 16; i++) { int a = (block[i / 2] >> (4 * (i & 1))) & 15; texel[i][3] = (unsigned char)(a * 17); }
				} else if (kind == 5) { // two alpha endpoints + 3-bit indices
					int a[8]; a[0] = block[0]; a[1] = block[1];
					if (a[0] > a[1]) for (int k = 1; k < 7; k++) a[k + 1] = ((7 - k) * a[0] + k * a[1] + 3) / 7;
					else { for (int k = 1; k < 5; k++) a[k + 1] = ((5 - k) * a[0] + k * a[1] + 2) / 5; a[6] = 0; a[7] = 255; }
					uint64_t bits = 0; for (int k = 0; k < 6; k++) bits |= uint64_t(block[2 + k]) << (8 * k);
					for (int i = 0; i < 16; i++) texel[i][3] = (unsigned char)a[(bits >> (3 * i)) & 7];
				}
				for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
					int x = bx * 4 + i, y = by * 4 + j;
					if (x < lw && y < lh) memcpy(&rgba[(size_t(y) * lw + x) * 4], texel[j * 4 + i], 4);
				}
			}
		}
		mip_levels.push_back(std::move(rgba));
		if (lw == 1 && lh == 1) break;
	}
	return !mip_levels.empty();
}
