// Texture files -> linear-light RGBA8 mip chains.
// Pipeline as in the reference (Src/Assets/TextureLoader.cpp:129-206): decode to 8-bit RGBA,
// sRGB -> linear in float, build each mip by box-filtering the previous level
// (Src/Math/Mipmap.cpp:72-152), quantise back to 8 bits by truncation. The reference then
// BC1-compresses power-of-two textures for the texture unit; CDNA has none, so the RGBA8
// levels are what the shade kernel filters.
// Decoders: TGA (every type and depth stb_image reads) and binary PPM (P6) here, PNG / BMP /
// DXT-compressed DDS in ImageDecoders.cpp, JPEG in JPEGDecoder.cpp; stb_image is not linked (PSD, GIF,
// PIC and Radiance-HDR textures are not read: such a texture gets the reference's 1x1 fallback).
#include "Scene.h"
#include "ImageDecoders.h"
#include "BlockCompression.h"
#include "Parser.h"

#include <cstdio>
#include <cstring>

namespace {

bool read_file(const std::string & filename, std::vector<unsigned char> & bytes) {
	FILE * f = fopen(filename.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long size = ftell(f);
	fseek(f, 0, SEEK_SET);
	bytes.resize(size_t(size));
	bool ok = fread(bytes.data(), 1, size_t(size), f) == size_t(size);
	fclose(f);
	return ok;
}

// Truevision TGA, with the coverage and the conventions of the reference's stb_image (stb_image.h
// stbi__tga_load): types 1 / 2 / 3 and their run-length forms 9 / 10 / 11; 8-bit grey, 16-bit grey + alpha,
// 15 / 16-bit RGB (5 bits per channel, the top bit is not alpha), 24 / 32-bit BGR(A); colour-mapped images with
// 8- or 16-bit indices into a 15 / 16 / 24 / 32-bit palette (an out-of-range index reads entry 0); bottom-up
// unless descriptor bit 5 is set. The right-to-left bit (4) is ignored, as stb_image ignores it.
bool decode_tga(const std::vector<unsigned char> & file, int & width, int & height, std::vector<unsigned char> & rgba) {
	if (file.size() < 18) return false;
	int id_length     = file[0];
	int indexed       = file[1];
	int image_type    = file[2];
	int palette_start = file[3] | (file[4] << 8);
	int palette_len   = file[5] | (file[6] << 8);
	int palette_bits  = file[7];
	width  = file[12] | (file[13] << 8);
	height = file[14] | (file[15] << 8);
	int bpp        = file[16];
	int descriptor = file[17];
	if (width <= 0 || height <= 0 || indexed > 1) return false;

	bool rle = image_type >= 8;
	if (rle) image_type -= 8;
	if (image_type < 1 || image_type > 3 || (image_type == 1) != (indexed == 1)) return false;
	if (indexed && !(bpp == 8 || bpp == 16)) return false;

	// components per decoded pixel: 1 grey, 2 grey + alpha, 3 rgb, 4 rgba; `packed16`: 5-5-5 in two bytes
	bool packed16 = false;
	auto components_of = [&packed16](int bits, bool grey) {
		switch (bits) {
			case 8:  return 1;
			case 16: if (grey) return 2; // fall through: 16-bit colour is 5-5-5
			case 15: packed16 = true; return 3;
			case 24: return 3;
			case 32: return 4;
			default: return 0;
		}
	};
	int comp = indexed ? components_of(palette_bits, false) : components_of(bpp, image_type == 3);
	if (!comp || (!indexed && image_type == 3 && comp > 2) || (!indexed && image_type == 2 && comp < 3)) return false;

	size_t pos = 18 + size_t(id_length);
	auto read_pixel = [&](unsigned char out[4]) { // one palette entry or one direct pixel, in file order
		if (packed16) {
			if (pos + 2 > file.size()) return false;
			unsigned px = file[pos] | (file[pos + 1] << 8); pos += 2;
			out[0] = (unsigned char)((((px >> 10) & 31) * 255) / 31); // already r, g, b
			out[1] = (unsigned char)((((px >> 5) & 31) * 255) / 31);
			out[2] = (unsigned char)(((px & 31) * 255) / 31);
			return true;
		}
		if (pos + size_t(comp) > file.size()) return false;
		for (int j = 0; j < comp; j++) out[j] = file[pos++];
		return true;
	};

	std::vector<unsigned char> palette;
	if (indexed) {
		pos += size_t(palette_start);
		palette.resize(size_t(palette_len) * comp);
		for (int i = 0; i < palette_len; i++) {
			unsigned char entry[4];
			if (!read_pixel(entry)) return false;
			memcpy(&palette[size_t(i) * comp], entry, size_t(comp));
		}
	}

	size_t pixel_count = size_t(width) * height;
	std::vector<unsigned char> raw(pixel_count * comp);
	unsigned char current[4] = { 0, 0, 0, 0 };
	int  run = 0;
	bool repeating = false;
	for (size_t i = 0; i < pixel_count; i++) {
		bool read_next = true;
		if (rle) {
			if (run == 0) {
				if (pos >= file.size()) return false;
				int command = file[pos++];
				run = 1 + (command & 127);
				repeating = (command >> 7) != 0;
			} else if (repeating) {
				read_next = false;
			}
		}
		if (read_next) {
			if (indexed) {
				size_t index_bytes = bpp == 8 ? 1 : 2;
				if (pos + index_bytes > file.size()) return false;
				int index = bpp == 8 ? file[pos] : (file[pos] | (file[pos + 1] << 8));
				pos += index_bytes;
				if (index >= palette_len) index = 0;
				if (palette_len == 0) return false;
				memcpy(current, &palette[size_t(index) * comp], size_t(comp));
			} else if (!read_pixel(current)) {
				return false;
			}
		}
		memcpy(&raw[i * comp], current, size_t(comp));
		run--;
	}

	bool top_down = (descriptor & 0x20) != 0;
	bool bgr = comp >= 3 && !packed16; // 24 / 32-bit data and palettes are stored blue first
	rgba.resize(pixel_count * 4);
	for (int y = 0; y < height; y++) {
		int src_y = top_down ? y : height - 1 - y; // row 0 of the output is the top of the image
		for (int x = 0; x < width; x++) {
			const unsigned char * s = &raw[(size_t(src_y) * width + x) * comp];
			unsigned char * d = &rgba[(size_t(y) * width + x) * 4];
			if (comp <= 2) { d[0] = d[1] = d[2] = s[0]; d[3] = comp == 2 ? s[1] : 255; }
			else { d[0] = bgr ? s[2] : s[0]; d[1] = s[1]; d[2] = bgr ? s[0] : s[2]; d[3] = comp == 4 ? s[3] : 255; }
		}
	}
	return true;
}

// Binary PNM: P6 (RGB) and P5 (grey), 8 bits per sample, '#' comments in the header (what stb_image reads)
bool decode_ppm(const std::vector<unsigned char> & file, int & width, int & height, std::vector<unsigned char> & rgba) {
	if (file.size() < 2 || file[0] != 'P' || (file[1] != '6' && file[1] != '5')) return false;
	int channels = file[1] == '6' ? 3 : 1;
	size_t pos = 2;
	auto next_int = [&](int & out) {
		while (pos < file.size()) {
			if (file[pos] == '#') { while (pos < file.size() && file[pos] != '\n' && file[pos] != '\r') pos++; }
			else if (file[pos] == ' ' || file[pos] == '\n' || file[pos] == '\r' || file[pos] == '\t' || file[pos] == '\f' || file[pos] == '\v') pos++;
			else break;
		}
		if (pos >= file.size() || !is_digit(char(file[pos]))) return false;
		out = 0;
		while (pos < file.size() && is_digit(char(file[pos])) && out < (1 << 24)) out = out * 10 + (file[pos++] - '0');
		return true;
	};
	int maxval = 0;
	if (!next_int(width) || !next_int(height) || !next_int(maxval) || maxval > 255 || maxval <= 0) return false;
	if (width <= 0 || height <= 0 || width > (1 << 15) || height > (1 << 15)) return false;
	pos++; // single whitespace after maxval
	size_t pixel_count = size_t(width) * height;
	if (file.size() < pos + pixel_count * channels) return false;
	rgba.resize(pixel_count * 4);
	for (size_t i = 0; i < pixel_count; i++) {
		const unsigned char * s = &file[pos + i * channels];
		rgba[4 * i + 0] = s[0];
		rgba[4 * i + 1] = s[channels == 3 ? 1 : 0];
		rgba[4 * i + 2] = s[channels == 3 ? 2 : 0];
		rgba[4 * i + 3] = 255;
	}
	return true;
}

// The three mip filters of the reference (Src/Math/Mipmap.cpp:14-52): each is a window function and its
// half-width in destination texels.
float sinc(float x) { // Math.h:136-142
	if (fabsf(x) < 0.0001f) return 1.0f + x * x * (-1.0f / 6.0f + x * x * 1.0f / 120.0f);
	return sinf(x) / x;
}
float bessel_0(float x) { // Math.h:145-162: power series until the term no longer changes the sum
	float xh = 0.5f * x, sum = 1.0f, pow = 1.0f, ds = 1.0f, k = 0.0f;
	while (ds > sum * 1e-6f) {
		k += 1.0f;
		pow = pow * (xh / k);
		ds  = pow * pow;
		sum = sum + ds;
	}
	return sum;
}
float filter_width(MipmapFilterType type) { return type == MipmapFilterType::BOX ? 0.5f : (type == MipmapFilterType::LANCZOS ? 3.0f : 7.0f); }
float filter_eval(MipmapFilterType type, float x) {
	switch (type) {
		case MipmapFilterType::BOX:     return fabsf(x) <= 0.5f ? 1.0f : 0.0f;
		case MipmapFilterType::LANCZOS: return fabsf(x) < 3.0f ? sinc(PI * x) * sinc(PI * x / 3.0f) : 0.0f;
		default: { // Kaiser window, alpha 4, width 7
			float t = x / 7.0f, t2 = t * t;
			return t2 < 1.0f ? sinc(PI * x * 1.0f) * bessel_0(4.0f * sqrtf(1.0f - t2)) / bessel_0(4.0f) : 0.0f;
		}
	}
}
} // namespace

// Separable downsampling with the kernel construction of the reference mip generator (Mipmap.cpp:54-152):
// the filter integrated over each source texel (32 sub-samples), normalised, applied along x into a
// transposed temporary and then along y; source indices clamp at the borders.
void TextureLoader::downsample(MipmapFilterType filter, int w_src, int h_src, int w_dst, int h_dst, const Vector4 * src, Vector4 * dst, std::vector<Vector4> & temp) {
	auto make_kernel = [filter](int n_src, int n_dst, std::vector<float> & kernel, float & half_width, float & inv_scale) {
		float scale = float(n_dst) / float(n_src);
		inv_scale  = 1.0f / scale;
		half_width = filter_width(filter) * inv_scale;
		int window = int(ceilf(half_width * 2.0f)) + 1;
		kernel.assign(window, 0.0f);
		float sum = 0.0f;
		for (int i = 0; i < window; i++) {
			float x = float(i - window / 2);
			float acc = 0.0f, sample = 0.5f;
			for (int s = 0; s < 32; s++, sample += 1.0f) {
				float p = (x + sample * (1.0f / 32.0f)) * scale;
				acc += filter_eval(filter, p);
			}
			kernel[i] = acc * (1.0f / 32.0f);
			sum += kernel[i];
		}
		for (float & k : kernel) k /= sum;
	};
	std::vector<float> kx, ky;
	float fwx, fwy, isx, isy;
	make_kernel(w_src, w_dst, kx, fwx, isx);
	make_kernel(h_src, h_dst, ky, fwy, isy);

	temp.resize(size_t(w_dst) * h_src);
	for (int y = 0; y < h_src; y++) {
		for (int x = 0; x < w_dst; x++) {
			int left = int(floorf((float(x) + 0.5f) * isx - fwx));
			Vector4 sum;
			for (size_t i = 0; i < kx.size(); i++) {
				const Vector4 & t = src[Math::clamp(left + int(i), 0, w_src - 1) + size_t(y) * w_src];
				sum.x += kx[i] * t.x; sum.y += kx[i] * t.y; sum.z += kx[i] * t.z; sum.w += kx[i] * t.w;
			}
			temp[size_t(x) * h_src + y] = sum;
		}
	}
	for (int x = 0; x < w_dst; x++) {
		for (int y = 0; y < h_dst; y++) {
			int top = int(floorf((float(y) + 0.5f) * isy - fwy));
			Vector4 sum;
			for (size_t i = 0; i < ky.size(); i++) {
				const Vector4 & t = temp[size_t(x) * h_src + Math::clamp(top + int(i), 0, h_src - 1)];
				sum.x += ky[i] * t.x; sum.y += ky[i] * t.y; sum.z += ky[i] * t.z; sum.w += ky[i] * t.w;
			}
			dst[x + size_t(y) * w_dst] = sum;
		}
	}
}

namespace {
unsigned char quantise(float v) { return (unsigned char)(Math::clamp(v * 255.0f, 0.0f, 255.0f)); }

} // namespace

bool TextureLoader::load(const std::string & filename, Texture * texture) {
	std::vector<unsigned char> file, rgba8;
	if (!read_file(filename, file)) return false;

	int width = 0, height = 0;

	// DDS files carry their own mip chain of block-compressed levels, used as stored -- no gamma
	// conversion, no re-filtering (reference: TextureLoader.cpp:19-106)
	std::vector<std::vector<unsigned char>> dds_levels;
	if (ImageDecoders::decode_dds(file, width, height, dds_levels)) {
		// The reference walks the chain by halving the BLOCK counts and stops when one reaches zero (:92-104): levels
		// narrower or lower than one full block of the previous halving -- the 2x2 and 1x1 tail -- are never used
		size_t kept = 0;
		for (int bw = (width + 3) / 4, bh = (height + 3) / 4; kept < dds_levels.size() && bw > 0 && bh > 0; bw /= 2, bh /= 2) kept++;
		dds_levels.resize(std::max<size_t>(kept, 1));
		if (!gpu_config.enable_mipmapping) dds_levels.resize(1);
		texture->width  = width;
		texture->height = height;
		texture->lod_width  = (width  + 3) / 4; // the reference keeps DDS sizes in blocks (TextureLoader.cpp:54-55), and so its LOD bias
		texture->lod_height = (height + 3) / 4;
		texture->mip_offsets.clear();
		texture->texels.clear();
		for (const std::vector<unsigned char> & level : dds_levels) {
			texture->mip_offsets.push_back(texture->texels.size() / 4);
			texture->texels.insert(texture->texels.end(), level.begin(), level.end());
		}
		return true;
	}

	if (!ImageDecoders::decode_png(file, width, height, rgba8) && !ImageDecoders::decode_jpeg(file, width, height, rgba8) &&
		!ImageDecoders::decode_bmp(file, width, height, rgba8) &&
		!decode_ppm(file, width, height, rgba8) && !decode_tga(file, width, height, rgba8)) return false;

	// Mip level sizes: halve each dimension down to 1 (reference mip_count, TextureLoader.cpp:108-127)
	std::vector<std::pair<int, int>> level_size;
	size_t total = 0;
	for (int w = width, h = height;;) {
		level_size.emplace_back(w, h);
		total += size_t(w) * h;
		if (!gpu_config.enable_mipmapping || (w == 1 && h == 1)) break;
		if (w > 1) w /= 2;
		if (h > 1) h /= 2;
	}

	std::vector<Vector4> linear(total);
	for (size_t i = 0; i < size_t(width) * height; i++) {
		linear[i] = Vector4(
			Math::gamma_to_linear(float(rgba8[4 * i + 0]) / 255.0f),
			Math::gamma_to_linear(float(rgba8[4 * i + 1]) / 255.0f),
			Math::gamma_to_linear(float(rgba8[4 * i + 2]) / 255.0f),
			Math::gamma_to_linear(float(rgba8[4 * i + 3]) / 255.0f));
	}

	texture->width  = width;
	texture->height = height;
	texture->mip_offsets.clear();
	size_t offset = 0;
	std::vector<Vector4> temp;
	for (size_t l = 0; l < level_size.size(); l++) {
		texture->mip_offsets.push_back(offset);
		if (l + 1 < level_size.size()) {
			size_t next = offset + size_t(level_size[l].first) * level_size[l].second;
			if (cpu_config.mipmap_filter == MipmapFilterType::BOX) { // the box filter can work from the previous level ...
				downsample(MipmapFilterType::BOX, level_size[l].first, level_size[l].second, level_size[l + 1].first, level_size[l + 1].second, &linear[offset], &linear[next], temp);
			} else {                                                  // ... the wider ones resample the original (TextureLoader.cpp:170-178)
				downsample(cpu_config.mipmap_filter, width, height, level_size[l + 1].first, level_size[l + 1].second, &linear[0], &linear[next], temp);
			}
			offset = next;
		}
	}

	texture->texels.resize(total * 4);
	for (size_t i = 0; i < total; i++) {
		texture->texels[4 * i + 0] = quantise(linear[i].x);
		texture->texels[4 * i + 1] = quantise(linear[i].y);
		texture->texels[4 * i + 2] = quantise(linear[i].z);
		texture->texels[4 * i + 3] = quantise(linear[i].w);
	}

	// Block compression as the reference applies it to power-of-two textures (TextureLoader.cpp:208-262): each
	// level is BC1-encoded in 4x4 blocks and -- there being no texture unit to decode it later -- decoded
	// again right away. The compressed chain ends at the level that is one block in size, and the LOD bias
	// is derived from the block counts (see Texture::lod_width).
	auto is_power_of_two = [](int x) { return x > 0 && (x & (x - 1)) == 0; };
	if (cpu_config.enable_block_compression && is_power_of_two(width) && is_power_of_two(height)) {
		int blocks_w = (width + 3) / 4, blocks_h = (height + 3) / 4;
		size_t kept_levels = 0;
		for (int w = blocks_w, h = blocks_h;;) {
			kept_levels++;
			if (!gpu_config.enable_mipmapping || (w == 1 && h == 1)) break;
			if (w > 1) w /= 2;
			if (h > 1) h /= 2;
		}
		if (kept_levels < texture->mip_offsets.size()) {
			texture->texels.resize(texture->mip_offsets[kept_levels] * 4);
			texture->mip_offsets.resize(kept_levels);
		}
		for (size_t l = 0; l < texture->mip_offsets.size(); l++) {
			BlockCompression::quantise_level_bc1(&texture->texels[texture->mip_offsets[l] * 4], std::max(width >> l, 1), std::max(height >> l, 1), &texture->bc1_blocks);
		}
		texture->lod_width  = blocks_w;
		texture->lod_height = blocks_h;
	}
	return true;
}
