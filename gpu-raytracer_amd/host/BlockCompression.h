// BC1 texture block compression (BlockCompression.cpp) and decoding (ImageDecoders.cpp)
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

namespace BlockCompression {
	// 16 RGBA texels (row-major 4x4) -> 8 bytes, equal to stb_compress_dxt_block(dst, rgba, 0, STB_DXT_HIGHQUAL)
	void compress_bc1_block(const unsigned char rgba[64], unsigned char dst[8]);
	// 8 bytes -> 16 RGBA texels (D3D rules: bit-replicated end points, thirds rounded to nearest, 3-colour mode when c0 <= c1)
	void decode_bc1_block(const unsigned char block[8], unsigned char rgba[16][4]);
	// Replaces a width x height RGBA8 level by what survives a trip through BC1
	// ... and, if `blocks` is given, appends the ((width+3)/4) x ((height+3)/4) compressed blocks (row-major, 8 bytes each)
	void quantise_level_bc1(unsigned char * rgba, int width, int height, std::vector<unsigned char> * blocks = nullptr);
}
