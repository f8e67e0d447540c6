// Texture file decoders (ImageDecoders.cpp, TextureLoader.cpp); all return 8-bit RGBA, row 0 = top.
#pragma once
#include <vector>

namespace ImageDecoders {
	bool decode_png(const std::vector<unsigned char> & file, int & width, int & height, std::vector<unsigned char> & rgba);
	bool decode_jpeg(const std::vector<unsigned char> & file, int & width, int & height, std::vector<unsigned char> & rgba);
	bool decode_bmp(const std::vector<unsigned char> & file, int & width, int & height, std::vector<unsigned char> & rgba);
	// DXT1 / DXT3 / DXT5: one RGBA image per mip level stored in the file
	bool decode_dds(const std::vector<unsigned char> & file, int & width, int & height, std::vector<std::vector<unsigned char>> & mip_levels);
}
