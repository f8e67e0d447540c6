// oracle/ref/ref_stb.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The reference decodes texture files with the stb_image.h it vendors (v2.19, /root/reference/include,
// used at Src/Assets/TextureLoader.cpp:129-139 as stbi_load(..., STBI_rgb_alpha)) and block-compresses them
// with its stb_dxt.h (v1.12, :208-262). Both headers are compiled here from where they lie -- nothing is
// copied -- so that the product's own decoders (gpu-raytracer_amd/host/ImageDecoders.cpp, TextureLoader.cpp)
// can be compared with what the reference would have loaded, byte for byte.
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image.h>
#define STB_DXT_IMPLEMENTATION
#include <stb_dxt.h>

#include <cstring>

extern "C" {

// RGBA8, row 0 = top; returns 0 when stb_image cannot decode the file. dst may be NULL to query the size.
int ref_stbi_load_rgba(const char * filename, int * width, int * height, unsigned char * dst, size_t dst_bytes) {
	int channels = 0;
	unsigned char * data = stbi_load(filename, width, height, &channels, STBI_rgb_alpha);
	if (!data) return 0;
	size_t bytes = size_t(*width) * size_t(*height) * 4;
	if (dst && dst_bytes >= bytes) memcpy(dst, data, bytes);
	stbi_image_free(data);
	return 1;
}

// One 4x4 RGBA block -> 8 bytes of BC1, exactly as TextureLoader.cpp:250 calls it
void ref_stb_compress_bc1_block(const unsigned char * rgba_block, unsigned char * dst8) {
	stb_compress_dxt_block(dst8, rgba_block, 0, STB_DXT_HIGHQUAL);
}

} // extern "C"

// Radiance .hdr as Sky::load reads it (Src/Renderer/Sky.cpp:12-35): stbi_loadf(..., STBI_rgb). Returns the
// number of floats written (width * height * 3), 0 on failure; dst may be NULL to query the size.
extern "C" int ref_stbi_loadf_rgb(const char * filename, int * width, int * height, float * dst, size_t dst_floats) {
	int channels = 0;
	float * data = stbi_loadf(filename, width, height, &channels, STBI_rgb);
	if (!data) return 0;
	size_t count = size_t(*width) * size_t(*height) * 3;
	if (dst && dst_floats >= count) memcpy(dst, data, count * sizeof(float));
	stbi_image_free(data);
	return int(count);
}
