#include "PMJ.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <dlfcn.h>

#include "../../include/gpu_raytracer_amd.h"

namespace {
inline unsigned mix(unsigned x) { // lowbias32 (Wellons) integer hash
	x ^= x >> 16; x *= 0x7feb352du;
	x ^= x >> 15; x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}

inline unsigned reverse_bits(unsigned x) {
	x = (x >> 16) | (x << 16);
	x = ((x & 0x00ff00ffu) << 8) | ((x & 0xff00ff00u) >> 8);
	x = ((x & 0x0f0f0f0fu) << 4) | ((x & 0xf0f0f0f0u) >> 4);
	x = ((x & 0x33333333u) << 2) | ((x & 0xccccccccu) >> 2);
	x = ((x & 0x55555555u) << 1) | ((x & 0xaaaaaaaau) >> 1);
	return x;
}

// Second Sobol' dimension via its generator matrix (Kollig & Keller 2002)
inline unsigned sobol_dim1(unsigned n) {
	unsigned r = 0;
	for (unsigned v = 1u << 31; n; n >>= 1, v ^= v >> 1) if (n & 1) r ^= v;
	return r;
}

// Exact nested-uniform (Owen) scramble in base 2: bit b is flipped by a coin that
// depends on all more significant bits of the input.
inline unsigned owen_scramble(unsigned x, unsigned seed) {
	unsigned out = x;
	for (int b = 0; b < 32; b++) {
		unsigned prefix = b == 0 ? 0u : (x >> (32 - b));
		unsigned coin = mix(prefix ^ mix(seed + 0x632be5abu * unsigned(b + 1)));
		out ^= (coin & 1u) << (31 - b);
	}
	return out;
}
}

std::vector<float> PMJ::generate(unsigned seed) {
	std::vector<float> table(size_t(RT_PMJ_NUM_SEQUENCES) * RT_PMJ_NUM_SAMPLES_PER_SEQUENCE * 2);
	for (unsigned s = 0; s < RT_PMJ_NUM_SEQUENCES; s++) {
		unsigned seed_x = mix(seed ^ mix(2 * s + 1));
		unsigned seed_y = mix(seed ^ mix(2 * s + 2));
		float * out = &table[size_t(s) * RT_PMJ_NUM_SAMPLES_PER_SEQUENCE * 2];
		for (unsigned i = 0; i < RT_PMJ_NUM_SAMPLES_PER_SEQUENCE; i++) {
			unsigned x = owen_scramble(reverse_bits(i), seed_x);
			unsigned y = owen_scramble(sobol_dim1(i),   seed_y);
			out[2 * i + 0] = float(x >> 8) * (1.0f / 16777216.0f); // 24 bits: exact, < 1
			out[2 * i + 1] = float(y >> 8) * (1.0f / 16777216.0f);
		}
	}
	return table;
}

std::string BlueNoise::asset_directory() {
	if (const char * env = getenv("GRT_ASSET_DIR")) return std::string(env) + "/";
	Dl_info info;
	if (dladdr((void *)&BlueNoise::asset_directory, &info) && info.dli_fname) {
		std::string lib(info.dli_fname);
		size_t slash = lib.find_last_of('/');
		std::string dir = slash == std::string::npos ? "." : lib.substr(0, slash);
		return dir + "/../../assets/"; // <repo>/gpu-raytracer_amd/host/libgrt_host.so -> <repo>/assets
	}
	return "assets/";
}

std::vector<unsigned char> BlueNoise::load() {
	const size_t size = size_t(RT_BLUE_NOISE_NUM_TEXTURES) * RT_BLUE_NOISE_TEXTURE_DIM * RT_BLUE_NOISE_TEXTURE_DIM * 2;
	std::string candidates[2] = { asset_directory() + "blue_noise_16x128x128_rg8.bin", "assets/blue_noise_16x128x128_rg8.bin" };
	for (const std::string & path : candidates) {
		FILE * f = fopen(path.c_str(), "rb");
		if (!f) continue;
		std::vector<unsigned char> data(size);
		size_t got = fread(data.data(), 1, size, f);
		fclose(f);
		if (got == size) return data;
	}
	throw std::runtime_error("blue noise asset 'blue_noise_16x128x128_rg8.bin' not found (set GRT_ASSET_DIR)");
}
