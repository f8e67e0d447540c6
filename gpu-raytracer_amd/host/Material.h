// Scene-level material / medium / texture descriptions
// (reference: Src/Renderer/Material.h, Medium.h, Texture.h).
#pragma once
#include <string>
#include <vector>

#include "Handle.h"

struct Texture {
	std::string name;
	int width = 0, height = 0;
	// Linear-light RGBA8 texels for every mip level, level 0 first.
	std::vector<unsigned char> texels;
	std::vector<size_t> mip_offsets; // in texels (4 bytes each)
	// Size that enters the LOD bias; 0 = width / height. Block counts for a BC1-compressed texture, as in
	// the reference (TextureLoader.cpp:256-258 overwrite width / height, Integrator.cpp:95 reads them).
	int lod_width = 0, lod_height = 0;
	// BC1 blocks (8 bytes per 4x4 texels, rows of blocks, all kept levels back to back) of a block-compressed
	// texture: what the device stores and decodes in its texel fetch. `texels` then holds the decoded levels --
	// the same values, for the host-side consumers (exporters, the parity checker).
	std::vector<unsigned char> bc1_blocks;

	int mip_levels() const { return int(mip_offsets.size()); }
};

struct Medium {
	std::string name;

	Vector3 C   = 1.0f; // multi-scatter albedo
	Vector3 mfp = 1.0f; // mean free path
	float   g   = 0.0f; // Henyey-Greenstein mean cosine

	// Van de Hulst albedo inversion, both directions (reference: Renderer/Medium.h:18-37)
	void from_sigmas(const Vector3 & sigma_a, const Vector3 & sigma_s) {
		Vector3 sigma_t = sigma_a + sigma_s;
		Vector3 alpha = sigma_s / sigma_t;
		Vector3 s = Vector3::apply((1.0f - alpha) / (1.0f - alpha * g), sqrtf);
		C   = (1.0f - s) * (1.0f - 0.139f * s) / (1.0f + 1.17f * s);
		mfp = 1.0f / sigma_t;
	}
	void to_sigmas(Vector3 & sigma_a, Vector3 & sigma_s) const {
		Vector3 s = 4.09712f + 4.20863f * C - Vector3::apply(9.59217f + 41.6808f * C + 17.7126f * C * C, sqrtf);
		Vector3 alpha = (1.0f - s * s) / (1.0f - Math::clamp(g, -0.999f, 0.999f) * s * s);
		Vector3 sigma_t = 1.0f / Vector3::max(mfp, 1e-6f);
		sigma_s = alpha * sigma_t;
		sigma_a = sigma_t - sigma_s;
	}
};

struct Material {
	std::string name;

	enum struct Type : char { LIGHT, DIFFUSE, PLASTIC, DIELECTRIC, CONDUCTOR };
	Type type = Type::DIFFUSE;

	Vector3 emission;

	Vector3         diffuse = Vector3(1.0f, 1.0f, 1.0f);
	Handle<Texture> texture_handle;

	Handle<Medium> medium_handle;
	float          index_of_refraction = 1.33f;

	Vector3 eta = Vector3(1.33f);
	Vector3 k   = Vector3(1.0f);

	float linear_roughness = 0.5f;

	bool is_light() const { return type == Type::LIGHT && Vector3::length_squared(emission) > 0.0f; }
};
