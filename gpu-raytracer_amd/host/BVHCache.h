// On-disk cache of a mesh's triangles and binary BVH: "<mesh file>.bvh", the reference's cache
// format (Src/Assets/BVHLoader.h:12-19, BVHLoader.cpp:19-33,196-260), so that caches written by
// either program are accepted by the other: a 28-byte header followed by ONE raw deflate stream
// (no zlib header) holding Triangle[num_triangles] (96 B each), BVHNode2[num_nodes] (32 B each),
// int[num_indices].
#pragma once
#include <string>
#include <vector>

#include "BVH.h"
#include "Config.h"

namespace BVHCache {
	constexpr const char * FILE_EXTENSION = ".bvh";
	constexpr int          FILETYPE_VERSION = 7;

	struct FileHeader {
		char filetype_identifier[4]; // "BVH\0"
		char filetype_version;

		// settings the tree was built with; a mismatch with the current ones rejects the file
		char  underlying_bvh_type;   // BVHType::BVH or BVHType::SBVH
		unsigned char bvh_is_optimized; // a bool on disk; read as a byte so that a damaged file cannot produce an invalid bool
		float sah_cost_node;
		float sah_cost_leaf;

		int num_triangles;
		int num_nodes;
		int num_indices;
	};
	static_assert(sizeof(FileHeader) == 28, "the cache header is 28 bytes on disk");

	inline std::string get_bvh_filename(const std::string & mesh_filename) { return mesh_filename + FILE_EXTENSION; }

	// The binary tree kind the cache stores for the current cpu_config.bvh_type (BVH.h:95-102)
	BVHType underlying_bvh_type();

	// False (and outputs untouched) when caching is forced off, either file is missing, the cache is
	// older than the mesh file, or its header does not match the current settings.
	bool try_to_load(const std::string & mesh_filename, const std::string & bvh_filename, std::vector<Triangle> * triangles, BVH2 * bvh);
	bool save(const std::string & bvh_filename, const std::vector<Triangle> & triangles, const BVH2 & bvh);
}
