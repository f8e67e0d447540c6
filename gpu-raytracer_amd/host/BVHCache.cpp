#include "BVHCache.h"
#include "Config.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <sys/stat.h>
#include <zlib.h>

BVHType BVHCache::underlying_bvh_type() {
	return cpu_config.bvh_type == BVHType::SBVH ? BVHType::SBVH : BVHType::BVH;
}

namespace {
struct File {
	FILE * f;
	File(const std::string & name, const char * mode) : f(fopen(name.c_str(), mode)) { }
	~File() { if (f) fclose(f); }
};

bool modification_time(const std::string & filename, timespec * out) {
	struct stat st;
	if (stat(filename.c_str(), &st) != 0) return false;
	*out = st.st_mtim;
	return true;
}

// Pulls a given number of bytes at a time out of one raw deflate stream that continues across calls.
struct Inflater {
	FILE *   file;
	z_stream z;
	bool     ok;
	unsigned char buffer[64 * 1024];

	explicit Inflater(FILE * file) : file(file) {
		memset(&z, 0, sizeof(z));
		ok = inflateInit2(&z, -15) == Z_OK;
	}
	~Inflater() { inflateEnd(&z); }

	bool read(void * dst, size_t bytes) {
		if (!ok) return false;
		z.next_out = (Bytef *)dst;
		size_t remaining = bytes;
		while (remaining > 0) {
			if (z.avail_in == 0) {
				z.next_in  = buffer;
				z.avail_in = uInt(fread(buffer, 1, sizeof(buffer), file));
				// At the end of the file zlib may still hold output (a match that spans the last read, the
				// end-of-block code in the final byte): keep calling inflate() and let IT report a truncated stream.
			}
			uInt chunk = uInt(remaining < (1u << 30) ? remaining : (1u << 30));
			uInt had_in = z.avail_in;
			z.avail_out = chunk;
			int status = inflate(&z, Z_NO_FLUSH);
			size_t produced = chunk - z.avail_out;
			remaining -= produced;
			if (status == Z_STREAM_END) break;
			if (status != Z_OK) return false;                                     // Z_BUF_ERROR: no progress possible = truncated
			if (produced == 0 && had_in == 0 && z.avail_in == 0) return false;   // nothing in, nothing out
		}
		return remaining == 0;
	}
};

struct Deflater {
	FILE *   file;
	z_stream z;
	bool     ok;
	unsigned char buffer[64 * 1024];

	explicit Deflater(FILE * file) : file(file) {
		memset(&z, 0, sizeof(z));
		ok = deflateInit2(&z, 6, Z_DEFLATED, -15, 9, Z_DEFAULT_STRATEGY) == Z_OK;
	}
	~Deflater() { deflateEnd(&z); }

	bool write(const void * src, size_t bytes, bool last) {
		if (!ok) return false;
		z.next_in = (Bytef *)src;
		size_t remaining = bytes;
		while (true) {
			uInt chunk = uInt(remaining < (1u << 30) ? remaining : (1u << 30));
			z.avail_in = chunk;
			bool finishing = last && chunk == remaining;
			int status;
			do {
				z.next_out  = buffer;
				z.avail_out = sizeof(buffer);
				status = deflate(&z, finishing ? Z_FINISH : Z_NO_FLUSH);
				if (status == Z_STREAM_ERROR) return false;
				size_t produced = sizeof(buffer) - z.avail_out;
				if (produced && fwrite(buffer, 1, produced, file) != produced) return false;
			} while (z.avail_out == 0 || (finishing && status != Z_STREAM_END));
			remaining -= chunk;
			if (remaining == 0) return true;
		}
	}
};
}

bool BVHCache::try_to_load(const std::string & mesh_filename, const std::string & bvh_filename, std::vector<Triangle> * triangles, BVH2 * bvh) {
	if (cpu_config.bvh_force_rebuild) return false;
	timespec mesh_time, cache_time;
	if (!modification_time(mesh_filename, &mesh_time) || !modification_time(bvh_filename, &cache_time)) return false;
	bool cache_is_older = cache_time.tv_sec < mesh_time.tv_sec || (cache_time.tv_sec == mesh_time.tv_sec && cache_time.tv_nsec < mesh_time.tv_nsec);
	if (cache_is_older) return false;

	File file(bvh_filename, "rb");
	if (!file.f) {
		fprintf(stderr, "WARNING: Failed to open BVH file '%s'!\n", bvh_filename.c_str());
		return false;
	}
	FileHeader header = { };
	if (fread(&header, sizeof(header), 1, file.f) != 1) {
		fprintf(stderr, "WARNING: Failed to read header of BVH file '%s'!\n", bvh_filename.c_str());
		return false;
	}
	if (memcmp(header.filetype_identifier, "BVH", 4) != 0 || header.filetype_version != FILETYPE_VERSION) return false;
	if (header.underlying_bvh_type != char(underlying_bvh_type()) ||
		(header.bvh_is_optimized != 0) != cpu_config.enable_bvh_optimization || header.bvh_is_optimized > 1 ||
		header.sah_cost_node       != cpu_config.sah_cost_node ||
		header.sah_cost_leaf       != cpu_config.sah_cost_leaf) {
		return false; // built with other settings: rebuild (BVHLoader.cpp:156-164)
	}
	if (header.num_triangles < 0 || header.num_nodes < 0 || header.num_indices < 0) return false;
	{	// deflate cannot expand by more than ~1032 : 1, so counts far beyond the file's size are a damaged header
		struct stat st;
		if (stat(bvh_filename.c_str(), &st) != 0) return false;
		double claimed = double(header.num_triangles) * sizeof(Triangle) + double(header.num_nodes) * sizeof(BVHNode2) + double(header.num_indices) * sizeof(int);
		if (claimed > double(st.st_size) * 1040.0 + 4096.0) return false;
	}

	std::vector<Triangle> loaded_triangles(header.num_triangles);
	BVH2 loaded;
	loaded.nodes  .resize(header.num_nodes);
	loaded.indices.resize(header.num_indices);

	Inflater inflater(file.f);
	bool success =
		inflater.read((void *)loaded_triangles.data(), loaded_triangles.size() * sizeof(Triangle)) &&
		inflater.read((void *)loaded.nodes    .data(), loaded.nodes    .size() * sizeof(BVHNode2)) &&
		inflater.read((void *)loaded.indices  .data(), loaded.indices  .size() * sizeof(int));
	if (!success) {
		fprintf(stderr, "WARNING: BVH file '%s' is truncated or corrupt, rebuilding\n", bvh_filename.c_str());
		return false;
	}
	// A cache is trusted for its content but not for memory safety: every index must stay in range.
	for (int index : loaded.indices) if (index < 0 || index >= header.num_triangles) return false;
	// ... and the nodes must form a tree: walk it from the root with a visited bitmap. Every node may be reached
	// at most once (no cycles, no shared subtrees), child pairs must lie inside the array, every box reached must
	// be finite. The order of the nodes in the array is NOT constrained: the builders emit children after their
	// parent, but BVHOptimizer re-inserts subtrees into freed slots in front of it (a third of Sponza's inner
	// nodes after -O), and such caches -- ours and the reference's -- have to load.
	if (loaded.nodes.size() < 2 || loaded.indices.empty()) return false;
	{
		std::vector<bool> visited(loaded.nodes.size(), false);
		std::vector<int> stack; stack.push_back(0);
		visited[0] = true;
		while (!stack.empty()) {
			const BVHNode2 & node = loaded.nodes[size_t(stack.back())]; stack.pop_back();
			const float * box = &node.aabb.min.x;
			for (int k = 0; k < 6; k++) if (!std::isfinite(box[k])) return false;
			if (node.is_leaf()) {
				if (node.count != 1) return false; // caches hold the builders' raw trees: one reference per leaf
				if (node.first < 0 || size_t(node.first) + node.count > loaded.indices.size()) return false;
				continue;
			}
			if (node.left < 2 || size_t(node.left) + 1 >= loaded.nodes.size()) return false; // slots 0 and 1: the root and its unused sibling
			for (int child = node.left; child <= node.left + 1; child++) {
				if (visited[size_t(child)]) return false;
				visited[size_t(child)] = true;
				stack.push_back(child);
			}
		}
	}
	*triangles = std::move(loaded_triangles);
	*bvh       = std::move(loaded);
	return true;
}

bool BVHCache::save(const std::string & bvh_filename, const std::vector<Triangle> & triangles, const BVH2 & bvh) {
	std::string tmp_filename = bvh_filename + ".tmp"; // renamed into place, so that readers never see half a file
	{
		File file(tmp_filename, "wb");
		if (!file.f) {
			fprintf(stderr, "WARNING: Failed to open BVH file '%s' for writing!\n", bvh_filename.c_str());
			return false;
		}
		FileHeader header = { };
		memcpy(header.filetype_identifier, "BVH", 4);
		header.filetype_version    = FILETYPE_VERSION;
		header.underlying_bvh_type = char(underlying_bvh_type());
		header.bvh_is_optimized    = cpu_config.enable_bvh_optimization ? 1 : 0;
		header.sah_cost_node       = cpu_config.sah_cost_node;
		header.sah_cost_leaf       = cpu_config.sah_cost_leaf;
		header.num_triangles = int(triangles.size());
		header.num_nodes     = int(bvh.nodes.size());
		header.num_indices   = int(bvh.indices.size());

		Deflater deflater(file.f);
		bool success = fwrite(&header, sizeof(header), 1, file.f) == 1 &&
			deflater.write(triangles  .data(), triangles  .size() * sizeof(Triangle), false) &&
			deflater.write(bvh.nodes  .data(), bvh.nodes  .size() * sizeof(BVHNode2), false) &&
			deflater.write(bvh.indices.data(), bvh.indices.size() * sizeof(int),      true);
		if (!success) {
			fprintf(stderr, "WARNING: Failed to write BVH file '%s'!\n", bvh_filename.c_str());
			remove(tmp_filename.c_str());
			return false;
		}
	}
	if (rename(tmp_filename.c_str(), bvh_filename.c_str()) != 0) {
		remove(tmp_filename.c_str());
		return false;
	}
	return true;
}
