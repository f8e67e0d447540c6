// CPU BVH construction: full-sweep SAH binary BVH (1 primitive per leaf) and its
// collapse into the 80-byte compressed wide BVH8 (CWBVH, Ylitie et al. 2017) that
// the trace kernels consume.  The node formats and every tie-breaking rule follow
// the reference so that the produced bytes are identical to its builder's
// (Src/BVH/BVH.h:11-80, Builders/SAHBuilder.cpp:13-104, Builders/BVHPartitions.cpp:6-54,
// Converters/BVH8Converter.cpp:7-335); tests/test_bvh_build.py checks this against
// the verbatim reference build in oracle/_ref.
#pragma once
#include <vector>
#include <cstdint>

#include "Triangle.h"

typedef unsigned char byte;

struct BVHNode2 {
	AABB aabb;
	union {
		int left;  // inner: index of left child, right child = left + 1
		int first; // leaf : first index into BVH::indices
	};
	unsigned count : 30; // > 0 for leaves
	unsigned axis  : 2;  // split axis of an inner node

	bool is_leaf() const { return count > 0; }
};
static_assert(sizeof(BVHNode2) == 32, "BVHNode2 is 32 bytes on the device");

struct BVHNode8 {
	Vector3 p;          // quantisation grid origin (node AABB min)
	byte e[3];          // per-axis grid scale exponent (biased, float exponent bits)
	byte imask;         // bit i set: slot i holds an inner node

	unsigned base_index_child;
	unsigned base_index_triangle;

	byte meta[8];       // leaf: (unary tri count << 5) | tri offset; inner: 0x20 | (24 + slot)

	byte quantized_min_x[8], quantized_max_x[8];
	byte quantized_min_y[8], quantized_max_y[8];
	byte quantized_min_z[8], quantized_max_z[8];
};
static_assert(sizeof(BVHNode8) == 80, "BVHNode8 is 80 bytes on the device");

// 4-wide BVH node, SoA over the (up to) four children (Src/BVH/BVH.h:25-57): count > 0 leaf with
// `count` primitives from `index`, count == 0 inner node `index`, count == -1 unused slot.
struct BVHNode4 {
	float aabb_min_x[4], aabb_min_y[4], aabb_min_z[4];
	float aabb_max_x[4], aabb_max_y[4], aabb_max_z[4];
	struct { int index, count; } index_and_count[4];

	int & get_index(int i)       { return index_and_count[i].index; }
	int   get_index(int i) const { return index_and_count[i].index; }
	int & get_count(int i)       { return index_and_count[i].count; }
	int   get_count(int i) const { return index_and_count[i].count; }
	bool  is_leaf(int i) const   { return get_count(i) > 0; }
	int   get_child_count() const { int n = 0; for (int i = 0; i < 4; i++) { if (get_count(i) == -1) break; n++; } return n; }
};
static_assert(sizeof(BVHNode4) == 128, "BVHNode4 is 128 bytes on the device");

struct BVH2 {
	std::vector<int>      indices;
	std::vector<BVHNode2> nodes;
};

struct BVH4 {
	std::vector<int>      indices;
	std::vector<BVHNode4> nodes;
};

struct BVH8 {
	std::vector<int>      indices;
	std::vector<BVHNode8> nodes;
};

struct Mesh;

// Top-down SAH builder over pre-sorted index lists.
struct SAHBuilder {
	BVH2 & bvh;

	std::vector<int> sorted[3];       // primitive ids ordered by centroid along x / y / z
	std::vector<float> sweep_cost;    // left-to-right partial SAH terms
	std::vector<int>   partition_tmp;
	std::vector<char>  goes_left;

	SAHBuilder(BVH2 & bvh, size_t primitive_count);

	void build(const std::vector<Triangle> & triangles);
	void build(const std::vector<Mesh>     & meshes);
};

struct BVH8Converter {
	BVH8       & bvh8;
	const BVH2 & bvh2;

	BVH8Converter(BVH8 & bvh8, const BVH2 & bvh2) : bvh8(bvh8), bvh2(bvh2) { }
	void convert();

private:
	enum Kind : char { LEAF, INTERNAL, DISTRIBUTE };
	struct Decision {
		Kind  kind;
		char  take_left, take_right; // how many of the 7 split budget go to each child
		float cost;
	};
	std::vector<Decision> table; // [node][7]

	int  fill_cost_table(int node_index);
	void gather_children(int node_index, int budget, int children[8], int & child_count);
	void assign_octant_slots(int node_index, int children[8], int child_count);
	int  emit_leaf_indices(int node_index);
	void emit_node(int out_index, int bvh2_index);
};

// Greedy top-down collapse of the binary BVH into 4-wide nodes (Converters/BVH4Converter.cpp:3-148):
// node i of the binary tree becomes 4-wide node i holding its two children; a node then repeatedly
// adopts the children of its largest (half-area) inner child while they fit. Node 1 is the entry
// point whose only child is the root; nodes that were adopted away stay in the array unused.
struct BVH4Converter {
	BVH4       & bvh4;
	const BVH2 & bvh2;

	BVH4Converter(BVH4 & bvh4, const BVH2 & bvh2) : bvh4(bvh4), bvh2(bvh2) { }
	void convert();

private:
	void collapse(int node_index);
};

namespace BVH {
	BVH2 create_from_triangles(const std::vector<Triangle> & triangles);
}
