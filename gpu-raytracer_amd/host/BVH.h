// CPU BVH construction: full-sweep SAH binary BVH (1 primitive per leaf) and its
// collapse into the 80-byte compressed wide BVH8 (CWBVH, Ylitie et al. 2017) that
// the trace kernels consume.  The node formats and every tie-breaking rule follow
// the reference so that the produced bytes are identical to its builder's
// (Src/BVH/BVH.h:11-80, Builders/SAHBuilder.cpp:13-104, Builders/BVHPartitions.cpp:6-54,
// Converters/BVH8Converter.cpp:7-335); tests/test_bvh_build.py checks this against
// the verbatim reference build in oracle/_ref.
#pragma once
#include <vector>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>

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

// Renumbers the nodes of a tree rooted in node 0 level by level (children of a node stay consecutive, in slot order: traversal does
// not notice). Afterwards the nodes of depth <= max_depth are the first `return value` nodes: what the traversal kernel of the
// flattened scene keeps together at the front of its node array -- every ray walks them, 40 % of all node steps on Sponza touch the top three levels.
int bvh8_order_breadth_first(BVH8 & bvh, int max_depth);

// Re-seats the children of every node in the octant slots of least cost as `rays` sample rays (seeded) over `triangles` -- the array bvh.indices points into --
// find it (SlotOrder.cpp); boxes, leaves and triangle order stay. For trees rooted in node 0 in the converter's own (depth-first) order: call it BEFORE
// bvh8_order_breadth_first. thread_count <= 0: all host threads. view (optional): three quarters of the sample rays are then paths from that camera.
struct SlotLearningView {   // what the caller is about to look at: the camera as the device gets it (camera_generate_ray: direction = corner + x_axis * px + y_axis * py)
	Vector3 position, bottom_left_corner, x_axis, y_axis; int width = 0, height = 0;
};
void bvh8_learn_slot_order(BVH8 & bvh, const std::vector<struct Triangle> & triangles, int rays, int thread_count = 0, const SlotLearningView * view = nullptr);

struct Mesh;

// Monotone float -> unsigned key (the reference radix-sorts on it, Core/Sort.h:133-140),
// so -0.0 orders before +0.0 and a stable sort on it reproduces the reference's order.
inline unsigned bvh_float_sort_key(float x) {
	unsigned u; memcpy(&u, &x, 4);
	unsigned mask = unsigned(-int(u >> 31)) | 0x80000000u;
	return u ^ mask;
}

struct BVHObjectSplit {
	int   index = -1;     // first primitive (position in the sorted list) of the right side
	int   axis  = -1;
	float cost  = INFINITY;
	AABB  left  = AABB::create_empty();
	AABB  right = AABB::create_empty();
};

// Full-sweep SAH over the three pre-sorted orders (reference: BVHPartitions.cpp:6-54);
// box_at(axis, i) is the box of the i-th primitive in that axis' order. '<=' keeps the LAST best
// candidate, i.e. later axes and smaller indices win ties.
template<typename BoxAt>
BVHObjectSplit bvh_find_object_split(BoxAt box_at, int first, int count, float * partial) {
	BVHObjectSplit s;
	for (int axis = 0; axis < 3; axis++) {
		AABB grow_l = AABB::create_empty();
		for (int i = 1; i < count; i++) {
			grow_l.expand(box_at(axis, first + i - 1));
			partial[i] = grow_l.surface_area() * float(i);
		}
		AABB grow_r = AABB::create_empty();
		for (int i = count - 1; i > 0; i--) {
			grow_r.expand(box_at(axis, first + i));
			float c = partial[i] + grow_r.surface_area() * float(count - i);
			if (c <= s.cost) {
				s.cost  = c;
				s.index = first + i;
				s.axis  = axis;
				s.right = grow_r;
			}
		}
	}
	if (s.axis < 0) {
		// Every candidate cost was NaN (non-finite boxes): no comparison above could succeed. Split in the
		// middle of the x order so that callers still get a valid partition.
		s.axis  = 0;
		s.index = first + count / 2;
		s.cost  = INFINITY;
		for (int i = s.index; i < first + count; i++) s.right.expand(box_at(0, i));
	}
	for (int i = first; i < s.index; i++) s.left.expand(box_at(s.axis, i));
	return s;
}

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
	void build(const std::vector<AABB>     & boxes);   // any boxes (the TLAS over flattened static geometry + the moving instances)
};

struct BVH8Converter {
	BVH8       & bvh8;
	const BVH2 & bvh2;

	float primitive_cost = 1.0f;   // SAH cost of a triangle test relative to a node step: 1 in the reference's converter (BVH8Converter.cpp:24-115)
	int   slot_assignment = 0;     // 0: the reference's greedy assignment of children to octant slots by their centres; otherwise (5): inner children only, by entry corner, least total cost (BVH.cpp: assign_octant_slots)
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

// SAH object splits + binned spatial splits (SBVH.cpp); leaves hold one reference each and a
// triangle may be referenced from several leaves, so indices.size() >= triangle count.
struct SBVHBuilder {
	struct Ref { int triangle; AABB box; }; // the part of a triangle that lies in a subtree

	BVH2 & bvh;

	std::vector<Ref>   refs[3];   // the references of the subtree under construction, per axis order
	std::vector<float> sweep_cost;
	std::vector<char>  goes_left;

	SBVHBuilder(BVH2 & bvh, size_t /*triangle_count*/) : bvh(bvh) { }

	void build(const std::vector<Triangle> & triangles);
};

// The binary tree behind the flattened static geometry (StaticBVHBuilder.cpp): binned SAH object + spatial splits, built by all
// host threads; one triangle reference per leaf, a triangle cut by spatial splits is listed once per leaf that holds a part.
namespace StaticBVHBuilder {
	void build(BVH2 & bvh, const std::vector<Triangle> & triangles, int thread_count = 0);
	// Early split clipping for the device builder: per reference the triangle it is a piece of and the piece's box (6 floats: min, max)
	void presplit(const std::vector<Triangle> & triangles, float limit, std::vector<int> & source, std::vector<float> & boxes, int max_pieces = 64);
}

namespace BVHOptimizer {
	void optimize(BVH2 & bvh); // BVHOptimizer.cpp: insertion-based optimisation (Bittner et al. 2013), cpu_config.enable_bvh_optimization
}

namespace BVHCollapser {
	void collapse(BVH2 & bvh); // SBVH.cpp
}

namespace BVH {
	// Binary BVH of cpu_config.bvh_type: spatial-split builder for SBVH, plain SAH otherwise
	// (reference: BVH.cpp:14-36)
	BVH2 create_from_triangles(const std::vector<Triangle> & triangles);
	BVH2 create_sah_from_triangles(const std::vector<Triangle> & triangles);
}
