#include "BVH.h"
#include "Config.h"
#include "Mesh.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ---------------------------------------------------------------------------------------------
// SAH builder
// ---------------------------------------------------------------------------------------------

SAHBuilder::SAHBuilder(BVH2 & bvh, size_t primitive_count) : bvh(bvh) {
	for (int d = 0; d < 3; d++) {
		sorted[d].resize(primitive_count);
		for (size_t i = 0; i < primitive_count; i++) sorted[d][i] = int(i);
	}
	sweep_cost   .resize(primitive_count);
	partition_tmp.resize(primitive_count);
	goes_left    .resize(primitive_count);
	bvh.nodes.reserve(2 * primitive_count);
}

namespace {
struct BuildContext {
	SAHBuilder & b;
	const std::vector<AABB> & prim_aabb;

	BVHObjectSplit find_split(int first, int count) const {
		return bvh_find_object_split([this](int axis, int i) -> const AABB & { return prim_aabb[b.sorted[axis][i]]; }, first, count, b.sweep_cost.data());
	}

	void build_node(int node_index, int first, int count) {
		if (count == 1) { // always split down to one primitive per leaf (SAHBuilder.cpp:15-24)
			BVHNode2 & leaf = b.bvh.nodes[node_index];
			leaf.first = first;
			leaf.count = 1;
			return;
		}
		BVHObjectSplit s = find_split(first, count);

		const int * split_order = b.sorted[s.axis].data();
		for (int i = first;   i < s.index;       i++) b.goes_left[split_order[i]] = 1;
		for (int i = s.index; i < first + count; i++) b.goes_left[split_order[i]] = 0;

		// Stable partition of the other two orderings so that they stay sorted per side.
		int n_left = s.index - first;
		for (int axis = 0; axis < 3; axis++) {
			if (axis == s.axis) continue;
			int * order = b.sorted[axis].data() + first;
			int * tmp = b.partition_tmp.data();
			int l = 0, r = n_left;
			for (int i = 0; i < count; i++) {
				int id = order[i];
				if (b.goes_left[id]) tmp[l++] = id; else tmp[r++] = id;
			}
			memcpy(order, tmp, size_t(count) * sizeof(int));
		}

		int child = int(b.bvh.nodes.size());
		b.bvh.nodes.emplace_back();
		b.bvh.nodes.emplace_back();
		memset(&b.bvh.nodes[child], 0, 2 * sizeof(BVHNode2));
		b.bvh.nodes[child    ].aabb = s.left;
		b.bvh.nodes[child + 1].aabb = s.right;

		BVHNode2 & node = b.bvh.nodes[node_index];
		node.left  = child;
		node.count = 0;
		node.axis  = unsigned(s.axis);

		build_node(child,     first,          n_left);
		build_node(child + 1, first + n_left, count - n_left);
	}
};
}

static void build_from_bounds(SAHBuilder & b, const std::vector<AABB> & prim_aabb, const std::vector<Vector3> & prim_center) {
	size_t n = prim_aabb.size();
	b.bvh.indices.clear();
	b.bvh.nodes.clear();
	b.bvh.nodes.reserve(std::max<size_t>(2 * n, 2));
	b.bvh.nodes.resize(2); // root + one dummy so that sibling pairs stay 64-byte aligned
	memset(b.bvh.nodes.data(), 0, 2 * sizeof(BVHNode2));

	AABB root = AABB::create_empty();
	for (size_t i = 0; i < n; i++) root.expand(prim_aabb[i]);
	b.bvh.nodes[0].aabb = root;

	// The index lists keep their previous order as the starting permutation (a per-frame
	// TLAS rebuild re-sorts the last frame's order), then get stably sorted by centroid.
	for (int axis = 0; axis < 3; axis++) {
		std::stable_sort(b.sorted[axis].begin(), b.sorted[axis].end(), [&](int l, int r) {
			return bvh_float_sort_key(prim_center[l][axis]) < bvh_float_sort_key(prim_center[r][axis]);
		});
	}

	BuildContext ctx { b, prim_aabb };
	ctx.build_node(0, 0, int(n));

	b.bvh.indices = b.sorted[0];
}

void SAHBuilder::build(const std::vector<Triangle> & triangles) {
	std::vector<AABB>    bounds(triangles.size());
	std::vector<Vector3> centers(triangles.size());
	for (size_t i = 0; i < triangles.size(); i++) {
		bounds [i] = triangles[i].get_aabb();
		centers[i] = triangles[i].get_center();
	}
	build_from_bounds(*this, bounds, centers);
}

void SAHBuilder::build(const std::vector<Mesh> & meshes) {
	std::vector<AABB>    bounds(meshes.size());
	std::vector<Vector3> centers(meshes.size());
	for (size_t i = 0; i < meshes.size(); i++) {
		bounds [i] = meshes[i].get_aabb();
		centers[i] = meshes[i].get_center();
	}
	build_from_bounds(*this, bounds, centers);
}

void SAHBuilder::build(const std::vector<AABB> & boxes) {
	std::vector<Vector3> centers(boxes.size());
	for (size_t i = 0; i < boxes.size(); i++) centers[i] = boxes[i].get_center();
	build_from_bounds(*this, boxes, centers);
}

BVH2 BVH::create_sah_from_triangles(const std::vector<Triangle> & triangles) {
	BVH2 bvh;
	SAHBuilder(bvh, triangles.size()).build(triangles);
	if (cpu_config.enable_bvh_optimization) BVHOptimizer::optimize(bvh); // reference: BVH.cpp:31-33
	return bvh;
}

BVH2 BVH::create_from_triangles(const std::vector<Triangle> & triangles) {
	if (cpu_config.bvh_type != BVHType::SBVH) return create_sah_from_triangles(triangles);
	BVH2 bvh;
	SBVHBuilder(bvh, triangles.size()).build(triangles);
	if (cpu_config.enable_bvh_optimization) BVHOptimizer::optimize(bvh);
	return bvh;
}

// ---------------------------------------------------------------------------------------------
// BVH2 -> BVH8 (CWBVH)
// ---------------------------------------------------------------------------------------------

int bvh8_order_breadth_first(BVH8 & bvh, int max_depth) {
	if (bvh.nodes.empty()) return 0;
	std::vector<int> order, depth;            // order[new index] = old index
	order.reserve(bvh.nodes.size()); depth.reserve(bvh.nodes.size());
	order.push_back(0); depth.push_back(0);
	std::vector<unsigned> new_child_base(bvh.nodes.size(), 0);
	for (size_t at = 0; at < order.size(); at++) {
		const BVHNode8 & node = bvh.nodes[size_t(order[at])];
		new_child_base[size_t(order[at])] = unsigned(order.size());
		int children = __builtin_popcount(unsigned(node.imask));
		for (int c = 0; c < children; c++) { order.push_back(int(node.base_index_child) + c); depth.push_back(depth[at] + 1); }
	}
	if (order.size() != bvh.nodes.size()) throw std::runtime_error("bvh8_order_breadth_first: the node array holds nodes the root does not reach");
	std::vector<BVHNode8> moved(bvh.nodes.size());
	int top = 0;
	for (size_t at = 0; at < order.size(); at++) {
		moved[at] = bvh.nodes[size_t(order[at])];
		if (moved[at].imask) moved[at].base_index_child = new_child_base[size_t(order[at])];
		if (depth[at] <= max_depth) top = int(at) + 1;
	}
	bvh.nodes.swap(moved);
	return top;
}

void BVH8Converter::convert() {
	bvh8.indices.clear();
	bvh8.indices.reserve(bvh2.indices.size());
	bvh8.nodes.clear();
	bvh8.nodes.reserve(bvh2.nodes.size());
	bvh8.nodes.emplace_back();
	memset(&bvh8.nodes[0], 0, sizeof(BVHNode8));

	table.assign(bvh2.nodes.size() * 7, Decision { LEAF, char(INVALID), char(INVALID), 0.0f });

	fill_cost_table(0); // bottom-up dynamic programme (BVH8Converter.cpp:24-115)
	emit_node(0, 0);    // top-down collapse          (BVH8Converter.cpp:223-335)
}

// table[n*7 + i] = cheapest way to represent the subtree of n as a forest of at most i+1
// roots; i == 0 decides between "one leaf" (<= 3 triangles) and "one wide inner node".
int BVH8Converter::fill_cost_table(int node_index) {
	const BVHNode2 & node = bvh2.nodes[node_index];
	Decision * row = &table[size_t(node_index) * 7];

	if (node.is_leaf()) {
		if (node.count != 1) throw std::runtime_error("BVH8 conversion needs exactly one primitive per BVH2 leaf");
		float cost_leaf = node.aabb.surface_area() * float(node.count) * primitive_cost;
		for (int i = 0; i < 7; i++) { row[i].kind = LEAF; row[i].cost = cost_leaf; }
		return int(node.count);
	}

	int num_primitives = fill_cost_table(node.left) + fill_cost_table(node.left + 1);
	const Decision * row_l = &table[size_t(node.left)     * 7];
	const Decision * row_r = &table[size_t(node.left + 1) * 7];

	{
		float cost_leaf = num_primitives <= 3 ? float(num_primitives) * node.aabb.surface_area() * primitive_cost : INFINITY;

		float cost_distribute = INFINITY;
		char  take_l = char(INVALID), take_r = char(INVALID);
		for (int k = 0; k < 7; k++) {
			float c = row_l[k].cost + row_r[6 - k].cost;
			if (c < cost_distribute) { cost_distribute = c; take_l = char(k); take_r = char(6 - k); }
		}
		// (boxes of non-finite vertices make every candidate NaN or +inf and none compares less: the budget is then halved -- any consistent split keeps the
		// collapse within 8 children; with finite boxes the first candidate always compares less than INFINITY and nothing changes)
		if (take_l == char(INVALID)) { take_l = char(3); take_r = char(3); }
		float cost_internal = cost_distribute + node.aabb.surface_area();

		if (cost_leaf < cost_internal) { row[0].kind = LEAF;     row[0].cost = cost_leaf; }
		else                           { row[0].kind = INTERNAL; row[0].cost = cost_internal; }
		row[0].take_left  = take_l;
		row[0].take_right = take_r;
	}

	for (int i = 1; i < 7; i++) {
		float cost_distribute = row[i - 1].cost;
		char  take_l = char(INVALID), take_r = char(INVALID);
		for (int k = 0; k < i; k++) {
			float c = row_l[k].cost + row_r[i - k - 1].cost;
			if (c < cost_distribute) { cost_distribute = c; take_l = char(k); take_r = char(i - k - 1); }
		}
		row[i].cost = cost_distribute;
		if (take_l != char(INVALID)) {
			row[i].kind = DISTRIBUTE;
			row[i].take_left  = take_l;
			row[i].take_right = take_r;
		} else {
			row[i] = row[i - 1];
		}
	}
	return num_primitives;
}

void BVH8Converter::gather_children(int node_index, int budget, int children[8], int & child_count) {
	const BVHNode2 & node = bvh2.nodes[node_index];
	if (child_count >= 8 || budget < 0 || budget > 6) throw std::runtime_error("BVH8 conversion: a node's children do not fit its eight slots (inconsistent cost table)");
	if (node.is_leaf()) { children[child_count++] = node_index; return; }

	int take_l = table[size_t(node_index) * 7 + budget].take_left;
	int take_r = table[size_t(node_index) * 7 + budget].take_right;
	if (take_l < 0 || take_r < 0 || take_l + take_r > 6) throw std::runtime_error("BVH8 conversion: inconsistent cost table");

	if (table[size_t(node.left) * 7 + take_l].kind == DISTRIBUTE) gather_children(node.left, take_l, children, child_count);
	else { if (child_count >= 8) throw std::runtime_error("BVH8 conversion: a node's children do not fit its eight slots (inconsistent cost table)"); children[child_count++] = node.left; }

	if (table[size_t(node.left + 1) * 7 + take_r].kind == DISTRIBUTE) gather_children(node.left + 1, take_r, children, child_count);
	else { if (child_count >= 8) throw std::runtime_error("BVH8 conversion: a node's children do not fit its eight slots (inconsistent cost table)"); children[child_count++] = node.left + 1; }
}

// Assignment of children to the 8 octant slots: slot s is entered first by rays whose direction signs are s, so a child should sit in
// the slot whose diagonal points towards it.
//   slot_assignment 0: the reference's rule (BVH8Converter.cpp:146-205) -- every child competes, by where its CENTRE lies, greedily
//                      (strict '<' => first minimum in (child, slot) order). Per-mesh trees and the TLAS: byte-identical to the reference.
//   slot_assignment 5: the flattened tree's rule (cpu_config.static_slot_assignment) -- only INNER children compete (a leaf's triangles are
//                      tested in the node's own round whatever its slot: leaves take what is left), by the CORNER a ray along the slot's
//                      diagonal enters the child's box at, and by the assignment of least TOTAL cost (children in order over the subsets
//                      of slots already given away: 8 x 256 steps). Other rules that were measured: profiles/r05_slot_assignment.txt.
void BVH8Converter::assign_octant_slots(int node_index, int children[8], int child_count) {
	Vector3 p = bvh2.nodes[node_index].aabb.get_center();

	float cost[8][8] = { };
	for (int c = 0; c < child_count; c++) {
		const AABB & b = bvh2.nodes[children[c]].aabb;
		Vector3 offset = b.get_center() - p;
		const bool leaf = table[size_t(children[c]) * 7].kind == LEAF;
		for (int s = 0; s < 8; s++) {
			Vector3 diagonal((s & 4) ? -1.0f : +1.0f, (s & 2) ? -1.0f : +1.0f, (s & 1) ? -1.0f : +1.0f);
			if (slot_assignment == 0) cost[c][s] = Vector3::dot(offset, diagonal);
			else {
				Vector3 near_corner((s & 4) ? b.max.x : b.min.x, (s & 2) ? b.max.y : b.min.y, (s & 1) ? b.max.z : b.min.z);
				cost[c][s] = leaf ? 0.0f : Vector3::dot(near_corner - p, diagonal);
			}
		}
	}

	int  slot_of_child[8] = { INVALID, INVALID, INVALID, INVALID, INVALID, INVALID, INVALID, INVALID };
	bool slot_taken[8] = { };
	if (slot_assignment != 0) {
		float best[256]; signed char from[256];
		for (int m = 0; m < 256; m++) { best[m] = INFINITY; from[m] = -1; }
		best[0] = 0.0f;
		for (int m = 0; m < 256; m++) {
			int c = __builtin_popcount(unsigned(m));
			if (c >= child_count || !(best[m] < INFINITY)) continue;
			for (int s = 0; s < 8; s++) if (!((m >> s) & 1)) { float v = best[m] + cost[c][s]; if (v < best[m | (1 << s)]) { best[m | (1 << s)] = v; from[m | (1 << s)] = (signed char)s; } }
		}
		int end = -1; float least = INFINITY;
		for (int m = 0; m < 256; m++) if (__builtin_popcount(unsigned(m)) == child_count && best[m] < least) { least = best[m]; end = m; }
		for (int c = child_count - 1, m = end; c >= 0 && m > 0; c--) { int s = from[m]; slot_of_child[c] = s; slot_taken[s] = true; m &= ~(1 << s); }
	} else
	while (true) {
		float best = INFINITY;
		int best_slot = INVALID, best_child = INVALID;
		for (int c = 0; c < child_count; c++) {
			if (slot_of_child[c] != INVALID) continue;
			for (int s = 0; s < 8; s++) {
				if (!slot_taken[s] && cost[c][s] < best) { best = cost[c][s]; best_slot = s; best_child = c; }
			}
		}
		if (best_slot == INVALID) break;
		slot_taken[best_slot] = true;
		slot_of_child[best_child] = best_slot;
	}

	// Children whose costs never compared (NaN boxes, or nothing but +inf left) take the free slots in order
	for (int c = 0; c < child_count; c++) {
		if (slot_of_child[c] != INVALID) continue;
		int s = 0;
		while (slot_taken[s]) s++;
		slot_taken[s] = true;
		slot_of_child[c] = s;
	}

	int unordered[8];
	for (int i = 0; i < 8; i++) { unordered[i] = children[i]; children[i] = INVALID; }
	for (int c = 0; c < child_count; c++) children[slot_of_child[c]] = unordered[c];
}

int BVH8Converter::emit_leaf_indices(int node_index) {
	const BVHNode2 & node = bvh2.nodes[node_index];
	if (node.is_leaf()) {
		for (unsigned i = 0; i < node.count; i++) bvh8.indices.push_back(bvh2.indices[node.first + i]);
		return int(node.count);
	}
	return emit_leaf_indices(node.left) + emit_leaf_indices(node.left + 1);
}

void BVH8Converter::emit_node(int out_index, int bvh2_index) {
	const AABB & aabb = bvh2.nodes[bvh2_index].aabb;

	BVHNode8 node;
	memset(&node, 0, sizeof(node));
	node.p = aabb.min;

	// Grid scale per axis: smallest power of two e with (max-min)/e <= 255, stored as its
	// 8-bit float exponent (BVH8Converter.cpp:229-253).
	constexpr float denom = 1.0f / float((1 << 8) - 1);
	Vector3 e(
		exp2f(ceilf(log2f((aabb.max.x - aabb.min.x) * denom))),
		exp2f(ceilf(log2f((aabb.max.y - aabb.min.y) * denom))),
		exp2f(ceilf(log2f((aabb.max.z - aabb.min.z) * denom))));
	Vector3 one_over_e(1.0f / e.x, 1.0f / e.y, 1.0f / e.z);
	for (int d = 0; d < 3; d++) {
		unsigned bits; float v = e[d]; memcpy(&bits, &v, 4);
		node.e[d] = byte(bits >> 23);
	}

	int child_count = 0;
	int children[8] = { INVALID, INVALID, INVALID, INVALID, INVALID, INVALID, INVALID, INVALID };
	gather_children(bvh2_index, 0, children, child_count);
	assign_octant_slots(bvh2_index, children, child_count);

	node.imask = 0;
	node.base_index_triangle = unsigned(bvh8.indices.size());
	node.base_index_child    = unsigned(bvh8.nodes.size());

	int num_internal = 0, num_triangles = 0;
	for (int i = 0; i < 8; i++) {
		int child = children[i];
		if (child == INVALID) continue;
		const AABB & cb = bvh2.nodes[child].aabb;

		node.quantized_min_x[i] = byte(floorf((cb.min.x - node.p.x) * one_over_e.x));
		node.quantized_min_y[i] = byte(floorf((cb.min.y - node.p.y) * one_over_e.y));
		node.quantized_min_z[i] = byte(floorf((cb.min.z - node.p.z) * one_over_e.z));
		node.quantized_max_x[i] = byte(ceilf ((cb.max.x - node.p.x) * one_over_e.x));
		node.quantized_max_y[i] = byte(ceilf ((cb.max.y - node.p.y) * one_over_e.y));
		node.quantized_max_z[i] = byte(ceilf ((cb.max.z - node.p.z) * one_over_e.z));

		if (table[size_t(child) * 7].kind == LEAF) {
			int triangle_count = emit_leaf_indices(child);
			for (int j = 0; j < triangle_count; j++) node.meta[i] |= byte(1 << (j + 5)); // unary count
			node.meta[i] |= byte(num_triangles);                                          // offset from base
			num_triangles += triangle_count;
		} else {
			node.meta[i] = byte((i + 24) | 0x20);
			node.imask |= byte(1 << i);
			num_internal++;
		}
	}

	for (int i = 0; i < num_internal; i++) {
		bvh8.nodes.emplace_back();
		memset(&bvh8.nodes.back(), 0, sizeof(BVHNode8));
	}
	bvh8.nodes[out_index] = node;

	int next = 0;
	for (int i = 0; i < 8; i++) {
		if (children[i] == INVALID) continue;
		if (node.imask & (1 << i)) emit_node(int(node.base_index_child) + next++, children[i]);
	}
}


// ---- BVH4 ------------------------------------------------------------------------------------------

static void set_child_box(BVHNode4 & node, int slot, const AABB & box) {
	node.aabb_min_x[slot] = box.min.x; node.aabb_min_y[slot] = box.min.y; node.aabb_min_z[slot] = box.min.z;
	node.aabb_max_x[slot] = box.max.x; node.aabb_max_y[slot] = box.max.y; node.aabb_max_z[slot] = box.max.z;
}

static void copy_child(BVHNode4 & dst, int dst_slot, const BVHNode4 & src, int src_slot) {
	dst.aabb_min_x[dst_slot] = src.aabb_min_x[src_slot]; dst.aabb_min_y[dst_slot] = src.aabb_min_y[src_slot]; dst.aabb_min_z[dst_slot] = src.aabb_min_z[src_slot];
	dst.aabb_max_x[dst_slot] = src.aabb_max_x[src_slot]; dst.aabb_max_y[dst_slot] = src.aabb_max_y[src_slot]; dst.aabb_max_z[dst_slot] = src.aabb_max_z[src_slot];
	dst.index_and_count[dst_slot] = src.index_and_count[src_slot];
}

void BVH4Converter::convert() {
	bvh4.nodes.assign(bvh2.nodes.size(), BVHNode4());
	memset((void *)bvh4.nodes.data(), 0, bvh4.nodes.size() * sizeof(BVHNode4));

	for (size_t i = 0; i < bvh4.nodes.size(); i++) {
		BVHNode4 & out = bvh4.nodes[i];
		if (i == 1) { // entry point: its first child is the root
			out.get_index(0) = 0;
			out.get_count(0) = 0;
			continue;
		}
		const BVHNode2 & in = bvh2.nodes[i];
		if (in.is_leaf()) continue;

		for (int side = 0; side < 2; side++) {
			const BVHNode2 & child = bvh2.nodes[in.left + side];
			set_child_box(out, side, child.aabb);
			if (child.is_leaf()) { out.get_index(side) = child.first;    out.get_count(side) = int(child.count); }
			else                 { out.get_index(side) = in.left + side; out.get_count(side) = 0; }
		}
		for (int slot = 2; slot < 4; slot++) { out.get_index(slot) = INVALID; out.get_count(slot) = INVALID; }
	}

	if (bvh2.nodes[0].is_leaf()) { // a single-leaf tree: the root node holds that leaf
		BVHNode4 & root = bvh4.nodes[0];
		set_child_box(root, 0, bvh2.nodes[0].aabb);
		root.get_index(0) = bvh2.nodes[0].first;
		root.get_count(0) = int(bvh2.nodes[0].count);
		for (int slot = 1; slot < 4; slot++) { root.get_index(slot) = INVALID; root.get_count(slot) = INVALID; }
	} else {
		collapse(0);
	}
	bvh4.indices = bvh2.indices;
}

void BVH4Converter::collapse(int node_index) {
	BVHNode4 & node = bvh4.nodes[node_index];

	while (true) {
		int child_count = node.get_child_count();

		// the adoptable inner child with the largest half area; ties keep the first
		float best_area = -INFINITY;
		int   best = INVALID;
		for (int i = 0; i < child_count; i++) {
			if (node.is_leaf(i)) continue;
			int grandchildren = bvh4.nodes[node.get_index(i)].get_child_count();
			if (child_count + grandchildren - 1 > 4) continue;
			float dx = node.aabb_max_x[i] - node.aabb_min_x[i];
			float dy = node.aabb_max_y[i] - node.aabb_min_y[i];
			float dz = node.aabb_max_z[i] - node.aabb_min_z[i];
			float half_area = dx * dy + dy * dz + dz * dx;
			if (half_area > best_area) { best_area = half_area; best = i; }
		}
		if (best == INVALID) break;

		const BVHNode4 adopted = bvh4.nodes[node.get_index(best)];
		int adopted_count = adopted.get_child_count();
		copy_child(node, best, adopted, 0);                                                  // its first child takes its slot
		for (int i = 1; i < adopted_count; i++) copy_child(node, child_count + i - 1, adopted, i); // the others are appended
	}

	for (int i = 0; i < 4; i++) {
		if (node.get_count(i) == INVALID) break;
		if (node.get_count(i) == 0) collapse(node.get_index(i));
	}
}
