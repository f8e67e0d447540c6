// oracle_tlas.cpp -- TEST INFRASTRUCTURE (never on the product path).
//
// CPU restatement of the device TLAS build (gpu-raytracer_amd/csrc/kernels_build.hip + rt_tlas_build.h): Morton-ordered
// instances, every node a contiguous run of that order cut into up to eight runs, CWBVH nodes (the reference's 80 bytes,
// CUDA/Raytracing/BVH8.h:19-25; exponent and quantisation rules of BVH/Converters/BVH8Converter.cpp:229-283, slot
// assignment of BVH8Converter.cpp:146-205) whose leaves are single instances.
//
// SELF-CONTAINED: nothing of the product is included. Round 3's version included the product's rt_tlas_build.h, so
// the byte comparison "kernel == restatement" compared that header with itself; here every step is written out again,
// in this file's own terms (std:: containers, doubles nowhere -- the arithmetic that decides bytes is float, in the
// order the build defines it -- recursion-free level loop), so that the comparison has two authors' worth of code on
// its two sides. What must agree with the device, and why:
//   * the world box of an instance: min / max over the 8 corners of m * corner, the products summed left to right;
//   * the Morton key: 10 bits per axis of the box centre's position inside the scene box, x highest;
//   * the cut of a run: where the highest differing key bit flips (equal keys: the middle);
//   * up to eight runs per node: cut the widest run (first such) until eight or all single;
//   * slots: greedy smallest (centre offset . slot diagonal), ties to the lower child, then the lower slot;
//   * node bytes: as the reference's converter writes them.
// tests/test_tlas.py additionally checks the result on its own terms (every instance in exactly one leaf, child boxes
// contain their instances, inner children consecutive) and tests/test_gpu_tlas.py that tracing it gives the hits of
// the host-built TLAS, whose builder IS the reference's (Integrator.cpp:399-430) byte for byte.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Box3 { float lo[3], hi[3]; };

const float HUGE_BOUND = 3.0e38f;

Box3 empty_box() { Box3 b; for (int a = 0; a < 3; a++) { b.lo[a] = HUGE_BOUND; b.hi[a] = -HUGE_BOUND; } return b; }

// (a < b ? a : b), not std::min / fminf: what happens to a NaN is part of the definition
float lesser(float a, float b)  { return a < b ? a : b; }
float greater(float a, float b) { return a > b ? a : b; }

void include(Box3 & into, const Box3 & other) {
	for (int a = 0; a < 3; a++) { into.lo[a] = lesser(into.lo[a], other.lo[a]); into.hi[a] = greater(into.hi[a], other.hi[a]); }
}

Box3 instance_world_box(const float * matrix3x4, const float * object_lo, const float * object_hi) {
	Box3 world = empty_box();
	for (int corner = 0; corner < 8; corner++) {
		const float point[3] = { (corner & 1) ? object_hi[0] : object_lo[0], (corner & 2) ? object_hi[1] : object_lo[1], (corner & 4) ? object_hi[2] : object_lo[2] };
		for (int row = 0; row < 3; row++) {
			const float * r = matrix3x4 + 4 * row;
			float coordinate = r[0] * point[0] + r[1] * point[1] + r[2] * point[2] + r[3];
			world.lo[row] = lesser(world.lo[row], coordinate);
			world.hi[row] = greater(world.hi[row], coordinate);
		}
	}
	return world;
}

// bit i of a 10-bit number moves to bit 3 i
uint32_t spread_by_three(uint32_t ten_bits) {
	uint32_t out = 0;
	for (int i = 0; i < 10; i++) out |= ((ten_bits >> i) & 1u) << (3 * i);
	return out;
}

uint32_t morton_of(const Box3 & box, const Box3 & scene) {
	uint32_t code = 0;
	for (int a = 0; a < 3; a++) {
		float extent = scene.hi[a] - scene.lo[a];
		float centre = 0.5f * (box.lo[a] + box.hi[a]);
		float position = extent > 0.0f ? (centre - scene.lo[a]) / extent : 0.5f;
		float cell = lesser(greater(position * 1024.0f, 0.0f), 1023.0f);
		code |= spread_by_three(uint32_t(cell)) << (2 - a);
	}
	return code;
}

uint32_t code_of(uint64_t key) { return uint32_t(key >> 32); }

int cut_of_run(const std::vector<uint64_t> & keys, int first, int end) {
	uint32_t a = code_of(keys[size_t(first)]), b = code_of(keys[size_t(end - 1)]);
	if (a == b) return (first + end) / 2;
	uint32_t differing = a ^ b, flip = 0x80000000u;
	while (!(differing & flip)) flip >>= 1;
	// the codes are sorted: the bit is clear up to some index and set from there on
	int clear = first, set = end - 1;
	while (set - clear > 1) { int middle = (clear + set) / 2; if (code_of(keys[size_t(middle)]) & flip) set = middle; else clear = middle; }
	return set;
}

// boundaries[0 .. count] of the runs [first, end) is cut into
int runs_of_node(const std::vector<uint64_t> & keys, int first, int end, int boundaries[9]) {
	std::vector<int> b = { first, end };
	while (int(b.size()) - 1 < 8) {
		int widest = -1, width = 1;
		for (int c = 0; c + 1 < int(b.size()); c++) if (b[size_t(c) + 1] - b[size_t(c)] > width) { width = b[size_t(c) + 1] - b[size_t(c)]; widest = c; }
		if (widest < 0) break;
		b.insert(b.begin() + widest + 1, cut_of_run(keys, b[size_t(widest)], b[size_t(widest) + 1]));
	}
	for (int c = 0; c < 9; c++) boundaries[c] = c < int(b.size()) ? b[size_t(c)] : end;
	return int(b.size()) - 1;
}

void slots_for_children(const Box3 & node, const Box3 * children, int count, int slot_of_child[8]) {
	float middle[3];
	for (int a = 0; a < 3; a++) middle[a] = 0.5f * (node.lo[a] + node.hi[a]);
	float cost[8][8];
	for (int c = 0; c < count; c++) {
		float offset[3];
		for (int a = 0; a < 3; a++) offset[a] = 0.5f * (children[c].lo[a] + children[c].hi[a]) - middle[a];
		for (int s = 0; s < 8; s++) {
			float dx = (s & 4) ? -1.0f : 1.0f, dy = (s & 2) ? -1.0f : 1.0f, dz = (s & 1) ? -1.0f : 1.0f;
			cost[c][s] = offset[0] * dx + offset[1] * dy + offset[2] * dz;
		}
	}
	bool slot_taken[8] = { false, false, false, false, false, false, false, false };
	for (int c = 0; c < 8; c++) slot_of_child[c] = -1;
	for (int placed = 0; placed < count; placed++) {
		float lowest = HUGE_BOUND; int which_child = -1, which_slot = -1;
		for (int c = 0; c < count; c++) {
			if (slot_of_child[c] >= 0) continue;
			for (int s = 0; s < 8; s++) if (!slot_taken[s] && cost[c][s] < lowest) { lowest = cost[c][s]; which_child = c; which_slot = s; }
		}
		if (which_slot < 0) break;   // only NaN costs are left
		slot_taken[which_slot] = true; slot_of_child[which_child] = which_slot;
	}
	for (int c = 0; c < count; c++) if (slot_of_child[c] < 0) {
		int s = 0; while (slot_taken[s]) s++;
		slot_taken[s] = true; slot_of_child[c] = s;
	}
}

uint32_t bits_of(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
float float_of(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// kind[s]: -1 empty slot, 0 a single instance, 1 an inner node
void write_node(const Box3 & node, const Box3 slot_box[8], const int kind[8], uint32_t first_child_node, uint32_t first_leaf, uint32_t * words20) {
	std::memset(words20, 0, 80);
	float inverse_scale[3]; uint32_t exponent_bytes = 0;
	for (int a = 0; a < 3; a++) {
		// the smallest power of two e with extent / e <= 255; a flat axis gets the smallest normal scale
		float extent = greater(node.hi[a] - node.lo[a], 1.0e-30f);
		uint32_t scale_bits = bits_of(extent * (1.0f / 255.0f));
		uint32_t exponent = scale_bits >> 23;
		if (scale_bits & 0x007fffffu) exponent += 1;                 // not a power of two already
		exponent = std::min(std::max(exponent, 1u), 254u);
		inverse_scale[a] = 1.0f / float_of(exponent << 23);
		exponent_bytes |= exponent << (8 * a);
		words20[a] = bits_of(node.lo[a]);
	}
	uint8_t * meta = reinterpret_cast<uint8_t *>(words20 + 6);
	uint8_t * planes = reinterpret_cast<uint8_t *>(words20 + 8);   // lo_x[8] hi_x[8] lo_y[8] hi_y[8] lo_z[8] hi_z[8]
	uint32_t inner_mask = 0, leaves_so_far = 0;
	for (int s = 0; s < 8; s++) {
		if (kind[s] < 0) continue;
		for (int a = 0; a < 3; a++) {
			float low  = std::floor((slot_box[s].lo[a] - node.lo[a]) * inverse_scale[a]);
			float high = std::ceil ((slot_box[s].hi[a] - node.lo[a]) * inverse_scale[a]);
			planes[16 * a + s]     = uint8_t(lesser(greater(low,  0.0f), 255.0f));
			planes[16 * a + 8 + s] = uint8_t(lesser(greater(high, 0.0f), 255.0f));
		}
		if (kind[s] == 1) { meta[s] = uint8_t(0x20 | (24 + s)); inner_mask |= 1u << s; }
		else              { meta[s] = uint8_t(0x20 | leaves_so_far); leaves_so_far++; }     // unary count 001, offset from first_leaf
	}
	words20[3] = exponent_bytes | (inner_mask << 24);
	words20[4] = first_child_node;
	words20[5] = first_leaf;
}

struct PendingNode { int node, first, end; };

} // namespace

extern "C" int oracle_tlas_build(const float * transforms /* 12 per instance */, const float * local_boxes /* 6 per instance */, int n,
                                 uint32_t * nodes /* 20 words x 2n */, int * order /* n */) {
	std::vector<Box3> world(static_cast<size_t>(n));
	Box3 scene = empty_box();
	for (int i = 0; i < n; i++) {
		world[size_t(i)] = instance_world_box(transforms + 12 * size_t(i), local_boxes + 6 * size_t(i), local_boxes + 6 * size_t(i) + 3);
		include(scene, world[size_t(i)]);
	}
	std::vector<uint64_t> keys(static_cast<size_t>(n));
	for (int i = 0; i < n; i++) keys[size_t(i)] = (uint64_t(morton_of(world[size_t(i)], scene)) << 32) | uint64_t(uint32_t(i));
	std::sort(keys.begin(), keys.end());
	auto instance_at = [&](int position) { return int(keys[size_t(position)] & 0xffffffffull); };

	// level by level; within a level the nodes number their inner children and their leaves in level order (what the device's
	// prefix sums over a level produce)
	std::vector<PendingNode> level = { { 0, 0, n } };
	int nodes_used = 1, leaves_used = 0;
	while (!level.empty()) {
		struct Cut { int boundaries[9]; int children, inner, leaves, first_child, first_leaf; };
		std::vector<Cut> cuts(level.size());
		int next_node = nodes_used, next_leaf = leaves_used;
		for (size_t k = 0; k < level.size(); k++) {
			Cut & cut = cuts[k];
			cut.children = runs_of_node(keys, level[k].first, level[k].end, cut.boundaries);
			cut.inner = 0;
			for (int c = 0; c < cut.children; c++) if (cut.boundaries[c + 1] - cut.boundaries[c] > 1) cut.inner++;
			cut.leaves = cut.children - cut.inner;
			cut.first_child = next_node; cut.first_leaf = next_leaf;
			next_node += cut.inner; next_leaf += cut.leaves;
		}
		std::vector<PendingNode> below(static_cast<size_t>(next_node - nodes_used));
		const int first_node_below = nodes_used;
		nodes_used = next_node; leaves_used = next_leaf;
		for (size_t k = 0; k < level.size(); k++) {
			const Cut & cut = cuts[k];
			Box3 node_box = empty_box(), child_box[8];
			for (int c = 0; c < cut.children; c++) {
				child_box[c] = empty_box();
				for (int position = cut.boundaries[c]; position < cut.boundaries[c + 1]; position++) include(child_box[c], world[size_t(instance_at(position))]);
				include(node_box, child_box[c]);
			}
			int slot_of_child[8];
			slots_for_children(node_box, child_box, cut.children, slot_of_child);
			Box3 slot_box[8]; int kind[8], child_in_slot[8];
			for (int s = 0; s < 8; s++) { kind[s] = -1; child_in_slot[s] = -1; }
			for (int c = 0; c < cut.children; c++) {
				int s = slot_of_child[c];
				slot_box[s] = child_box[c]; child_in_slot[s] = c;
				kind[s] = cut.boundaries[c + 1] - cut.boundaries[c] > 1 ? 1 : 0;
			}
			write_node(node_box, slot_box, kind, uint32_t(cut.first_child), uint32_t(cut.first_leaf), nodes + 20 * size_t(level[k].node));
			int inner_seen = 0, leaves_seen = 0;
			for (int s = 0; s < 8; s++) {   // inner children and leaves are numbered in slot order
				int c = child_in_slot[s];
				if (c < 0) continue;
				if (kind[s] == 1) { below[size_t(cut.first_child - first_node_below + inner_seen)] = { cut.first_child + inner_seen, cut.boundaries[c], cut.boundaries[c + 1] }; inner_seen++; }
				else              { order[cut.first_leaf + leaves_seen] = instance_at(cut.boundaries[c]); leaves_seen++; }
			}
		}
		level.swap(below);
	}
	return nodes_used;
}
