// oracle_tlas.cpp -- TEST INFRASTRUCTURE (never on the product path).
//
// CPU restatement of the device TLAS build (gpu-raytracer_amd/csrc/kernels_build.hip): the same phases -- instance
// boxes, Morton keys, sort, level-synchronous construction with prefix-sum numbering, table gather -- run by one thread,
// calling the very node arithmetic the kernel calls (rt_tlas_build.h is plain C++). The kernel must reproduce these bytes;
// tests/test_tlas.py checks the result itself (every instance in exactly one leaf, child boxes contain their instances,
// inner children consecutive) and that tracing it gives the hits of the host-built TLAS, whose builder is the
// reference's (Integrator.cpp:399-430), byte for byte.
#include "../gpu-raytracer_amd/csrc/rt_tlas_build.h"

#include <algorithm>
#include <cstring>
#include <vector>

extern "C" int oracle_tlas_build(const float * transforms /* 12 per instance */, const float * local_boxes /* 6 per instance */, int n,
                                 uint32_t * nodes /* 20 words x 2n */, int * order /* n */) {
	std::vector<TlasBox> boxes(static_cast<size_t>(n));
	TlasBox scene; tlas_box_empty(scene);
	for (int i = 0; i < n; i++) { boxes[size_t(i)] = tlas_world_box(transforms + 12 * size_t(i), local_boxes + 6 * size_t(i), local_boxes + 6 * size_t(i) + 3); tlas_box_grow(scene, boxes[size_t(i)]); }
	std::vector<uint64_t> keys(static_cast<size_t>(n));
	for (int i = 0; i < n; i++) keys[size_t(i)] = (uint64_t(tlas_morton(boxes[size_t(i)], scene)) << 32) | uint64_t(i);
	std::sort(keys.begin(), keys.end());

	struct Entry { int node, lo, hi; };
	std::vector<Entry> level; level.push_back({ 0, 0, n });
	int nodes_used = 1, leaves_used = 0;
	while (!level.empty()) {
		const int count = int(level.size());
		std::vector<int> runs(static_cast<size_t>(count) * 12), bases(static_cast<size_t>(count) * 2);
		for (int k = 0; k < count; k++) {
			int begin[9];
			int children = tlas_child_runs(keys.data(), level[size_t(k)].lo, level[size_t(k)].hi, begin);
			int inner = 0;
			for (int c = 0; c < children; c++) inner += begin[c + 1] - begin[c] > 1;
			for (int c = 0; c <= children; c++) runs[12 * size_t(k) + c] = begin[c];
			runs[12 * size_t(k) + 9] = children; runs[12 * size_t(k) + 10] = inner; runs[12 * size_t(k) + 11] = children - inner;
		}
		int node_base = nodes_used, leaf_base = leaves_used;
		for (int k = 0; k < count; k++) { bases[2 * size_t(k)] = node_base; bases[2 * size_t(k) + 1] = leaf_base; node_base += runs[12 * size_t(k) + 10]; leaf_base += runs[12 * size_t(k) + 11]; }
		const int first_child = nodes_used;
		std::vector<Entry> next(static_cast<size_t>(node_base - nodes_used));
		nodes_used = node_base; leaves_used = leaf_base;
		for (int k = 0; k < count; k++) {
			const int * r = &runs[12 * size_t(k)];
			const int children = r[9];
			TlasBox node; tlas_box_empty(node);
			TlasBox child_boxes[8];
			for (int c = 0; c < children; c++) {
				tlas_box_empty(child_boxes[c]);
				for (int i = r[c]; i < r[c + 1]; i++) tlas_box_grow(child_boxes[c], boxes[size_t(keys[size_t(i)] & 0xffffffffull)]);
				tlas_box_grow(node, child_boxes[c]);
			}
			int slot_of_child[8];
			tlas_assign_slots(node, child_boxes, children, slot_of_child);
			TlasBox slot_boxes[8]; int is_inner[8], child_of_slot[8];
			for (int s = 0; s < 8; s++) { is_inner[s] = -1; child_of_slot[s] = -1; }
			for (int c = 0; c < children; c++) { int s = slot_of_child[c]; slot_boxes[s] = child_boxes[c]; is_inner[s] = r[c + 1] - r[c] > 1; child_of_slot[s] = c; }
			tlas_encode_node(node, slot_boxes, is_inner, uint32_t(bases[2 * size_t(k)]), uint32_t(bases[2 * size_t(k) + 1]), nodes + 20 * size_t(level[size_t(k)].node));
			int next_inner = 0, next_leaf = 0;
			for (int s = 0; s < 8; s++) {
				int c = child_of_slot[s];
				if (c < 0) continue;
				if (is_inner[s]) { next[size_t(bases[2 * size_t(k)] - first_child + next_inner)] = { bases[2 * size_t(k)] + next_inner, r[c], r[c + 1] }; next_inner++; }
				else { order[bases[2 * size_t(k) + 1] + next_leaf] = int(keys[size_t(r[c])] & 0xffffffffull); next_leaf++; }
			}
		}
		level.swap(next);
	}
	return nodes_used;
}
