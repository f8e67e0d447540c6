// rt_tlas_build.h -- the per-node arithmetic of the device TLAS builder (kernels_build.hip).
//
// Replaces, for scenes whose instances move every frame, the host work of Integrator::build_tlas
// (reference Renderer/Integrators/Integrator.cpp:399-430: SAHBuilder over the mesh AABBs, BVH8Converter, re-ordering of
// the five per-instance tables) by one kernel launch. The output is the same data structure the traversal kernels read
// -- 80-byte CWBVH nodes (BVH8.h:19-25) whose leaves are runs of instances -- but not the same tree: instead of a
// top-down SAH sweep on one core the instances are sorted along a Morton curve and every node covers a contiguous run
// of that order, cut into up to eight runs at the highest differing Morton bits (the largest run is cut again until
// there are eight). Closest hits do not depend on the shape of the tree; tests/test_gpu_tlas.py checks that.
//
// Everything here is plain C++ shared by the kernel and by the CPU restatement of the same build in
// oracle/oracle_tlas.cpp (test infrastructure), which lets the node logic be checked without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#include <cmath>
#endif

#define RT_TLAS_BUILD_MAX 4096   // instances one launch sorts in LDS (32 KB of keys); larger scenes keep the host builder

struct TlasBox { float min[3], max[3]; };

RT_HD float tlas_minf(float a, float b) { return a < b ? a : b; }
RT_HD float tlas_maxf(float a, float b) { return a > b ? a : b; }

RT_HD void tlas_box_empty(TlasBox & b) { for (int d = 0; d < 3; d++) { b.min[d] = 3.0e38f; b.max[d] = -3.0e38f; } }
RT_HD void tlas_box_grow(TlasBox & b, const TlasBox & o) { for (int d = 0; d < 3; d++) { b.min[d] = tlas_minf(b.min[d], o.min[d]); b.max[d] = tlas_maxf(b.max[d], o.max[d]); } }

// World-space box of an instance: the 8 corners of the BLAS' object-space box through the 3x4 instance matrix (rows
// r0, r1, r2 of 4 floats each), as Mesh::calc_aabb does on the host.
RT_HD TlasBox tlas_world_box(const float * m, const float * local_min, const float * local_max) {
	TlasBox out; tlas_box_empty(out);
	for (int corner = 0; corner < 8; corner++) {
		float x = (corner & 1) ? local_max[0] : local_min[0];
		float y = (corner & 2) ? local_max[1] : local_min[1];
		float z = (corner & 4) ? local_max[2] : local_min[2];
		for (int d = 0; d < 3; d++) {
			float v = m[4 * d + 0] * x + m[4 * d + 1] * y + m[4 * d + 2] * z + m[4 * d + 3];
			out.min[d] = tlas_minf(out.min[d], v); out.max[d] = tlas_maxf(out.max[d], v);
		}
	}
	return out;
}

RT_HD uint32_t tlas_expand_bits(uint32_t v) { // 10 bits -> every third bit
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}

// 30-bit Morton code of a box centre inside the scene box
RT_HD uint32_t tlas_morton(const TlasBox & box, const TlasBox & scene) {
	uint32_t code = 0;
	for (int d = 0; d < 3; d++) {
		float extent = scene.max[d] - scene.min[d];
		float centre = 0.5f * (box.min[d] + box.max[d]);
		float t = extent > 0.0f ? (centre - scene.min[d]) / extent : 0.5f;
		t = tlas_minf(tlas_maxf(t * 1024.0f, 0.0f), 1023.0f);
		code |= tlas_expand_bits(uint32_t(t)) << (2 - d);
	}
	return code;
}

RT_HD int tlas_clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }   // (a compiler builtin on the host and on the device: v_ffbh_u32)

// Cuts the run [lo, hi) of the sorted keys (Morton code in the upper 32 bits) where the highest differing code bit
// flips; runs of identical codes are cut in the middle.
RT_HD int tlas_split(const uint64_t * keys, int lo, int hi) {
	uint32_t first = uint32_t(keys[lo] >> 32), last = uint32_t(keys[hi - 1] >> 32);
	if (first == last) return (lo + hi) / 2;
	uint32_t bit = 0x80000000u >> tlas_clz(first ^ last);
	int a = lo, b = hi - 1; // keys[a] has the bit clear, keys[b] has it set
	while (b - a > 1) {
		int mid = (a + b) / 2;
		if (uint32_t(keys[mid] >> 32) & bit) b = mid; else a = mid;
	}
	return b;
}

// Up to eight child runs of [lo, hi): the largest run is cut until there are eight or only single instances are left.
// Returns the number of runs; run c is [begin[c], begin[c + 1]). (Every index into begin[] is a loop counter of a loop with
// constant bounds: unrolled on the device the array lives in registers, not in scratch memory.)
RT_HD int tlas_child_runs(const uint64_t * keys, int lo, int hi, int begin[9]) {
	int count = 1;
	begin[0] = lo; begin[1] = hi;
	for (int c = 2; c < 9; c++) begin[c] = hi;
	for (int round = 0; round < 7; round++) {
		int widest = -1, width = 1, widest_lo = 0, widest_hi = 0;
		for (int c = 0; c < 8; c++) if (c < count && begin[c + 1] - begin[c] > width) { width = begin[c + 1] - begin[c]; widest = c; widest_lo = begin[c]; widest_hi = begin[c + 1]; }
		if (widest < 0) break;
		int cut = tlas_split(keys, widest_lo, widest_hi);
		for (int c = 8; c >= 1; c--) { if (c > widest + 1) begin[c] = begin[c - 1]; else if (c == widest + 1) begin[c] = cut; }
		count++;
	}
	return count;
}

// Children to octant slots: slot s is entered first by rays whose direction signs are s, so a child should sit in the
// slot whose diagonal points towards it -- the greedy assignment of the reference's converter (BVH8Converter.cpp:146-205).
RT_HD void tlas_assign_slots(const TlasBox & node, const TlasBox * children, int count, int slot_of_child[8]) {
	float cost[8][8];
	float centre[3];
	for (int d = 0; d < 3; d++) centre[d] = 0.5f * (node.min[d] + node.max[d]);
	for (int c = 0; c < count; c++) {
		float offset[3];
		for (int d = 0; d < 3; d++) offset[d] = 0.5f * (children[c].min[d] + children[c].max[d]) - centre[d];
		for (int s = 0; s < 8; s++) cost[c][s] = offset[0] * ((s & 4) ? -1.0f : 1.0f) + offset[1] * ((s & 2) ? -1.0f : 1.0f) + offset[2] * ((s & 1) ? -1.0f : 1.0f);
	}
	bool taken[8];
	for (int s = 0; s < 8; s++) taken[s] = false;
	for (int c = 0; c < 8; c++) slot_of_child[c] = -1;
	for (int round = 0; round < count; round++) {
		float best = 3.0e38f; int best_slot = -1, best_child = -1;
		for (int c = 0; c < count; c++) {
			if (slot_of_child[c] >= 0) continue;
			for (int s = 0; s < 8; s++) if (!taken[s] && cost[c][s] < best) { best = cost[c][s]; best_slot = s; best_child = c; }
		}
		if (best_slot < 0) break;
		taken[best_slot] = true; slot_of_child[best_child] = best_slot;
	}
	for (int c = 0; c < count; c++) if (slot_of_child[c] < 0) { int s = 0; while (taken[s]) s++; taken[s] = true; slot_of_child[c] = s; } // NaN boxes
}

// One CWBVH node (20 words = 5 x float4, BVH8.h:19-25). boxes / is_inner are in SLOT order (is_inner < 0: empty slot,
// 0: one instance, 1: inner node); inner children occupy consecutive node indices from base_child in slot order, the
// leaves consecutive instance positions from base_leaf in slot order.
RT_HD void tlas_encode_node(const TlasBox & node, const TlasBox boxes[8], const int is_inner[8], uint32_t base_child, uint32_t base_leaf, uint32_t out[20]) {
	for (int i = 0; i < 20; i++) out[i] = 0;
	float e[3], inv_e[3];
	uint32_t exponents = 0;
	for (int d = 0; d < 3; d++) {
		// smallest power of two e with extent / e <= 255 (BVH8Converter.cpp:229-253); a flat axis gets the smallest normal scale
		float extent = tlas_maxf(node.max[d] - node.min[d], 1.0e-30f);
		union { float f; uint32_t u; } scale;
		scale.f = extent * (1.0f / 255.0f);
		uint32_t biased = scale.u >> 23;
		if (scale.u & 0x7FFFFFu) biased++;              // not a power of two: round the exponent up
		if (biased < 1u) biased = 1u;
		if (biased > 254u) biased = 254u;
		scale.u = biased << 23;
		e[d] = scale.f; inv_e[d] = 1.0f / scale.f;
		exponents |= biased << (8 * d);
		union { float f; uint32_t u; } origin; origin.f = node.min[d];
		out[d] = origin.u;
	}
	uint32_t imask = 0, leaves = 0;
	uint8_t * meta = (uint8_t *)&out[6];
	uint8_t * q = (uint8_t *)&out[8];   // min_x[8] max_x[8] min_y[8] max_y[8] min_z[8] max_z[8]
	for (int s = 0; s < 8; s++) {
		if (is_inner[s] < 0) continue;
		for (int d = 0; d < 3; d++) {
			float lo = floorf((boxes[s].min[d] - node.min[d]) * inv_e[d]);
			float hi = ceilf ((boxes[s].max[d] - node.min[d]) * inv_e[d]);
			lo = tlas_minf(tlas_maxf(lo, 0.0f), 255.0f); hi = tlas_minf(tlas_maxf(hi, 0.0f), 255.0f);
			q[16 * d + s]     = uint8_t(lo);
			q[16 * d + 8 + s] = uint8_t(hi);
		}
		if (is_inner[s]) { meta[s] = uint8_t(0x20 | (24 + s)); imask |= 1u << s; }
		else             { meta[s] = uint8_t(0x20 | leaves); leaves++; }   // one instance: unary count 001, offset from base_leaf
	}
	(void)e;
	out[3] = exponents | (imask << 24);
	out[4] = base_child;
	out[5] = base_leaf;
}
