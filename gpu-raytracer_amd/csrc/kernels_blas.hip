// kernels_blas.hip -- bottom-level acceleration structures built ON THE DEVICE (the remainder).
//
// Replaces, for geometry that changes (deforming meshes, streamed assets) or simply has to be ready fast, what the
// reference does per mesh on host threads at load time: SAHBuilder (Src/BVH/Builders/SAHBuilder.cpp:13-104) followed by
// BVH8Converter (Src/BVH/Converters/BVH8Converter.cpp:7-335). The OUTPUT is the same data structure -- 80-byte CWBVH
// nodes (BVH8.h:19-25) over triangles stored in leaf order, up to 3 triangles per leaf, octant-ordered child slots,
// quantised child boxes -- which is all the traversal kernels know about; the TREE is not the reference's: a linear BVH.
//   1  box of every triangle (one thread each), box of every mesh (integer atomics on order-preserving keys, one set per wave)
//   2  64-bit keys: mesh << 32 | 30-bit Morton code of the triangle's box centre inside its mesh's box; ONE radix sort of all
//      triangles of all meshes (rocPRIM; a sort is a solved problem and not on the frame path)
//   3  breadth-first over ALL meshes at once, one level per round of four launches: a node covers a run of the sorted order;
//        runs     one thread per node cuts its run at the highest differing Morton bits -- the piece with the LARGEST SURFACE AREA again and
//                 again (the usual rule of a wide collapse: the box a ray is most likely to enter is the one worth refining; round 4, it was
//                 the widest piece) -- until there are 8 pieces or none holds more than 3 triangles (pieces of <= 3 triangles are leaves).
//                 A piece's box comes from a table of the boxes of all power-of-two runs of the sorted order (two look-ups per piece).
//        scan     exclusive prefix sums over the level: inner children -> node indices, leaf triangles -> triangle positions
//        boxes    one WAVE per child: the union of its run's triangle boxes (runs are long near the roots)
//        nodes    eight lanes per node: octant slots by the greedy assignment of the reference's converter
//                 (BVH8Converter.cpp:146-205, shared with the TLAS build: rt_tlas_build.h), quantisation, the node, the
//                 leaf order of its triangles, the ranges of its inner children = the next level
//   4  the 96-byte shading triangles and the 48-byte traversal copy are gathered into leaf order
// Closest hits do not depend on the shape of the tree (up to exact ties in t between different triangles, which a scene
// with duplicated geometry can have); tests/test_gpu_blas.py checks hits against the host-built trees and the node
// invariants with the independent decoder of tests/test_tlas.py. A Morton tree is cheaper to build and dearer to traverse
// than the SAH tree (measured: profiles/r03_device_blas.txt): the host classes use it on request (device_blas = 1).
#include "rt_math.h"
#include "rt_tlas_build.h"
#include <algorithm>

#include <cstring>   // (rocPRIM's texture iterator calls memset from host code without including it)
#include <rocprim/rocprim.hpp>

#define RT_BLAS_LEAF 3     // triangles per leaf (the meta byte holds their count in unary: BVH8Converter.cpp:293-303)

struct BlasBuildArgs {
	int triangle_count, mesh_count, first_node;     // first_node: index of mesh 0's root (node slots below it are reserved for the TLAS)
	const float4 * triangles;                        // input, 6 float4 each, meshes back to back
	const int * mesh_first;                          // [mesh_count + 1]
	float4 * triangles_out, * positions_out;         // leaf order: 6 resp. 3 float4 per triangle
	uint32_t * nodes;                                // 20 words per node
	int * order, * position;                         // leaf position -> input triangle, and back
	// scratch
	TlasBox * triangle_boxes, * sorted_boxes, * mesh_boxes, * child_boxes;
	TlasBox * box_table;                             // [table_levels][triangle_count]: entry (j, i) = box of sorted triangles [i, i + 2^(j+1)), clamped at the end (level 0 of the idea is sorted_boxes)
	int table_levels, split_widest;                  // split_widest: the round-3 rule (the piece with the most triangles first), for comparison
	int * triangle_mesh;
	uint64_t * keys; int * ids;                      // unsorted
	uint64_t * sorted_keys; int * sorted_ids;
	int2 * range;                                    // per node: its run [lo, hi) of the sorted order
	int * runs;                                      // per node of the level: begin[9], children, inner children, leaf triangles
	int * inner_count, * leaf_count, * inner_base, * leaf_base;   // per node of the level
	int * level_state;                               // { nodes used, triangles placed, nodes of the next level }
	const float * given_boxes; int given_first, given_count;   // rt_set_build_boxes: triangles [given_first, given_first + given_count) come with boxes of their own (6 floats: min, max), null / 0: none
};

// Step 1 in three launches, all of them as wide as the input: a mesh's box used to be reduced by ONE workgroup per mesh, which is the whole chip
// idle behind one CU when the "mesh" is a flattened scene of half a million triangles (2 of that build's 2.8 ms). Floats are accumulated as integers
// whose order is the floats' (the usual sign flip): atomicMin / atomicMax then do the reduction.
RT_DEV int blas_ordered(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
RT_DEV float blas_unordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void __launch_bounds__(256) kernel_blas_mesh_boxes_begin(BlasBuildArgs a) {
	const int mesh = blockIdx.x * blockDim.x + threadIdx.x;
	if (mesh >= a.mesh_count) return;
	int * box = (int *)&a.mesh_boxes[mesh];
	for (int d = 0; d < 3; d++) { box[d] = blas_ordered(3.0e38f); box[3 + d] = blas_ordered(-3.0e38f); }   // (tlas_box_empty)
}

__global__ void __launch_bounds__(256) kernel_blas_triangle_boxes(BlasBuildArgs a) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid = i < a.triangle_count;
	int mesh = 0;
	TlasBox box; tlas_box_empty(box);
	if (valid) {
		int lo = 0, hi = a.mesh_count;   // the mesh whose range [mesh_first[m], mesh_first[m + 1]) holds triangle i (empty meshes have empty ranges and are skipped)
		while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (a.mesh_first[mid] <= i) lo = mid; else hi = mid; }
		mesh = lo;
		const float4 * t = a.triangles + size_t(i) * 6;
		float4 t0 = t[0], t1 = t[1], t2 = t[2];
		const float p0[3] = { t0.x, t0.y, t0.z }, e1[3] = { t0.w, t1.x, t1.y }, e2[3] = { t1.z, t1.w, t2.x };
		// a REFERENCE: the triangle is a copy of one that the caller has cut into pieces (early split clipping, host/StaticBVHBuilder.cpp: presplit), and
		// this copy stands for the piece inside the given box -- the tree is built over the pieces' boxes, the leaf still tests the whole triangle
		const bool given = a.given_boxes && i >= a.given_first && i - a.given_first < a.given_count;
		const float * piece = given ? a.given_boxes + size_t(i - a.given_first) * 6 : nullptr;
		for (int d = 0; d < 3; d++) {
			float v1 = p0[d] + e1[d], v2 = p0[d] + e2[d];   // the vertices the traversal's Moeller-Trumbore test sees
			box.min[d] = fminf(p0[d], fminf(v1, v2)); box.max[d] = fmaxf(p0[d], fmaxf(v1, v2));
			if (given) { box.min[d] = fmaxf(box.min[d], piece[d]); box.max[d] = fminf(box.max[d], piece[3 + d]); if (box.max[d] < box.min[d]) box.max[d] = box.min[d]; }   // (never larger than the triangle's own)
			// no flat boxes: the node test is `tmin < tmax`, a box of zero thickness is never entered -- an axis-aligned wall would
			// be hit only where the quantisation grid happens to pad it. The reference's rule (AABB::fix_if_needed, AABB.h:27-38)
			float eps = 0.001f;
			while (box.max[d] - box.min[d] < eps) { box.min[d] -= eps; box.max[d] += eps; eps *= 2.0f; }
		}
		a.triangle_boxes[i] = box; a.triangle_mesh[i] = mesh;
	}
	// the mesh's box: one set of atomics per wave where the whole wave is inside one mesh (all but the waves that straddle a boundary). Every lane of the
	// wave stays in the reduction -- the ones beyond the input hold the empty box -- so that no shuffle reads a lane that has left
	const unsigned long long active = __ballot(valid);
	if (active == 0ull) return;   // (the whole wave)
	const int leader = __ffsll((long long)active) - 1;
	const int first_mesh = __shfl(mesh, leader);
	const bool uniform = __ballot(valid && mesh == first_mesh) == active;
	int * target = (int *)&a.mesh_boxes[valid ? mesh : first_mesh];
	if (uniform) {
		#pragma unroll
		for (int d = 0; d < 3; d++) {
			float lo = box.min[d], hi = box.max[d];
			for (int offset = 32; offset > 0; offset >>= 1) { lo = fminf(lo, __shfl_xor(lo, offset)); hi = fmaxf(hi, __shfl_xor(hi, offset)); }
			if (int(threadIdx.x & 63) == leader) { atomicMin(&target[d], blas_ordered(lo)); atomicMax(&target[3 + d], blas_ordered(hi)); }
		}
	} else if (valid) {
		for (int d = 0; d < 3; d++) { atomicMin(&target[d], blas_ordered(box.min[d])); atomicMax(&target[3 + d], blas_ordered(box.max[d])); }
	}
}

__global__ void __launch_bounds__(256) kernel_blas_mesh_boxes_end(BlasBuildArgs a) {
	const int mesh = blockIdx.x * blockDim.x + threadIdx.x;
	if (mesh >= a.mesh_count) return;
	int * box = (int *)&a.mesh_boxes[mesh];
	for (int d = 0; d < 6; d++) box[d] = __float_as_int(blas_unordered(box[d]));
	a.range[a.first_node + mesh] = make_int2(a.mesh_first[mesh], a.mesh_first[mesh + 1]);   // the root of the mesh covers all of it
}

__global__ void __launch_bounds__(256) kernel_blas_keys(BlasBuildArgs a) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.triangle_count) return;
	const int mesh = a.triangle_mesh[i];
	a.keys[i] = (uint64_t(uint32_t(mesh)) << 32) | uint64_t(tlas_morton(a.triangle_boxes[i], a.mesh_boxes[mesh]));
	a.ids[i] = i;
}

__global__ void __launch_bounds__(256) kernel_blas_sorted_boxes(BlasBuildArgs a) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < a.triangle_count) a.sorted_boxes[i] = a.triangle_boxes[a.sorted_ids[i]];
}

// boxes of the runs [i, i + 2^level) of the sorted order, level = 1 .. table_levels, each from the two halves one level below
__global__ void __launch_bounds__(256) kernel_blas_box_table(BlasBuildArgs a, int level) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.triangle_count) return;
	const TlasBox * below = level == 1 ? a.sorted_boxes : a.box_table + size_t(level - 2) * a.triangle_count;
	TlasBox box = below[i];
	const int other = i + (1 << (level - 1));
	if (other < a.triangle_count) tlas_box_grow(box, below[other]);
	a.box_table[size_t(level - 1) * a.triangle_count + i] = box;
}
// surface area of the box of the sorted triangles [lo, hi), hi > lo: the union of the two (overlapping) power-of-two runs that cover it
RT_DEV float blas_run_area(const BlasBuildArgs & a, int lo, int hi) {
	const int k = 31 - __clz(hi - lo);
	const TlasBox * runs = k == 0 ? a.sorted_boxes : a.box_table + size_t(k - 1) * a.triangle_count;
	TlasBox box = runs[lo];
	tlas_box_grow(box, runs[hi - (1 << k)]);
	const float dx = box.max[0] - box.min[0], dy = box.max[1] - box.min[1], dz = box.max[2] - box.min[2];
	return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// [lo, hi) cut where the highest differing Morton bit flips; equal codes: in the middle
RT_DEV int blas_split(const uint64_t * __restrict__ keys, int lo, int hi) {
	const uint32_t first = uint32_t(keys[lo]), last = uint32_t(keys[hi - 1]);
	if (first == last) return (lo + hi) / 2;
	const uint32_t bit = 0x80000000u >> __clz(int(first ^ last));
	int below = lo, above = hi - 1;   // keys[below] has the bit clear, keys[above] has it set
	while (above - below > 1) {
		int mid = (below + above) / 2;
		if (uint32_t(keys[mid]) & bit) above = mid; else below = mid;
	}
	return above;
}

__global__ void __launch_bounds__(256) kernel_blas_runs(BlasBuildArgs a, int level_first, int level_nodes) {
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= level_nodes) return;
	const int2 range = a.range[level_first + k];
	int begin[9];
	begin[0] = range.x;
	#pragma unroll
	for (int c = 1; c < 9; c++) begin[c] = range.y;
	int count = range.y > range.x ? 1 : 0;
	// what a piece weighs when the next one to cut is chosen: its surface area (or, split_widest, its number of triangles)
	float weight[8];
	#pragma unroll
	for (int c = 0; c < 8; c++) weight[c] = 0.0f;
	if (range.y - range.x > RT_BLAS_LEAF) weight[0] = a.split_widest ? float(range.y - range.x) : blas_run_area(a, range.x, range.y);
	for (int round = 0; round < 7 && count > 0; round++) {
		int widest = -1, widest_lo = 0, widest_hi = 0; float heaviest = -1.0f;
		#pragma unroll
		for (int c = 0; c < 8; c++) if (c < count && begin[c + 1] - begin[c] > RT_BLAS_LEAF && !(weight[c] <= heaviest)) {   /* (written so that a piece whose area is NaN -- non-finite vertices -- is still cut: `>` would skip it for ever) */ heaviest = weight[c]; widest = c; widest_lo = begin[c]; widest_hi = begin[c + 1]; }
		if (widest < 0) break;
		const int cut = blas_split(a.sorted_keys, widest_lo, widest_hi);
		float left = float(cut - widest_lo), right = float(widest_hi - cut);
		if (!a.split_widest) {   // (a piece of <= 3 triangles is never cut again: its weight is not looked at)
			left  = cut - widest_lo > RT_BLAS_LEAF ? blas_run_area(a, widest_lo, cut) : 0.0f;
			right = widest_hi - cut > RT_BLAS_LEAF ? blas_run_area(a, cut, widest_hi) : 0.0f;
		}
		#pragma unroll
		for (int c = 8; c >= 1; c--) { if (c > widest + 1) begin[c] = begin[c - 1]; else if (c == widest + 1) begin[c] = cut; }
		#pragma unroll
		for (int c = 7; c >= 0; c--) { if (c > widest + 1) weight[c] = weight[c - 1]; else if (c == widest + 1) weight[c] = right; else if (c == widest) weight[c] = left; }
		count++;
	}
	int inner = 0, leaf_triangles = 0;
	#pragma unroll
	for (int c = 0; c < 8; c++) if (c < count) { int n = begin[c + 1] - begin[c]; if (n > RT_BLAS_LEAF) inner++; else leaf_triangles += n; }
	int * r = a.runs + 12 * size_t(k);
	#pragma unroll
	for (int c = 0; c < 9; c++) r[c] = begin[c];
	r[9] = count; r[10] = inner; r[11] = leaf_triangles;
	a.inner_count[k] = inner; a.leaf_count[k] = leaf_triangles;
}

// one wave per (node, child): the box of the child's run
__global__ void __launch_bounds__(256) kernel_blas_child_boxes(BlasBuildArgs a, int level_nodes) {
	const int pair = blockIdx.x * 4 + int(threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (pair >= level_nodes * 8) return;
	const int k = pair >> 3, c = pair & 7;
	const int * r = a.runs + 12 * size_t(k);
	TlasBox box; tlas_box_empty(box);
	if (c < r[9]) for (int i = r[c] + lane; i < r[c + 1]; i += 64) tlas_box_grow(box, a.sorted_boxes[i]);
	#pragma unroll
	for (int d = 0; d < 3; d++) {
		for (int offset = 32; offset > 0; offset >>= 1) { box.min[d] = fminf(box.min[d], __shfl_xor(box.min[d], offset)); box.max[d] = fmaxf(box.max[d], __shfl_xor(box.max[d], offset)); }
	}
	if (lane == 0) a.child_boxes[pair] = box;
}

RT_DEV float blas_group8_fmin(float v) { v = fminf(v, __shfl_xor(v, 1)); v = fminf(v, __shfl_xor(v, 2)); return fminf(v, __shfl_xor(v, 4)); }
RT_DEV float blas_group8_fmax(float v) { v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); return fmaxf(v, __shfl_xor(v, 4)); }
RT_DEV unsigned blas_group8_bits(unsigned v) { v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); return v | __shfl_xor(v, 4); }

// eight lanes per node, lane c = child c (run order); see kernels_build.hip step 3c for the slot assignment
__global__ void __launch_bounds__(256) kernel_blas_nodes(BlasBuildArgs a, int level_first, int level_nodes, int next_level_first, int triangles_before) {
	__shared__ uint32_t node_stage[32][20];
	const int tid = threadIdx.x, lane = tid & 63, group = tid >> 3, c = tid & 7, group_lane0 = lane & ~7;
	const int k = blockIdx.x * 32 + group;
	const bool node_exists = k < level_nodes;   // (no early return: the shuffles below want whole groups, and a wave holds eight of them)
	const int * r = a.runs + 12 * size_t(node_exists ? k : 0);
	const int children = node_exists ? r[9] : 0;
	const bool valid = c < children;
	const int run_lo = valid ? r[c] : 0, run_hi = valid ? r[c + 1] : 0, run_triangles = run_hi - run_lo;
	TlasBox box; tlas_box_empty(box);
	if (valid) box = a.child_boxes[size_t(k) * 8 + c];
	TlasBox node;
	#pragma unroll
	for (int d = 0; d < 3; d++) { node.min[d] = blas_group8_fmin(box.min[d]); node.max[d] = blas_group8_fmax(box.max[d]); }

	float cost[8];
	{
		float offset[3];
		#pragma unroll
		for (int d = 0; d < 3; d++) offset[d] = 0.5f * (box.min[d] + box.max[d]) - 0.5f * (node.min[d] + node.max[d]);
		#pragma unroll
		for (int s = 0; s < 8; s++) cost[s] = offset[0] * ((s & 4) ? -1.0f : 1.0f) + offset[1] * ((s & 2) ? -1.0f : 1.0f) + offset[2] * ((s & 1) ? -1.0f : 1.0f);
	}
	int my_slot = -1; unsigned taken = 0;
	for (int round = 0; round < 8; round++) {   // (uniform trip count over the wave; groups with fewer children idle through the rest)
		float best = 3.0e38f; int best_slot = -1;
		if (valid && my_slot < 0 && round < children) {
			#pragma unroll
			for (int s = 0; s < 8; s++) if (!((taken >> s) & 1u) && cost[s] < best) { best = cost[s]; best_slot = s; }
		}
		float offer = best; int offer_child = best_slot >= 0 ? c : 8, offer_slot = best_slot;
		#pragma unroll
		for (int step = 1; step < 8; step <<= 1) {
			float other = __shfl_xor(offer, step); int other_child = __shfl_xor(offer_child, step), other_slot = __shfl_xor(offer_slot, step);
			bool take = other_child < 8 && (offer_child >= 8 || other < offer || (other == offer && other_child < offer_child));
			if (take) { offer = other; offer_child = other_child; offer_slot = other_slot; }
		}
		if (offer_child < 8) { if (c == offer_child) my_slot = offer_slot; taken |= 1u << offer_slot; }
	}
	for (int child = 0; child < 8; child++) {   // whatever the comparisons left out (NaN boxes) takes the free slots in child order
		int slot_of_that_child = __shfl(my_slot, group_lane0 + child);
		if (child >= children || slot_of_that_child >= 0) continue;
		int free_slot = __ffs(int(~taken & 0xffu)) - 1;
		if (c == child) my_slot = free_slot;
		taken |= 1u << free_slot;
	}

	const bool inner = valid && run_triangles > RT_BLAS_LEAF, leaf = valid && !inner;
	const unsigned inner_mask = blas_group8_bits(inner ? 1u << my_slot : 0u);
	// triangles of the leaves in front of mine, in slot order
	int triangles_in_front = 0;
	#pragma unroll
	for (int other = 0; other < 8; other++) {
		int other_slot = __shfl(my_slot, group_lane0 + other), other_triangles = __shfl(leaf ? run_triangles : 0, group_lane0 + other);
		if (other_slot >= 0 && other_slot < my_slot) triangles_in_front += other_triangles;
	}
	const int node_base = node_exists ? next_level_first + a.inner_base[k] : 0, leaf_base = node_exists ? triangles_before + a.leaf_base[k] : 0;

	uint32_t header[4]; uint32_t exponents = 0; float inv_e[3];
	#pragma unroll
	for (int d = 0; d < 3; d++) {
		float extent = tlas_maxf(node.max[d] - node.min[d], 1.0e-30f);
		uint32_t bits = __float_as_uint(extent * (1.0f / 255.0f));
		uint32_t biased = bits >> 23;
		if (bits & 0x7FFFFFu) biased++;
		if (biased < 1u) biased = 1u;
		if (biased > 254u) biased = 254u;
		inv_e[d] = 1.0f / __uint_as_float(biased << 23);
		exponents |= biased << (8 * d);
		header[d] = __float_as_uint(children > 0 ? node.min[d] : 0.0f);
	}
	header[3] = exponents | (inner_mask << 24);
	volatile uint32_t * stage = node_stage[group];
	#pragma unroll
	for (int j = 0; j < 3; j++) {
		const int w = c + 8 * j;
		if (w < 20) stage[w] = w == 0 ? header[0] : w == 1 ? header[1] : w == 2 ? header[2] : w == 3 ? header[3] : w == 4 ? uint32_t(node_base) : w == 5 ? uint32_t(leaf_base) : 0u;
	}
	if (valid) {
		volatile uint8_t * bytes = (volatile uint8_t *)stage;
		const unsigned unary = (1u << run_triangles) - 1u;   // 1, 3, 7 for 1, 2, 3 triangles
		bytes[24 + my_slot] = uint8_t(inner ? (0x20 | (24 + my_slot)) : ((unary << 5) | unsigned(triangles_in_front)));
		#pragma unroll
		for (int d = 0; d < 3; d++) {
			float lo = floorf((box.min[d] - node.min[d]) * inv_e[d]);
			float hi = ceilf ((box.max[d] - node.min[d]) * inv_e[d]);
			lo = tlas_minf(tlas_maxf(lo, 0.0f), 255.0f); hi = tlas_minf(tlas_maxf(hi, 0.0f), 255.0f);
			bytes[32 + 16 * d + my_slot]     = uint8_t(lo);
			bytes[32 + 16 * d + 8 + my_slot] = uint8_t(hi);
		}
	}
	if (!node_exists) return;
	uint32_t * dst = a.nodes + 20 * size_t(level_first + k);
	#pragma unroll
	for (int j = 0; j < 3; j++) { const int w = c + 8 * j; if (w < 20) dst[w] = stage[w]; }

	if (inner) {
		a.range[node_base + __popc(inner_mask & ((1u << my_slot) - 1u))] = make_int2(run_lo, run_hi);
	} else if (leaf) {
		for (int j = 0; j < run_triangles; j++) {
			const int source = a.sorted_ids[run_lo + j], place = leaf_base + triangles_in_front + j;
			a.order[place] = source; a.position[source] = place;
		}
	}
}

// after the scans of a level: how many nodes the next level has, running totals
__global__ void kernel_blas_level_totals(BlasBuildArgs a, int level_nodes) {
	const int inner = a.inner_base[level_nodes - 1] + a.inner_count[level_nodes - 1], leaves = a.leaf_base[level_nodes - 1] + a.leaf_count[level_nodes - 1];
	a.level_state[0] += inner; a.level_state[1] += leaves; a.level_state[2] = inner;
}

__global__ void __launch_bounds__(256) kernel_blas_gather(BlasBuildArgs a) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per float4 of the output triangles
	if (i >= a.triangle_count * 6) return;
	const int place = i / 6, part = i % 6;
	const float4 v = a.triangles[size_t(a.order[place]) * 6 + part];
	a.triangles_out[i] = v;
	if (part < 3) a.positions_out[size_t(place) * 3 + part] = part < 2 ? v : make_float4(v.x, 0.0f, 0.0f, 0.0f);   // position_0, edge_1, edge_2 (+ padding)
}

// Host side of the build: launches, the sort, the per-level scans and the one number per level the host has to know (how
// many nodes the next level has). `scratch` holds everything BlasBuildArgs names besides the outputs. Returns the node count.
size_t rt_blas_build_scratch_bytes(size_t triangles, size_t meshes) {
	size_t sort_bytes = 0, scan_bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int *)nullptr, (int *)nullptr, triangles, 0, 64);
	(void)rocprim::exclusive_scan(nullptr, scan_bytes, (int *)nullptr, (int *)nullptr, 0, triangles + meshes, rocprim::plus<int>());
	return sort_bytes + scan_bytes + 512;
}

hipError_t rt_blas_build(BlasBuildArgs a, void * library_scratch, size_t library_scratch_bytes, int * pinned_state, hipStream_t stream, int * out_node_count, size_t node_capacity) {
	const int T = a.triangle_count, M = a.mesh_count;
	hipError_t e;
	if ((e = hipMemsetAsync(a.level_state, 0, 3 * sizeof(int), stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(kernel_blas_mesh_boxes_begin, dim3((M + 255) / 256), dim3(256), 0, stream, a);
	if (T > 0) hipLaunchKernelGGL(kernel_blas_triangle_boxes, dim3((T + 255) / 256), dim3(256), 0, stream, a);
	hipLaunchKernelGGL(kernel_blas_mesh_boxes_end, dim3((M + 255) / 256), dim3(256), 0, stream, a);
	if (T > 0) {
		hipLaunchKernelGGL(kernel_blas_keys, dim3((T + 255) / 256), dim3(256), 0, stream, a);
		int mesh_bits = 1; while ((1ll << mesh_bits) < M) mesh_bits++;
		size_t bytes = library_scratch_bytes;
		if ((e = rocprim::radix_sort_pairs(library_scratch, bytes, a.keys, a.sorted_keys, a.ids, a.sorted_ids, size_t(T), 0, 32 + mesh_bits, stream)) != hipSuccess) return e;
		hipLaunchKernelGGL(kernel_blas_sorted_boxes, dim3((T + 255) / 256), dim3(256), 0, stream, a);
		if (!a.split_widest) for (int level = 1; level <= a.table_levels; level++) hipLaunchKernelGGL(kernel_blas_box_table, dim3((T + 255) / 256), dim3(256), 0, stream, a, level);
	}
	int level_first = a.first_node, level_nodes = M, nodes_used = a.first_node + M, triangles_placed = 0;
	int levels = 0;
	while (level_nodes > 0) {
		// A level writes its own nodes and the ranges of up to 8 children per node behind them; a build that would leave the arrays, or
		// that goes on for more levels than 30 Morton bits + the halving of equal keys can give, is given up (bad input) instead of followed.
		// (an inner child holds more than RT_BLAS_LEAF triangles and the children of a level are disjoint: at most T / 4 of them)
		if (size_t(nodes_used) + std::min(size_t(level_nodes) * 8, size_t(T) / (RT_BLAS_LEAF + 1) + 1) > node_capacity || ++levels > 192) return hipErrorInvalidValue;
		hipLaunchKernelGGL(kernel_blas_runs, dim3((level_nodes + 255) / 256), dim3(256), 0, stream, a, level_first, level_nodes);
		size_t bytes = library_scratch_bytes;
		if ((e = rocprim::exclusive_scan(library_scratch, bytes, a.inner_count, a.inner_base, 0, size_t(level_nodes), rocprim::plus<int>(), stream)) != hipSuccess) return e;
		bytes = library_scratch_bytes;
		if ((e = rocprim::exclusive_scan(library_scratch, bytes, a.leaf_count, a.leaf_base, 0, size_t(level_nodes), rocprim::plus<int>(), stream)) != hipSuccess) return e;
		hipLaunchKernelGGL(kernel_blas_level_totals, dim3(1), dim3(1), 0, stream, a, level_nodes);
		hipLaunchKernelGGL(kernel_blas_child_boxes, dim3((level_nodes * 8 + 3) / 4), dim3(256), 0, stream, a, level_nodes);
		hipLaunchKernelGGL(kernel_blas_nodes, dim3((level_nodes + 31) / 32), dim3(256), 0, stream, a, level_first, level_nodes, nodes_used, triangles_placed);
		if ((e = hipMemcpyAsync(pinned_state, a.level_state, 3 * sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
		if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
		level_first = nodes_used; level_nodes = pinned_state[2];
		nodes_used = a.first_node + M + pinned_state[0]; triangles_placed = pinned_state[1];
	}
	if (T > 0) hipLaunchKernelGGL(kernel_blas_gather, dim3((T * 6 + 255) / 256), dim3(256), 0, stream, a);
	*out_node_count = nodes_used;
	return hipGetLastError();
}
