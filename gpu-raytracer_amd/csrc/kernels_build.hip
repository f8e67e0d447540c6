// kernels_build.hip -- the per-frame TLAS build on the device.
//
// Replaces Integrator::build_tlas of the reference (Renderer/Integrators/Integrator.cpp:399-430: SAH build over the
// instance boxes, CWBVH conversion, re-ordering of the per-instance tables -- the only CPU work inside the frame loop of
// an animated scene) by ONE launch of ONE workgroup: a TLAS has hundreds to a few thousand leaves, far too few to need
// the machine, and one workgroup can keep the whole sort in LDS and separate its phases with barriers instead of
// launches. The per-node arithmetic is in rt_tlas_build.h; this file is the parallel schedule around it:
//
//   1  world box of every instance (its BLAS' object-space box through the instance matrix), scene box by reduction
//   2  64-bit keys (30-bit Morton code of the box centre << 32 | instance) sorted by a bitonic network in LDS; the boxes
//      once more in sorted order (in LDS up to 1 024 instances)
//   3  breadth-first, level by level: every node covers a run of the sorted order;
//        a) one thread per node cuts its run into up to 8 child runs (single instances become leaves)
//        b) the per-node counts become node indices / instance positions by a workgroup-wide exclusive scan (the numbering
//           is that of a serial build, bit for bit -- the CPU restatement in oracle/oracle_tlas.cpp produces the same bytes)
//        c) EIGHT LANES per node, one per child: the lane unions the boxes of its child's run, the group reduces the node
//           box, assigns octant slots (the serial greedy assignment, round by round: every lane offers its cheapest free
//           slot, the group takes the cheapest offer), quantises, assembles the 80-byte node in LDS and writes it with
//           the order of its leaf instances and its inner children's entries of the next level's queue
//   4  the five per-instance tables are gathered into TLAS order, and position[scene index] is written for the light
//      tables (which name instances by scene index when the TLAS is built here)
//
// The first version ran step 3c with one THREAD per node on arrays indexed by loop variables (8 x 8 costs, the child
// boxes): they lived in scratch memory and one node took 30 us -- 0.43-0.53 ms of the kernel's 0.75 ms at 443 instances
// (diagnostic build RT_TLAS_PHASE_TIMES, profiles/r02_device_tlas_phases.txt). Everything a lane touches now has a
// compile-time index.
#include "rt_math.h"
#include "rt_tlas_build.h"

#define RT_BUILD_THREADS 1024
#define RT_BUILD_SMALL   1024   // instances up to which 256 threads build and the sorted boxes live in LDS

struct TlasBuildArgs {
	int count;                       // instances
	// scene order inputs
	const int    * root_indices;     // BLAS root | identity flag << 31
	const int    * material_ids;
	const float4 * transforms, * transforms_inv, * transforms_prev; // 3 float4 per instance
	const float  * local_boxes;      // 6 floats per instance: object-space min, max of its BLAS
	// TLAS order outputs
	uint32_t * nodes;                // 20 words per node, capacity 2 * count nodes
	int      * out_root_indices, * out_material_ids;
	float4   * out_transforms, * out_transforms_inv, * out_transforms_prev;
	int      * order;                // position -> scene index
	int      * position;             // scene index -> position
	int      * node_count;
	// scratch (global): boxes[count], queues 2 x count x {node, lo, hi}, per level-node records
	TlasBox  * boxes, * sorted_boxes;
	int      * queue;                // [2][count][3]
	int      * runs;                 // [count][12]: begin[9], children, inner children, leaf children
	int      * bases;                // [count][2]: first child node index, first leaf position
	TlasBox  * child_boxes;          // (unused since the lane-per-child build; kept for the layout of the scratch area)
};

// min / max / or over the 8 lanes of a group (xor shuffles stay inside the group)
RT_DEV float group8_fmin(float v) { v = tlas_minf(v, __shfl_xor(v, 1)); v = tlas_minf(v, __shfl_xor(v, 2)); return tlas_minf(v, __shfl_xor(v, 4)); }
RT_DEV float group8_fmax(float v) { v = tlas_maxf(v, __shfl_xor(v, 1)); v = tlas_maxf(v, __shfl_xor(v, 2)); return tlas_maxf(v, __shfl_xor(v, 4)); }
RT_DEV unsigned group8_bits(unsigned v) { v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); return v | __shfl_xor(v, 4); }

__global__ void __launch_bounds__(RT_BUILD_THREADS) kernel_build_tlas(TlasBuildArgs a) {
	__shared__ uint64_t lds_raw[5120];   // 40 KB: the keys (padded count), behind them the sorted boxes of a small scene
	__shared__ uint32_t node_stage[RT_BUILD_THREADS / 8][20];
	__shared__ float reduce[RT_BUILD_THREADS / RT_WAVE_SIZE][6];
	__shared__ int wave_totals[RT_BUILD_THREADS / RT_WAVE_SIZE][2];
	__shared__ TlasBox scene;
	__shared__ int level_count, next_count, nodes_used, leaves_used, scan_carry[2];
	const int tid = threadIdx.x, n = a.count;
	const int threads = int(blockDim.x);   // 256 for small scenes (cheaper barriers), RT_BUILD_THREADS beyond
	const int lane = tid & (RT_WAVE_SIZE - 1), wave = tid / RT_WAVE_SIZE;
	uint64_t * keys = lds_raw;
#ifdef RT_TLAS_PHASE_TIMES   // diagnostic build: where the time of the one workgroup goes (100 MHz wall clock, thread 0)
	unsigned long long stamp[8] = { }; unsigned long long t_prev = wall_clock64();
	#define RT_TLAS_STAMP(k) { unsigned long long now = wall_clock64(); stamp[k] += now - t_prev; t_prev = now; }
#else
	#define RT_TLAS_STAMP(k)
#endif

	// ---- 1: instance boxes, scene box
	TlasBox mine; tlas_box_empty(mine);
	for (int i = tid; i < n; i += threads) {
		TlasBox box = tlas_world_box((const float *)(a.transforms + 3 * size_t(i)), a.local_boxes + 6 * size_t(i), a.local_boxes + 6 * size_t(i) + 3);
		a.boxes[i] = box;
		tlas_box_grow(mine, box);
	}
	for (int d = 0; d < 3; d++) {
		float lo = mine.min[d], hi = mine.max[d];
		for (int offset = 32; offset > 0; offset >>= 1) { lo = fminf(lo, __shfl_xor(lo, offset)); hi = fmaxf(hi, __shfl_xor(hi, offset)); }
		if ((tid & 63) == 0) { reduce[tid >> 6][d] = lo; reduce[tid >> 6][3 + d] = hi; }
	}
	__syncthreads();
	if (tid == 0) {
		tlas_box_empty(scene);
		for (int w = 0; w < threads / RT_WAVE_SIZE; w++) for (int d = 0; d < 3; d++) { scene.min[d] = fminf(scene.min[d], reduce[w][d]); scene.max[d] = fmaxf(scene.max[d], reduce[w][3 + d]); }
	}
	__syncthreads();
	RT_TLAS_STAMP(0)

	// ---- 2: Morton keys, bitonic sort (padded with the largest key to a power of two)
	int padded = 1; while (padded < n) padded <<= 1;
	for (int i = tid; i < padded; i += threads) keys[i] = i < n ? (uint64_t(tlas_morton(a.boxes[i], scene)) << 32) | uint64_t(i) : ~0ull;
	__syncthreads();
	for (int size = 2; size <= padded; size <<= 1) {
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			for (int i = tid; i < padded; i += threads) {
				int partner = i ^ stride;
				if (partner > i) {
					bool ascending = (i & size) == 0;
					uint64_t x = keys[i], y = keys[partner];
					if ((x > y) == ascending) { keys[i] = y; keys[partner] = x; }
				}
			}
			__syncthreads();
		}
	}

	// the boxes once more in sorted order: the runs of step 3c read consecutive memory instead of chasing the keys
	const bool boxes_in_lds = n <= RT_BUILD_SMALL;
	TlasBox * sorted = boxes_in_lds ? (TlasBox *)(lds_raw + padded) : a.sorted_boxes;
	for (int i = tid; i < n; i += threads) sorted[i] = a.boxes[int(keys[i] & 0xffffffffull)];
	__syncthreads();
	RT_TLAS_STAMP(1)

	// ---- 3: breadth-first build
	int * queue[2] = { a.queue, a.queue + 3 * size_t(n) };
	if (tid == 0) { level_count = 1; nodes_used = 1; leaves_used = 0; queue[0][0] = 0; queue[0][1] = 0; queue[0][2] = n; }
	__syncthreads();
	for (int level = 0; level_count > 0; level++) {
		const int * in = queue[level & 1]; int * out = queue[(level + 1) & 1];
		const int count = level_count;
		// a) child runs
		for (int k = tid; k < count; k += threads) {
			int begin[9];
			int children = tlas_child_runs(keys, in[3 * k + 1], in[3 * k + 2], begin);
			int inner = 0;
			#pragma unroll
			for (int c = 0; c < 8; c++) inner += (c < children && begin[c + 1] - begin[c] > 1) ? 1 : 0;
			int * r = a.runs + 12 * size_t(k);
			#pragma unroll
			for (int c = 0; c < 9; c++) r[c] = begin[c];
			r[9] = children; r[10] = inner; r[11] = children - inner;
		}
		if (tid == 0) { scan_carry[0] = nodes_used; scan_carry[1] = leaves_used; }
		__syncthreads();
		RT_TLAS_STAMP(2)
		// b) numbering: exclusive prefix sums of the inner / leaf child counts over the nodes of the level, in queue order
		for (int base = 0; base < count; base += threads) {
			const int k = base + tid;
			const int mine_inner = k < count ? a.runs[12 * size_t(k) + 10] : 0, mine_leaf = k < count ? a.runs[12 * size_t(k) + 11] : 0;
			int sum_inner = mine_inner, sum_leaf = mine_leaf;   // inclusive scan within the wave
			#pragma unroll
			for (int offset = 1; offset < RT_WAVE_SIZE; offset <<= 1) {
				int up_inner = __shfl_up(sum_inner, offset), up_leaf = __shfl_up(sum_leaf, offset);
				if (lane >= offset) { sum_inner += up_inner; sum_leaf += up_leaf; }
			}
			if (lane == RT_WAVE_SIZE - 1) { wave_totals[wave][0] = sum_inner; wave_totals[wave][1] = sum_leaf; }
			__syncthreads();
			int before_inner = scan_carry[0], before_leaf = scan_carry[1];
			for (int w = 0; w < wave; w++) { before_inner += wave_totals[w][0]; before_leaf += wave_totals[w][1]; }
			if (k < count) { a.bases[2 * k] = before_inner + sum_inner - mine_inner; a.bases[2 * k + 1] = before_leaf + sum_leaf - mine_leaf; }
			__syncthreads();
			if (tid == threads - 1) { scan_carry[0] = before_inner + sum_inner; scan_carry[1] = before_leaf + sum_leaf; }
			__syncthreads();
		}
		if (tid == 0) { next_count = scan_carry[0] - nodes_used; nodes_used = scan_carry[0]; leaves_used = scan_carry[1]; }
		__syncthreads();
		RT_TLAS_STAMP(3)
		// c) eight lanes per node: child boxes, node box, slots, the node, leaf order, next level
		{
			const int group = tid >> 3, groups = threads >> 3, c = tid & 7, group_lane0 = lane & ~7;
			const int first_node_of_next_level = nodes_used - next_count;
			volatile uint32_t * stage = node_stage[group];
			for (int k = group; k < count; k += groups) {
				const int * r = a.runs + 12 * size_t(k);
				const int children = r[9];
				const bool valid = c < children;
				const int run_lo = valid ? r[c] : 0, run_hi = valid ? r[c + 1] : 0;
				TlasBox box; tlas_box_empty(box);
				for (int i = run_lo; i < run_hi; i++) tlas_box_grow(box, sorted[i]);
				TlasBox node;
				#pragma unroll
				for (int d = 0; d < 3; d++) { node.min[d] = group8_fmin(box.min[d]); node.max[d] = group8_fmax(box.max[d]); }

				// octant slots: the greedy assignment of tlas_assign_slots, its (child, slot) scan order kept by the tie rules
				float cost[8];
				{
					float offset[3];
					#pragma unroll
					for (int d = 0; d < 3; d++) offset[d] = 0.5f * (box.min[d] + box.max[d]) - 0.5f * (node.min[d] + node.max[d]);
					#pragma unroll
					for (int s = 0; s < 8; s++) cost[s] = offset[0] * ((s & 4) ? -1.0f : 1.0f) + offset[1] * ((s & 2) ? -1.0f : 1.0f) + offset[2] * ((s & 1) ? -1.0f : 1.0f);
				}
				int my_slot = -1; unsigned taken = 0;
				for (int round = 0; round < children; round++) {
					float best = 3.0e38f; int best_slot = -1;
					if (valid && my_slot < 0) {
						#pragma unroll
						for (int s = 0; s < 8; s++) if (!((taken >> s) & 1u) && cost[s] < best) { best = cost[s]; best_slot = s; }
					}
					// the cheapest offer of the group; equal costs: the lower child (the serial scan reaches it first)
					float offer = best; int offer_child = best_slot >= 0 ? c : 8, offer_slot = best_slot;
					#pragma unroll
					for (int step = 1; step < 8; step <<= 1) {
						float other = __shfl_xor(offer, step); int other_child = __shfl_xor(offer_child, step), other_slot = __shfl_xor(offer_slot, step);
						bool take = other_child < 8 && (offer_child >= 8 || other < offer || (other == offer && other_child < offer_child));
						if (take) { offer = other; offer_child = other_child; offer_slot = other_slot; }
					}
					if (offer_child >= 8) break;   // nothing comparable left (NaN boxes): the rest takes free slots in child order
					if (c == offer_child) my_slot = offer_slot;
					taken |= 1u << offer_slot;
				}
				for (int child = 0; child < children; child++) {
					int slot_of_that_child = __shfl(my_slot, group_lane0 + child);
					if (slot_of_that_child >= 0) continue;
					int free_slot = __ffs(int(~taken & 0xffu)) - 1;
					if (c == child) my_slot = free_slot;
					taken |= 1u << free_slot;
				}

				// the node (tlas_encode_node, one child per lane)
				const bool inner = valid && run_hi - run_lo > 1;
				const unsigned inner_mask = group8_bits(inner ? 1u << my_slot : 0u), leaf_mask = group8_bits(valid && !inner ? 1u << my_slot : 0u);
				const int node_base = a.bases[2 * k], leaf_base = a.bases[2 * k + 1];
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
					header[d] = __float_as_uint(node.min[d]);
				}
				header[3] = exponents | (inner_mask << 24);
				#pragma unroll
				for (int j = 0; j < 3; j++) {
					const int w = c + 8 * j;
					if (w < 20) stage[w] = w == 0 ? header[0] : w == 1 ? header[1] : w == 2 ? header[2] : w == 3 ? header[3] : w == 4 ? uint32_t(node_base) : w == 5 ? uint32_t(leaf_base) : 0u;
				}
				if (valid) {
					volatile uint8_t * bytes = (volatile uint8_t *)stage;
					bytes[24 + my_slot] = uint8_t(inner ? (0x20 | (24 + my_slot)) : (0x20 | __popc(leaf_mask & ((1u << my_slot) - 1u))));
					#pragma unroll
					for (int d = 0; d < 3; d++) {
						float lo = floorf((box.min[d] - node.min[d]) * inv_e[d]);
						float hi = ceilf ((box.max[d] - node.min[d]) * inv_e[d]);
						lo = tlas_minf(tlas_maxf(lo, 0.0f), 255.0f); hi = tlas_minf(tlas_maxf(hi, 0.0f), 255.0f);
						bytes[32 + 16 * d + my_slot]     = uint8_t(lo);
						bytes[32 + 16 * d + 8 + my_slot] = uint8_t(hi);
					}
				}
				uint32_t * dst = a.nodes + 20 * size_t(in[3 * k]);
				#pragma unroll
				for (int j = 0; j < 3; j++) { const int w = c + 8 * j; if (w < 20) dst[w] = stage[w]; }

				// leaf order and the inner children's entries of the next level, both in slot order
				if (inner) {
					int rank = __popc(inner_mask & ((1u << my_slot) - 1u));
					int * q = out + 3 * size_t(node_base - first_node_of_next_level + rank);
					q[0] = node_base + rank; q[1] = run_lo; q[2] = run_hi;
				} else if (valid) {
					a.order[leaf_base + __popc(leaf_mask & ((1u << my_slot) - 1u))] = int(keys[run_lo] & 0xffffffffull);
				}
			}
		}
		__syncthreads();
		if (tid == 0) level_count = next_count;
		__syncthreads();
		RT_TLAS_STAMP(4)
	}

	// ---- 4: the per-instance tables in TLAS order
	if (tid == 0) *a.node_count = nodes_used;
	for (int pos = tid; pos < n; pos += threads) {
		int src = a.order[pos];
		a.position[src] = pos;
		a.out_root_indices[pos] = a.root_indices[src];
		a.out_material_ids[pos] = a.material_ids[src];
		for (int r = 0; r < 3; r++) {
			a.out_transforms     [3 * size_t(pos) + r] = a.transforms     [3 * size_t(src) + r];
			a.out_transforms_inv [3 * size_t(pos) + r] = a.transforms_inv [3 * size_t(src) + r];
			a.out_transforms_prev[3 * size_t(pos) + r] = a.transforms_prev[3 * size_t(src) + r];
		}
	}
#ifdef RT_TLAS_PHASE_TIMES
	__syncthreads();
	RT_TLAS_STAMP(5)
	if (tid == 0) printf("kernel_build_tlas n=%d threads=%d: boxes %.1f us, sort %.1f us, levels: runs %.1f + numbering %.1f + nodes %.1f us, tables %.1f us\n", n, threads,
		stamp[0] * 0.01, stamp[1] * 0.01, stamp[2] * 0.01, stamp[3] * 0.01, stamp[4] * 0.01, stamp[5] * 0.01);
#endif
}

void rt_launch_build_tlas(const TlasBuildArgs & args, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_build_tlas, dim3(1), dim3(args.count <= RT_BUILD_SMALL ? 256 : RT_BUILD_THREADS), 0, stream, args);
}
