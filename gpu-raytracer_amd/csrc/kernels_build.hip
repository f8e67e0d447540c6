// kernels_build.hip -- the per-frame TLAS build on the device.
//
// Replaces Integrator::build_tlas of the reference (Renderer/Integrators/Integrator.cpp:399-430: SAH build over the
// instance boxes, CWBVH conversion, re-ordering of the per-instance tables -- the only CPU work inside the frame loop of
// an animated scene) by ONE launch of ONE workgroup: a TLAS has hundreds to a few thousand leaves, far too few to need
// the machine, and one workgroup can keep the whole sort in LDS and separate its phases with barriers instead of
// launches. The per-node arithmetic is in rt_tlas_build.h; this file is the parallel schedule around it:
//
//   1  world box of every instance (its BLAS' object-space box through the instance matrix), scene box by reduction
//   2  64-bit keys (30-bit Morton code of the box centre << 32 | instance) sorted by a bitonic network in LDS
//   3  breadth-first, level by level: every node covers a run of the sorted order;
//        a) one thread per node cuts its run into up to 8 child runs (single instances become leaves)
//        b) thread 0 turns the per-node counts into node indices / instance positions (prefix sums: the numbering is
//           that of a serial build, bit for bit -- the CPU restatement in oracle/oracle_tlas.cpp produces the same bytes)
//        c) one thread per (node, child) unions the boxes of the child's run
//        d) one thread per node assigns octant slots, quantises and writes the 80-byte node, the order of its leaf
//           instances, and its inner children into the next level's queue
//   4  the five per-instance tables are gathered into TLAS order, and position[scene index] is written for the light
//      tables (which name instances by scene index when the TLAS is built here)
#include "rt_math.h"
#include "rt_tlas_build.h"

#define RT_BUILD_THREADS 1024

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
	TlasBox  * child_boxes;          // [count][8]
};

__global__ void __launch_bounds__(RT_BUILD_THREADS) kernel_build_tlas(TlasBuildArgs a) {
	__shared__ uint64_t keys[RT_TLAS_BUILD_MAX];
	__shared__ float reduce[RT_BUILD_THREADS / RT_WAVE_SIZE][6];
	__shared__ TlasBox scene;
	__shared__ int level_count, next_count, nodes_used, leaves_used;
	const int tid = threadIdx.x, n = a.count;
	const int threads = int(blockDim.x);   // 256 for small scenes (cheaper barriers), RT_BUILD_THREADS beyond

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
	for (int i = tid; i < n; i += threads) a.sorted_boxes[i] = a.boxes[int(keys[i] & 0xffffffffull)];
	__syncthreads();

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
			for (int c = 0; c < children; c++) inner += begin[c + 1] - begin[c] > 1;
			int * r = a.runs + 12 * size_t(k);
			for (int c = 0; c <= children; c++) r[c] = begin[c];
			r[9] = children; r[10] = inner; r[11] = children - inner;
		}
		__syncthreads();
		// b) numbering: prefix sums over the nodes of the level, in queue order
		if (tid == 0) {
			int node_base = nodes_used, leaf_base = leaves_used;
			for (int k = 0; k < count; k++) {
				a.bases[2 * k] = node_base; a.bases[2 * k + 1] = leaf_base;
				node_base += a.runs[12 * size_t(k) + 10]; leaf_base += a.runs[12 * size_t(k) + 11];
			}
			next_count = node_base - nodes_used;
			nodes_used = node_base; leaves_used = leaf_base;
		}
		// c) child boxes
		for (int t = tid; t < 8 * count; t += threads) {
			int k = t >> 3, c = t & 7;
			const int * r = a.runs + 12 * size_t(k);
			if (c >= r[9]) continue;
			TlasBox box; tlas_box_empty(box);
			for (int i = r[c]; i < r[c + 1]; i++) tlas_box_grow(box, a.sorted_boxes[i]);
			a.child_boxes[8 * size_t(k) + c] = box;
		}
		__syncthreads();
		// d) slots, node, leaf order, next level
		for (int k = tid; k < count; k += threads) {
			const int * r = a.runs + 12 * size_t(k);
			const int children = r[9];
			TlasBox node; tlas_box_empty(node);
			TlasBox boxes[8];
			for (int c = 0; c < children; c++) { boxes[c] = a.child_boxes[8 * size_t(k) + c]; tlas_box_grow(node, boxes[c]); }
			int slot_of_child[8];
			tlas_assign_slots(node, boxes, children, slot_of_child);
			TlasBox slot_boxes[8]; int is_inner[8], child_of_slot[8];
			for (int s = 0; s < 8; s++) { is_inner[s] = -1; child_of_slot[s] = -1; }
			for (int c = 0; c < children; c++) { int s = slot_of_child[c]; slot_boxes[s] = boxes[c]; is_inner[s] = r[c + 1] - r[c] > 1; child_of_slot[s] = c; }
			const int node_base = a.bases[2 * k], leaf_base = a.bases[2 * k + 1];
			uint32_t words[20];
			tlas_encode_node(node, slot_boxes, is_inner, uint32_t(node_base), uint32_t(leaf_base), words);
			uint32_t * dst = a.nodes + 20 * size_t(in[3 * k]);
			for (int w = 0; w < 20; w++) dst[w] = words[w];
			int next_inner = 0, next_leaf = 0;
			const int queue_base = node_base - (nodes_used - next_count); // position of this node's first inner child in the next level
			for (int s = 0; s < 8; s++) {
				int c = child_of_slot[s];
				if (c < 0) continue;
				if (is_inner[s]) {
					int * q = out + 3 * size_t(queue_base + next_inner);
					q[0] = node_base + next_inner; q[1] = r[c]; q[2] = r[c + 1];
					next_inner++;
				} else {
					a.order[leaf_base + next_leaf] = int(keys[r[c]] & 0xffffffffull);
					next_leaf++;
				}
			}
		}
		__syncthreads();
		if (tid == 0) level_count = next_count;
		__syncthreads();
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
}

void rt_launch_build_tlas(const TlasBuildArgs & args, hipStream_t stream) {
	hipLaunchKernelGGL(kernel_build_tlas, dim3(1), dim3(args.count <= 1024 ? 256 : RT_BUILD_THREADS), 0, stream, args);
}
