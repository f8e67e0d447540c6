// rt_math.h -- small device vector library. Every operator is a plain component-wise
// IEEE fp32 operation (the translation units are built with -ffp-contract=off); a fused
// multiply-add only happens where fmaf is written out. That makes the traversal arithmetic
// reproducible bit for bit on the host (see DESIGN.md "arithmetic contract").
#pragma once
#include "rt_types.h"

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

#define RT_DEV __device__ __forceinline__

RT_DEV f2 mk2(float x, float y) { return { x, y }; }
RT_DEV f3 mk3(float x, float y, float z) { return { x, y, z }; }
RT_DEV f3 mk3(float s) { return { s, s, s }; }
RT_DEV f3 mk3(f4 v) { return { v.x, v.y, v.z }; }
RT_DEV f3 mk3(float4 v) { return { v.x, v.y, v.z }; }
RT_DEV f4 mk4(float x, float y, float z, float w) { return { x, y, z, w }; }
RT_DEV f4 mk4(float s) { return { s, s, s, s }; }
RT_DEV f4 mk4(f3 v) { return { v.x, v.y, v.z, 0.0f }; }
RT_DEV f4 mk4(float4 v) { return { v.x, v.y, v.z, v.w }; }
RT_DEV float4 to_float4(f4 v) { return make_float4(v.x, v.y, v.z, v.w); }
RT_DEV float4 to_float4(f3 v) { return make_float4(v.x, v.y, v.z, 0.0f); }

RT_DEV f2 operator+(f2 a, f2 b) { return { a.x + b.x, a.y + b.y }; }
RT_DEV f2 operator-(f2 a, f2 b) { return { a.x - b.x, a.y - b.y }; }
RT_DEV f2 operator*(float s, f2 a) { return { s * a.x, s * a.y }; }
RT_DEV f2 operator*(f2 a, float s) { return { a.x * s, a.y * s }; }
RT_DEV float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }

RT_DEV f3 operator-(f3 a) { return { -a.x, -a.y, -a.z }; }
RT_DEV f3 operator+(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
RT_DEV f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
RT_DEV f3 operator*(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
RT_DEV f3 operator/(f3 a, f3 b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
RT_DEV f3 operator*(f3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
RT_DEV f3 operator*(float s, f3 a) { return { s * a.x, s * a.y, s * a.z }; }
RT_DEV f3 operator/(f3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
RT_DEV f3 operator/(float s, f3 a) { return { s / a.x, s / a.y, s / a.z }; }
RT_DEV f3 operator+(f3 a, float s) { return { a.x + s, a.y + s, a.z + s }; }
RT_DEV f3 operator-(f3 a, float s) { return { a.x - s, a.y - s, a.z - s }; }
RT_DEV f3 operator-(float s, f3 a) { return { s - a.x, s - a.y, s - a.z }; }
RT_DEV f3 & operator+=(f3 & a, f3 b) { a = a + b; return a; }
RT_DEV f3 & operator*=(f3 & a, f3 b) { a = a * b; return a; }
RT_DEV f3 & operator*=(f3 & a, float s) { a = a * s; return a; }
RT_DEV f3 & operator/=(f3 & a, float s) { a = a / s; return a; }

RT_DEV f4 operator+(f4 a, f4 b) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
RT_DEV f4 operator-(f4 a, f4 b) { return { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; }
RT_DEV f4 operator*(f4 a, float s) { return { a.x * s, a.y * s, a.z * s, a.w * s }; }
RT_DEV f4 operator*(float s, f4 a) { return { s * a.x, s * a.y, s * a.z, s * a.w }; }
RT_DEV f4 operator*(f4 a, f4 b) { return { a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; }
RT_DEV f4 operator/(f4 a, float s) { return { a.x / s, a.y / s, a.z / s, a.w / s }; }
RT_DEV f4 & operator+=(f4 & a, f4 b) { a = a + b; return a; }
RT_DEV f4 & operator*=(f4 & a, float s) { a = a * s; return a; }

RT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV f3 cross(f3 a, f3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
RT_DEV float length(f3 a) { return sqrtf(dot(a, a)); }
RT_DEV float length(f2 a) { return sqrtf(dot(a, a)); }
RT_DEV f3 normalize(f3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
RT_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
RT_DEV float saturate(float v) { return clampf(v, 0.0f, 1.0f); }
RT_DEV float square(float x) { return x * x; }
RT_DEV float safe_sqrt(float x) { return sqrtf(fmaxf(0.0f, x)); }

// Explicitly fused forms used by traversal (arithmetic contract)
RT_DEV float dot_fma(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RT_DEV f3 cross_fma(f3 a, f3 b) {
	return {
		__builtin_fmaf(a.y, b.z, -(a.z * b.y)),
		__builtin_fmaf(a.z, b.x, -(a.x * b.z)),
		__builtin_fmaf(a.x, b.y, -(a.y * b.x)) };
}

RT_DEV f3 load3(RtVec3SoA v, int i) { return { v.x[i], v.y[i], v.z[i] }; }
RT_DEV void store3(RtVec3SoA v, int i, f3 a) { v.x[i] = a.x; v.y[i] = a.y; v.z[i] = a.z; }

// ---- wave64 helpers -------------------------------------------------------------------------
RT_DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// One returning atomic per wave for all lanes that append to a queue: the active lanes are
// ranked with mbcnt over the exec mask and the leader adds popcount(exec).
RT_DEV int wave_aggregated_append(int * counter) {
	unsigned long long active = __ballot(1);
	unsigned rank  = __builtin_amdgcn_mbcnt_hi(unsigned(active >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(active), 0u));
	int leader = __ffsll((long long)active) - 1;
	int base = 0;
	if (rank == 0) base = atomicAdd(counter, __popcll(active));
	base = __shfl(base, leader);
	return base + int(rank);
}

// One returning atomic per WORKGROUP and queue. A queue counter is one word; the L2 retires about 88
// returning atomics per microsecond on one word (MI355X_MICROARCH.md, "dequeue"), and the sort /
// shade kernels of a 2 M ray bounce issued 65 000 wave-level appends to two words: they ran AT that
// limit (0.30 ms for a pass that moves 200 MB). Aggregating over the WAVES waves of a workgroup
// divides the atomics by WAVES. Must be called by every thread of the workgroup in uniform control
// flow; queue = -1 (nothing to append) or 0..NQ-1; returns the slot in that queue.
template<int NQ, int WAVES>
struct BlockAppendLDS { int count[WAVES][NQ]; int base[WAVES][NQ]; };

template<int NQ, int WAVES>
RT_DEV int block_aggregated_append(int queue, int * const (&counters)[NQ], BlockAppendLDS<NQ, WAVES> & lds) {
	unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned rank = 0;
	#pragma unroll
	for (int k = 0; k < NQ; k++) {
		unsigned long long mask = __ballot(queue == k);
		if (queue == k) rank = __builtin_amdgcn_mbcnt_hi(unsigned(mask >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(mask), 0u));
		if (lane == 0) lds.count[wave][k] = __popcll(mask);
	}
	__syncthreads();
	if (threadIdx.x < NQ) {
		int k = threadIdx.x, total = 0;
		#pragma unroll
		for (int w = 0; w < WAVES; w++) { int c = lds.count[w][k]; lds.base[w][k] = total; total += c; }
		int base = total > 0 ? atomicAdd(counters[k], total) : 0;
		#pragma unroll
		for (int w = 0; w < WAVES; w++) lds.base[w][k] += base;
	}
	__syncthreads();
	return queue >= 0 ? lds.base[wave][queue] + int(rank) : -1;
}

// The same single atomic per workgroup, with the appended entries of the workgroup ORDERED BY A BUCKET (0..7): the block's
// range of the queue holds its bucket-0 entries first (wave after wave, lane after lane), then bucket 1, ... Used for the
// ray queues with the direction octant as bucket: the 256 rays a shade workgroup emits come from neighbouring pixels, and
// with equal octants next to each other the lanes of a traversal wave walk the CWBVH in the same child order and touch the
// same nodes at the same time (tools/trace_sort_experiment.py: +11 % on incoherent Sponza rays for octant runs of ~8 rays
// inside a spatial cell, nothing for spatial order alone; larger buckets lose the spatial neighbourhood again). Queue
// order is unspecified in the reference (one atomicAdd per thread); per-pixel results do not depend on it.
// Must be called by every thread of the workgroup in uniform control flow; returns the slot or -1.
template<int WAVES>
struct BlockBucketLDS { int count[8][WAVES]; int base[8][WAVES]; };

RT_DEV unsigned direction_octant(f3 d) { return (d.x < 0.0f ? 4u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 1u : 0u); }

template<int WAVES>
RT_DEV int block_bucketed_append(bool active, unsigned bucket, int * counter, BlockBucketLDS<WAVES> & lds) {
	static_assert(8 * WAVES <= 64, "the scan below is one wave wide");
	unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned rank = 0;
	#pragma unroll
	for (unsigned k = 0; k < 8; k++) {
		bool mine = active && bucket == k;
		unsigned long long mask = __ballot(mine);
		if (mine) rank = __builtin_amdgcn_mbcnt_hi(unsigned(mask >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(mask), 0u));
		if (lane == 0) lds.count[k][wave] = __popcll(mask);
	}
	__syncthreads();
	if (wave == 0) {   // exclusive scan over (bucket, wave) in bucket-major order by the first 8 * WAVES lanes
		int * flat_count = &lds.count[0][0], * flat_base = &lds.base[0][0];
		int c = lane < 8 * WAVES ? flat_count[lane] : 0, inclusive = c;
		#pragma unroll
		for (int d = 1; d < 8 * WAVES; d <<= 1) { int up = __shfl_up(inclusive, d); if (int(lane) >= d) inclusive += up; }
		int total = __shfl(inclusive, 8 * WAVES - 1);
		int base = 0;
		if (lane == 0 && total > 0) base = atomicAdd(counter, total);
		base = __shfl(base, 0);
		if (lane < 8 * WAVES) flat_base[lane] = base + inclusive - c;
	}
	__syncthreads();
	return active ? lds.base[bucket][wave] + int(rank) : -1;
}

// Two bucketed appends -- to two different queues -- for the price of one: the shade kernels end a round with a shadow ray and a
// continuation ray per thread, and two calls of the function above are four workgroup barriers around two returning device-scope
// atomics one after the other. Here both counts of every (queue, bucket, wave) cell are scanned by ONE wave in one pass (the
// first 8 * WAVES lanes hold queue A's cells, the next 8 * WAVES queue B's), the two atomics leave in the same instruction from
// two lanes, and the workgroup meets twice instead of four times. Same order within each queue as block_bucketed_append.
template<int WAVES>
struct BlockBucket2LDS { int count[2][8][WAVES]; int base[2][8][WAVES]; };

template<int WAVES>
RT_DEV void block_bucketed_append2(bool active_a, unsigned bucket_a, int * counter_a, bool active_b, unsigned bucket_b, int * counter_b,
                                   BlockBucket2LDS<WAVES> & lds, int & index_a, int & index_b) {
	static_assert(16 * WAVES <= 64, "the scan below is one wave wide");
	unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned rank_a = 0, rank_b = 0;
	#pragma unroll
	for (unsigned k = 0; k < 8; k++) {
		bool mine_a = active_a && bucket_a == k, mine_b = active_b && bucket_b == k;
		unsigned long long mask_a = __ballot(mine_a), mask_b = __ballot(mine_b);
		if (mine_a) rank_a = __builtin_amdgcn_mbcnt_hi(unsigned(mask_a >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(mask_a), 0u));
		if (mine_b) rank_b = __builtin_amdgcn_mbcnt_hi(unsigned(mask_b >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(mask_b), 0u));
		if (lane == 0) { lds.count[0][k][wave] = __popcll(mask_a); lds.count[1][k][wave] = __popcll(mask_b); }
	}
	__syncthreads();
	if (wave == 0) {
		int * flat_count = &lds.count[0][0][0], * flat_base = &lds.base[0][0][0];
		int c = lane < 16 * WAVES ? flat_count[lane] : 0, inclusive = c;
		#pragma unroll
		for (int d = 1; d < 16 * WAVES; d <<= 1) { int up = __shfl_up(inclusive, d); if (int(lane) >= d) inclusive += up; }
		const int total_a = __shfl(inclusive, 8 * WAVES - 1), total_b = __shfl(inclusive, 16 * WAVES - 1) - total_a;
		const bool second = lane >= 8 * WAVES;
		int base = 0;
		if ((lane == 0 && total_a > 0) || (lane == 8 * WAVES && total_b > 0)) base = atomicAdd(second ? counter_b : counter_a, second ? total_b : total_a);
		const int base_a = __shfl(base, 0), base_b = __shfl(base, 8 * WAVES);
		if (lane < 16 * WAVES) flat_base[lane] = second ? base_b + (inclusive - c - total_a) : base_a + (inclusive - c);
	}
	__syncthreads();
	index_a = active_a ? lds.base[0][bucket_a][wave] + int(rank_a) : -1;
	index_b = active_b ? lds.base[1][bucket_b][wave] + int(rank_b) : -1;
}
