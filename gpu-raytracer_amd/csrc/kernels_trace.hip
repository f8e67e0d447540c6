// kernels_trace.hip -- BVH8 (CWBVH) closest-hit and any-hit traversal for gfx950.
//
// Replaces kernel_trace_bvh8 / kernel_trace_shadow_bvh8 of the reference
// (CUDA/Pathtracer.cu:149-196, CUDA/Raytracing/BVH8.h:29-444).  Same algorithm per ray --
// octant-ordered traversal of 80-byte compressed 8-wide nodes, TLAS -> BLAS instancing with
// un-normalised object-space rays, Moeller-Trumbore on pre-subtracted edges -- re-designed
// for CDNA4:
//   * persistent 64-lane waves claim rays in blocks of 64..256 with ONE returning atomic per
//     block and deal them to their lanes as they go idle (ballot + mbcnt prefix rank), instead
//     of one atomicAdd per lane;
//   * per round a lane does one node step and then at most one batch of RT_TRI_BATCH triangle
//     tests (no inner triangle loop: on a 64-wide wave it ran at 10 % lane occupancy);
//   * the traversal stack lives in LDS, striped [entry][lane] so that a ds_write_b64 /
//     ds_read_b64 of any mix of per-lane depths is bank-conflict free (entry stride is 512 B,
//     a multiple of the 256 B bank row); only entries beyond RT_LDS_STACK spill, to a
//     coalesced [entry][grid lane] area in HBM (no compiler scratch);
//   * the reciprocal ray direction is computed once per ray (and per BLAS entry) so the
//     node test is 6 fma + min3/max3 per child; child bytes are unpacked with
//     v_cvt_f32_ubyte0..3;
//   * the dynamic-fetch heuristic of Ylitie et al. (section 4.4) is re-expressed for 64
//     lanes: a wave goes back to fetch rays once it has lost RT_N_W lane-iterations.
// Triangle postponing (BVH8.h:200,234-240) is intentionally absent: every ray visits nodes
// and triangles in exactly the order of the sequential algorithm, which keeps hits
// bit-identical to the CPU oracle even when two triangles tie in t.
// A wave-cooperative variant (nodes fetched coalesced by 5 adjacent lanes and handed to their
// owners through LDS, the "LDS-staged node" of the design brief) was written and measured 3x
// slower than private fetches; it was removed, see DESIGN.md 4.1 and profiles/r01_trace_variants.txt.
#include "rt_math.h"

#include <cstdio>
#include <cstdlib>

#ifndef RT_LDS_STACK
#define RT_LDS_STACK   10   // stack entries per lane kept in LDS (5 KB per wave; 24 waves per CU = 120 KB of 160 KB)
#endif
#define RT_STACK_SIZE  32   // total entries per lane (reference: BVH_STACK_SIZE, Common.h:103)
#define RT_TRACE_BLOCK 256  // 4 waves per workgroup
#ifndef RT_TRACE_WAVES_PER_SIMD
// Round 1 (profiles/r01_trace_variants.txt): 5 waves/SIMD (<= 96 VGPRs, no spills) beat 6 and 8. Round 2, with the loop
// ~50 instructions shorter and its dependent loads cut (profiles/r02_traversal_loop.txt): 6 waves/SIMD (80 VGPRs, 20 dwords
// spilled, 3 scratch accesses per round) is 2.2 % faster per step than 5, 7 (72 VGPRs, 49 dwords) 2.7 % slower than 6.
#define RT_TRACE_WAVES_PER_SIMD 6
#endif
// (Per-XCD chunks of the ray range -- L2 affinity via HW_REG_XCC_ID -- were tried and measured slower
// than one shared cursor: 1.84 vs 2.01 Grays/s at 8M incoherent rays; see profiles/r01_trace_variants.txt.)
#define RT_NUM_XCD 1
#ifndef RT_TRI_BATCH
#define RT_TRI_BATCH 2           // triangles whose loads are issued together
#endif
#ifndef RT_FETCH_BLOCK_MAX
#define RT_FETCH_BLOCK_MAX 256   // most rays claimed per cursor atomic. Rounds 1-5: 128 (launches of 8-16 M rays: profiles/r01_trace_fetch_block.txt, r04 run 13). With a burst's 40-80 M-ray
                                 // launches the shared cursor is the busiest word of the chip (633 K same-word atomics in a 10 ms launch at ~88 per us): 256 is -5 % traversal on every
                                 // workload measured in round 6 (profiles/r06_ray_blocks.txt)
#endif
// (Guided self-scheduling -- a claim takes (rays left) / (D x waves), large blocks first, small ones at the queue's end -- was built and measured in round 6: worse at every D
// (0.864 ... 1.016 ms per step against 0.850): what a block size buys is not fewer atomics or a shorter tail but lanes that hold NEIGHBOURS of the queue; profiles/r06_ray_blocks.txt.)
#ifndef RT_ENDGAME
#define RT_ENDGAME 1   // the merged wavefront's launches deal the end of a queue in per-wave regions (bvh8_trace_engine: `regions`); 0: the shared cursor to the last ray
#endif
// (The WHOLE queue dealt in regions -- a wave owns rays / waves consecutive rays, no shared cursor at all -- was measured in round 6: +3.7 % on the whole frame's launches, -1 % on a
// rank's: with the shared cursor the 1.5 M rays in flight at one time are ONE stretch of the queue, which the caches like; profiles/r06_endgame.txt.)
#ifndef RT_ENDGAME_REGION_MAX
#define RT_ENDGAME_REGION_MAX 256   // rays of a queue's end that a wave owns
#endif
#ifndef RT_ENDGAME_SCAN_LIMIT
#define RT_ENDGAME_SCAN_LIMIT 2     // looks (one load per asking lane, spread over the ring of regions) a wave takes, after the region it worked on is finished, before it concludes that there is nothing left to help with
#endif
#ifndef RT_FETCH_BLOCK_DIVISOR
#define RT_FETCH_BLOCK_DIVISOR 2   // a launch's block size: rays / (this x waves of the grid), between 64 and RT_FETCH_BLOCK_MAX
#endif
// Measured and not kept (profiles/r03_variants_at_20_steps.txt, profiles/r04_shadow_rays.txt, profiles/r04_traversal_experiments.txt; the code is gone, the
// numbers are there): holding the triangle phase back until 8 / 16 lanes want it (+-0), shadow rays taking a node's children far end first (-6.8 % node
// steps, 0 % time), 96-byte decoded nodes (+5 %), the top of the tree in LDS (+1.3 %), the next node's loads issued behind the triangle tests (+-0 at
// one wave per SIMD less), v_pk_fma_f32 slab tests (+3 %), per-XCD ray cursors (-8 % rays/s).
#ifndef RT_N_D
#define RT_N_D 4            // dynamic fetch: tolerated idle lanes per iteration (N_d = 4 of 32 in the reference)
#define RT_N_W 16           // dynamic fetch: lost lane-iterations before refilling  (N_w = 16 of 32); swept 0/1 .. 24/64, profiles/r01_trace_fetch_block.txt
#endif

// Issue priority of a wave by phase of its round (s_setprio), flattened scene's wide engine only. RT_PRIO is five decimal digits, the level (0-3) of: the top of a
// round until the node's five loads are out | the node test | from there until the triangles' loads are out | the triangle tests | the end of the round (retire / pop).
// 0: no s_setprio at all. A round is two dependent memory round trips with ~220 and ~120 vector instructions behind them; a wave that is about to ISSUE loads should win
// the arbitration against waves that grind through a test, so that its latency starts running. Measured (round 5, profiles/r05_traversal_experiments.txt, traversal ms
// per step, shipped = no hint): 20202 0.958 / 0.977 / 0.959, 0.962 against 0.975 / 0.979 / 0.971, 0.966, 0.969 on three boxes (-0.2 ... -1.7 %); 10101 and 20222 the same;
// the inverse (02020: the tests first) +1.9 %. Same instructions, same floats: a scheduling hint only.
#ifndef RT_PRIO
#define RT_PRIO 20202
#endif
#ifndef RT_PRIO_EVERY_ENGINE
#define RT_PRIO_EVERY_ENGINE 1   // 1: the engine that walks a TLAS takes the hint as well (the reference's layout, one call: traversal 1.6235 -> 1.6023 ms per step); 0: the flattened scene's only
#endif
#define RT_SETPRIO(phase) do { if (RT_PRIO && (FLAT || RT_PRIO_EVERY_ENGINE) && !NARROW) __builtin_amdgcn_s_setprio((RT_PRIO / (phase)) % 10); } while (0)
enum { RT_PHASE_TOP = 10000, RT_PHASE_NODE_TEST = 1000, RT_PHASE_TRI_ADDRESS = 100, RT_PHASE_TRI_TEST = 10, RT_PHASE_END = 1 };

struct Ray3 { f3 origin, direction; };

// Measured in round 5 and not kept (profiles/r05_traversal_experiments.txt): the non-temporal hint on the triangle loads (they would leave the CU's L1 to the
// nodes) costs +37 % -- neighbouring lanes and the two triangles of a batch share lines after all --, on the ray loads / hit stores +1.6 %.

RT_DEV unsigned msb(unsigned x) { return 31u - unsigned(__clz(int(x))); }
RT_DEV unsigned extract_byte(unsigned x, unsigned i) { return (x >> (i * 8)) & 0xffu; }
RT_DEV unsigned sign_extend_s8x4(unsigned x) { return ((x >> 7) & 0x01010101u) * 0xffu; }

RT_DEV unsigned ray_get_octant_inv4(f3 d) {
	return (d.x < 0.0f ? 0u : 0x04040404u) | (d.y < 0.0f ? 0u : 0x02020202u) | (d.z < 0.0f ? 0u : 0x01010101u);
}

RT_DEV f3 reciprocal(f3 d) { return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

RT_DEV f3 transform_position(const float4 * m, f3 p) {
	float4 r0 = m[0], r1 = m[1], r2 = m[2];
	return mk3(
		__builtin_fmaf(r0.x, p.x, __builtin_fmaf(r0.y, p.y, __builtin_fmaf(r0.z, p.z, r0.w))),
		__builtin_fmaf(r1.x, p.x, __builtin_fmaf(r1.y, p.y, __builtin_fmaf(r1.z, p.z, r1.w))),
		__builtin_fmaf(r2.x, p.x, __builtin_fmaf(r2.y, p.y, __builtin_fmaf(r2.z, p.z, r2.w))));
}
RT_DEV f3 transform_direction(const float4 * m, f3 d) {
	float4 r0 = m[0], r1 = m[1], r2 = m[2];
	return mk3(
		__builtin_fmaf(r0.x, d.x, __builtin_fmaf(r0.y, d.y, r0.z * d.z)),
		__builtin_fmaf(r1.x, d.x, __builtin_fmaf(r1.y, d.y, r1.z * d.z)),
		__builtin_fmaf(r2.x, d.x, __builtin_fmaf(r2.y, d.y, r2.z * d.z)));
}

// 8 slab tests against the quantised child boxes of one node; returns the hit mask
// (bits 24..31: inner children in octant order, bits 0..23: triangles / TLAS leaves).
// ---- skip behind the hit (rt_set_skip_behind_hit; the flattened scene's walk) ------------------------------------------------------
// The reference visits a node's children by falling `slot ^ octant` and keeps the others as ONE stack entry (child base, hit mask):
// an entry carries no distance, so a closest-hit ray still walks into every stacked child after a hit in front of it has been
// found -- 12 % of the node visits on Sponza (profiles/r05_traversal_experiments.txt g, h), each a full round that enters no child.
// Here the node test also keeps the two smallest KEYS of the children it enters: key = entry distance tmin (its low byte replaced
// by the child's bit index, unique within a node; tmin >= 0, so keys order like the distances as SIGNED integers -- a -0 sorts in
// front of everything, which is only conservative). The stacked rest of a group is everything but the child visited first (the
// highest inner bit of the mask), so its bound is the smallest key that is not that child's: `second` if the nearest child IS the
// first one, else `least`. Cut down to 16 bits (sign, exponent, 7 mantissa bits: rounded towards zero, again conservative) it
// rides in bits 8..23 of the group's mask word, which the reference leaves empty; at a pop a group whose bound is not in front of
// the hit held (shadow rays: the maximum distance -- never, their children were tested against it already) is dropped unvisited.
// Every child of such a group has tmin >= bound >= hit.t: its own test `tmin < min(.., hit.t)` in the parent fails today, the
// drop merely evaluates it with the hit distance of the pop instead of the push. Leaf children (tested in the node's own round)
// take part in the minima as well: excluding them costs instructions and changes 11.505 to 11.520 node steps per ray.
// Counted on the CPU (tests/slot_eval.py, the benchmark's rays): 13.09 -> 11.52 node steps per closest-hit ray (-12 %).
RT_DEV unsigned skip_bound_bits(unsigned hit_mask, int least, int second) {
	// (least's bit index == msb(hit_mask), written on the leading-zero count: 31 - x == ~x & 31. An empty mask has no group to bound: any value will do.)
	int key = unsigned(__builtin_clz(hit_mask)) == (~unsigned(least) & 0x1fu) ? second : least;
	return (unsigned(key) >> 8) & 0x00ffff00u;
}
RT_DEV bool skip_group_is_behind(unsigned group_y, float limit) { return int((group_y << 8) & 0xffff0000u) >= __float_as_int(limit); }
#define RT_SKIP_NO_KEY 0x7fffffff

template<bool BOUND = false>
RT_DEV unsigned bvh8_node_intersect(const Ray3 & ray, f3 inv_dir, unsigned oct_inv4, float max_distance,
                                    float4 n0, float4 n1, float4 n2, float4 n3, float4 n4, unsigned * bound_bits = nullptr) {
	f3 p = mk3(n0.x, n0.y, n0.z);
	unsigned e_imask = __float_as_uint(n0.w);

	f3 adjusted_dir_inv = mk3(
		__uint_as_float(extract_byte(e_imask, 0) << 23) * inv_dir.x,
		__uint_as_float(extract_byte(e_imask, 1) << 23) * inv_dir.y,
		__uint_as_float(extract_byte(e_imask, 2) << 23) * inv_dir.z);
	f3 adjusted_origin = (p - ray.origin) * inv_dir;

	bool neg_x = ray.direction.x < 0.0f, neg_y = ray.direction.y < 0.0f, neg_z = ray.direction.z < 0.0f;

	unsigned hit_mask = 0;
	int least = RT_SKIP_NO_KEY, second = RT_SKIP_NO_KEY;
	#pragma unroll
	for (int i = 0; i < 2; i++) {
		unsigned meta4 = __float_as_uint(i == 0 ? n1.z : n1.w);

		unsigned is_inner4   = (meta4 & (meta4 << 1)) & 0x10101010u;
		unsigned inner_mask4 = sign_extend_s8x4(is_inner4 << 3);
		unsigned bit_index4  = (meta4 ^ (oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
		unsigned child_bits4 = (meta4 >> 5) & 0x07070707u;

		unsigned q_lo_x = __float_as_uint(i == 0 ? n2.x : n2.y), q_hi_x = __float_as_uint(i == 0 ? n2.z : n2.w);
		unsigned q_lo_y = __float_as_uint(i == 0 ? n3.x : n3.y), q_hi_y = __float_as_uint(i == 0 ? n3.z : n3.w);
		unsigned q_lo_z = __float_as_uint(i == 0 ? n4.x : n4.y), q_hi_z = __float_as_uint(i == 0 ? n4.z : n4.w);

		unsigned x_min = neg_x ? q_hi_x : q_lo_x, x_max = neg_x ? q_lo_x : q_hi_x;
		unsigned y_min = neg_y ? q_hi_y : q_lo_y, y_max = neg_y ? q_lo_y : q_hi_y;
		unsigned z_min = neg_z ? q_hi_z : q_lo_z, z_max = neg_z ? q_lo_z : q_hi_z;

		#pragma unroll
		for (int j = 0; j < 4; j++) {
			float tx0 = __builtin_fmaf(float(extract_byte(x_min, j)), adjusted_dir_inv.x, adjusted_origin.x);
			float ty0 = __builtin_fmaf(float(extract_byte(y_min, j)), adjusted_dir_inv.y, adjusted_origin.y);
			float tz0 = __builtin_fmaf(float(extract_byte(z_min, j)), adjusted_dir_inv.z, adjusted_origin.z);
			float tx1 = __builtin_fmaf(float(extract_byte(x_max, j)), adjusted_dir_inv.x, adjusted_origin.x);
			float ty1 = __builtin_fmaf(float(extract_byte(y_max, j)), adjusted_dir_inv.y, adjusted_origin.y);
			float tz1 = __builtin_fmaf(float(extract_byte(z_max, j)), adjusted_dir_inv.z, adjusted_origin.z);

			float tmin = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, 0.0f));
			float tmax = fminf(fminf(tx1, ty1), fminf(tz1, max_distance));

			if (tmin < tmax) {
				unsigned child_bits = extract_byte(child_bits4, j);
				unsigned bit_index  = extract_byte(bit_index4,  j);
				hit_mask |= child_bits << bit_index;
				if (BOUND) {
					int key = int((__float_as_uint(tmin) & 0xffffff00u) | bit_index);
					second = max(min(second, key), min(max(second, key), least));   // the median of the three: the second smallest so far
					least  = min(least, key);
				}
			}
		}
	}
	if (BOUND) *bound_bits = skip_bound_bits(hit_mask, least, second);
	return hit_mask;
}

// ---- the node test again, written for the two vector pipes of this chip ---------------------------------------------------------
// profiles/r04_instruction_costs.txt: on MI355X v_fma / v_mul / v_add / v_sub_f32, v_and / v_or / v_xor / v_bitop3_b32, v_add_u32,
// v_lshrrev_b32 and v_mov_b32 retire a 64-lane wave in ~2.2 cycles and OVERLAP with the other class (v_cvt_f32_ubyte, v_min / v_max
// / v_max3, v_cmp, v_cndmask, v_bfe, v_lshlrev, v_perm, v_or3, v_mul_lo ...: 4.1 cycles; a 4-cycle and a 2-cycle instruction
// interleaved cost 4.5 cycles per PAIR). The node test above spends ~150 instructions of the 4-cycle class per node -- the 48
// conversions and 32 min / max it cannot avoid, and ~70 that select, extract and assemble -- beside ~75 of the 2-cycle class that
// ride along for free. Same arithmetic, same floats, same hit mask; what changes is which instructions move the integers:
//   * plane selection by direction sign: a bitfield select with a per-ray sign mask (v_bitop3_b32) instead of 12 v_cndmask;
//   * which children are inner nodes: shifts RIGHT, and, subtract (0x07 per inner child = y - (y >> 3) with y = bits 3 and 4 of the
//     meta byte both set) instead of two left shifts and a multiply;
//   * a child's contribution (count bits << bit index): ONE v_lshlrev_b32 with both operands byte-selected (SDWA) instead of
//     v_bfe + v_lshrrev + v_lshlrev;
//   * the hit test: v_cmpx_lt_f32 narrows the execution mask to the lanes that hit, ONE v_or_b32 adds the contribution, a scalar
//     move restores the mask -- instead of v_cmp + v_cndmask + half a v_or3.
// Measured (profiles/r04_traversal_experiments.txt, items 3 and 7): with every launch carrying ONE bounce of a burst (rt_set_stream_batch) -1.4 % of the
// traversal time against the plain test at 7 waves, on one box, twice; the flattened scene's launch runs it at 6 waves (80 registers, no scratch).
// 1: on (gfx950 only: v_bitop3_b32, SDWA and v_cmpx are this chip's; any other target compiles the plain test). 0: the plain test everywhere.
#ifndef RT_FAST_NODE
#define RT_FAST_NODE 1
#endif
template<bool BOUND = false>
RT_DEV unsigned bvh8_node_intersect_fast(const Ray3 & ray, f3 inv_dir, unsigned oct_inv4, float max_distance,
                                         float4 n0, float4 n1, float4 n2, float4 n3, float4 n4, unsigned * bound_bits = nullptr) {
#if !defined(__gfx950__)
	return bvh8_node_intersect<BOUND>(ray, inv_dir, oct_inv4, max_distance, n0, n1, n2, n3, n4, bound_bits);   // (host pass and any other target)
#else
	f3 p = mk3(n0.x, n0.y, n0.z);
	unsigned e_imask = __float_as_uint(n0.w);
	f3 adjusted_dir_inv = mk3(
		__uint_as_float(extract_byte(e_imask, 0) << 23) * inv_dir.x,
		__uint_as_float(extract_byte(e_imask, 1) << 23) * inv_dir.y,
		__uint_as_float(extract_byte(e_imask, 2) << 23) * inv_dir.z);
	f3 adjusted_origin = (p - ray.origin) * inv_dir;

	// all ones where the direction component is negative (oct_inv4 carries the three comparisons: bit 2 / 1 / 0 of every byte is SET
	// for a component that is not negative). Opaque to the optimiser, which would turn the selects below back into v_cndmask.
	unsigned neg_x = ((oct_inv4 >> 2) & 1u) - 1u, neg_y = ((oct_inv4 >> 1) & 1u) - 1u, neg_z = (oct_inv4 & 1u) - 1u;
	asm("" : "+v"(neg_x)); asm("" : "+v"(neg_y)); asm("" : "+v"(neg_z));

	unsigned hit_mask = 0;
	int least = RT_SKIP_NO_KEY, second = RT_SKIP_NO_KEY;   // BOUND: see "skip behind the hit" above
	// The lanes that run this test, read by an instruction of its own directly in front of the eight places that narrow and restore the mask
	// (volatile asm statements keep their order; the code between here and the last restore is straight-line: the loops are unrolled).
	unsigned long long all_lanes;
	asm volatile("s_mov_b64 %0, exec" : "=s"(all_lanes));
	#pragma unroll
	for (int i = 0; i < 2; i++) {
		unsigned meta4 = __float_as_uint(i == 0 ? n1.z : n1.w);
		unsigned both  = (meta4 & (meta4 >> 1)) & 0x08080808u;       // bits 3 and 4 of a meta byte set: offset >= 24, an inner child
		unsigned inner7 = both - (both >> 3);                          // 0x07 in the byte of every inner child
		unsigned bit_index4  = meta4 ^ (oct_inv4 & inner7);            // (a shift uses the low 5 bits of its amount: the count bits above them need no mask)
		unsigned child_bits4 = (meta4 >> 5) & 0x07070707u;

		unsigned q_lo_x = __float_as_uint(i == 0 ? n2.x : n2.y), q_hi_x = __float_as_uint(i == 0 ? n2.z : n2.w);
		unsigned q_lo_y = __float_as_uint(i == 0 ? n3.x : n3.y), q_hi_y = __float_as_uint(i == 0 ? n3.z : n3.w);
		unsigned q_lo_z = __float_as_uint(i == 0 ? n4.x : n4.y), q_hi_z = __float_as_uint(i == 0 ? n4.z : n4.w);

		// v_bitop3_b32 with the truth table of (c ? b : a) over the index a * 4 + b * 2 + c: 0xd8 (written out: the compiler's own choice
		// for (a & ~c) | (b & c) is v_bfi_b32, an instruction of the 4-cycle class)
		unsigned x_min = __builtin_amdgcn_bitop3_b32(q_lo_x, q_hi_x, neg_x, 0xd8), x_max = __builtin_amdgcn_bitop3_b32(q_hi_x, q_lo_x, neg_x, 0xd8);
		unsigned y_min = __builtin_amdgcn_bitop3_b32(q_lo_y, q_hi_y, neg_y, 0xd8), y_max = __builtin_amdgcn_bitop3_b32(q_hi_y, q_lo_y, neg_y, 0xd8);
		unsigned z_min = __builtin_amdgcn_bitop3_b32(q_lo_z, q_hi_z, neg_z, 0xd8), z_max = __builtin_amdgcn_bitop3_b32(q_hi_z, q_lo_z, neg_z, 0xd8);

		#pragma unroll
		for (int j = 0; j < 4; j++) {
			float tx0 = __builtin_fmaf(float(extract_byte(x_min, j)), adjusted_dir_inv.x, adjusted_origin.x);
			float ty0 = __builtin_fmaf(float(extract_byte(y_min, j)), adjusted_dir_inv.y, adjusted_origin.y);
			float tz0 = __builtin_fmaf(float(extract_byte(z_min, j)), adjusted_dir_inv.z, adjusted_origin.z);
			float tx1 = __builtin_fmaf(float(extract_byte(x_max, j)), adjusted_dir_inv.x, adjusted_origin.x);
			float ty1 = __builtin_fmaf(float(extract_byte(y_max, j)), adjusted_dir_inv.y, adjusted_origin.y);
			float tz1 = __builtin_fmaf(float(extract_byte(z_max, j)), adjusted_dir_inv.z, adjusted_origin.z);

			float tmin = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, 0.0f));
			float tmax = fminf(fminf(tx1, ty1), fminf(tz1, max_distance));

			unsigned contribution;
			if (j == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(contribution) : "v"(bit_index4), "v"(child_bits4));
			if (j == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(contribution) : "v"(bit_index4), "v"(child_bits4));
			if (j == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(contribution) : "v"(bit_index4), "v"(child_bits4));
			if (j == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(contribution) : "v"(bit_index4), "v"(child_bits4));
			// (tmin < tmax) ? hit_mask |= contribution : nothing -- as a narrowed execution mask; v_cmpx_lt_f32 is false for unordered operands, as the comparison above
			if (BOUND) {
				// key = tmin with its low byte replaced by the child's meta byte (low 5 bits: the bit index) -- ONE v_perm_b32; the two running minima are
				// updated under the narrowed mask, by the lanes that enter the child
				unsigned key = __builtin_amdgcn_perm(__float_as_uint(tmin), bit_index4, 0x07060500u + unsigned(j));
				asm volatile("v_cmpx_lt_f32 %3, %4\n\tv_or_b32 %0, %0, %5\n\tv_med3_i32 %2, %1, %2, %6\n\tv_min_i32 %1, %1, %6\n\ts_mov_b64 exec, %7"
				             : "+v"(hit_mask), "+v"(least), "+v"(second) : "v"(tmin), "v"(tmax), "v"(contribution), "v"(key), "s"(all_lanes) : "vcc");
			} else
			asm volatile("v_cmpx_lt_f32 %1, %2\n\tv_or_b32 %0, %0, %3\n\ts_mov_b64 exec, %4" : "+v"(hit_mask) : "v"(tmin), "v"(tmax), "v"(contribution), "s"(all_lanes) : "vcc");
		}
	}
	if (BOUND) *bound_bits = skip_bound_bits(hit_mask, least, second);
	return hit_mask;
#endif
}

// ---- narrow mode: 8 lanes per ray ---------------------------------------------------------------
// A launch of a few thousand rays cannot fill 64-lane waves with one ray per lane, and its duration is
// the dependent chain of its longest ray: ~150 steps of ~400 VALU instructions (4 cycles each on a
// 16-wide SIMD). In narrow mode the 8 lanes of a group work on ONE ray: lane c tests child slot c of a
// node (the same fused multiply-adds as the wide loop body), the group ORs the partial hit masks with
// three DPP moves; up to 8 triangles of a leaf group are tested at once, one per lane, and the
// sequential outcome (smallest t, ties to the triangle tested first) is rebuilt with a group minimum.
// All other per-ray state is simply replicated in the 8 lanes. A step costs about half the issue slots.
RT_DEV unsigned group8_or(unsigned v) {
	v |= unsigned(__builtin_amdgcn_mov_dpp(int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
	v |= unsigned(__builtin_amdgcn_mov_dpp(int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
	v |= unsigned(__builtin_amdgcn_mov_dpp(int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror: lane i <- lane 7 - i
	return v;
}
RT_DEV int group8_min_signed(int v) {
	v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));
	v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));
	v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));
	return v;
}
RT_DEV unsigned group8_min(unsigned v) {
	v = min(v, unsigned(__builtin_amdgcn_mov_dpp(int(v), 0xB1, 0xf, 0xf, true)));
	v = min(v, unsigned(__builtin_amdgcn_mov_dpp(int(v), 0x4E, 0xf, 0xf, true)));
	v = min(v, unsigned(__builtin_amdgcn_mov_dpp(int(v), 0x141, 0xf, 0xf, true)));
	return v;
}

// The part of bvh8_node_intersect's hit mask that child slot `child` (0..7) contributes.
// key (optional): this child's key for the bound of "skip behind the hit", RT_SKIP_NO_KEY when the child is not entered.
RT_DEV unsigned bvh8_node_intersect_child(const Ray3 & ray, f3 inv_dir, unsigned oct_inv4, float max_distance,
                                          float4 n0, float4 n1, float4 n2, float4 n3, float4 n4, unsigned child, int * key = nullptr) {
	f3 p = mk3(n0.x, n0.y, n0.z);
	unsigned e_imask = __float_as_uint(n0.w);

	f3 adjusted_dir_inv = mk3(
		__uint_as_float(extract_byte(e_imask, 0) << 23) * inv_dir.x,
		__uint_as_float(extract_byte(e_imask, 1) << 23) * inv_dir.y,
		__uint_as_float(extract_byte(e_imask, 2) << 23) * inv_dir.z);
	f3 adjusted_origin = (p - ray.origin) * inv_dir;

	bool neg_x = ray.direction.x < 0.0f, neg_y = ray.direction.y < 0.0f, neg_z = ray.direction.z < 0.0f;
	bool upper = child >= 4u; // slots 4..7 live in the second word of each pair
	unsigned j = child & 3u;

	unsigned meta4 = __float_as_uint(upper ? n1.w : n1.z);
	unsigned is_inner4   = (meta4 & (meta4 << 1)) & 0x10101010u;
	unsigned inner_mask4 = sign_extend_s8x4(is_inner4 << 3);
	unsigned bit_index4  = (meta4 ^ (oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
	unsigned child_bits4 = (meta4 >> 5) & 0x07070707u;

	unsigned q_lo_x = __float_as_uint(upper ? n2.y : n2.x), q_hi_x = __float_as_uint(upper ? n2.w : n2.z);
	unsigned q_lo_y = __float_as_uint(upper ? n3.y : n3.x), q_hi_y = __float_as_uint(upper ? n3.w : n3.z);
	unsigned q_lo_z = __float_as_uint(upper ? n4.y : n4.x), q_hi_z = __float_as_uint(upper ? n4.w : n4.z);

	unsigned x_min = neg_x ? q_hi_x : q_lo_x, x_max = neg_x ? q_lo_x : q_hi_x;
	unsigned y_min = neg_y ? q_hi_y : q_lo_y, y_max = neg_y ? q_lo_y : q_hi_y;
	unsigned z_min = neg_z ? q_hi_z : q_lo_z, z_max = neg_z ? q_lo_z : q_hi_z;

	float tx0 = __builtin_fmaf(float(extract_byte(x_min, j)), adjusted_dir_inv.x, adjusted_origin.x);
	float ty0 = __builtin_fmaf(float(extract_byte(y_min, j)), adjusted_dir_inv.y, adjusted_origin.y);
	float tz0 = __builtin_fmaf(float(extract_byte(z_min, j)), adjusted_dir_inv.z, adjusted_origin.z);
	float tx1 = __builtin_fmaf(float(extract_byte(x_max, j)), adjusted_dir_inv.x, adjusted_origin.x);
	float ty1 = __builtin_fmaf(float(extract_byte(y_max, j)), adjusted_dir_inv.y, adjusted_origin.y);
	float tz1 = __builtin_fmaf(float(extract_byte(z_max, j)), adjusted_dir_inv.z, adjusted_origin.z);

	float tmin = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, 0.0f));
	float tmax = fminf(fminf(tx1, ty1), fminf(tz1, max_distance));

	if (key) *key = tmin < tmax ? int((__float_as_uint(tmin) & 0xffffff00u) | extract_byte(bit_index4, j)) : RT_SKIP_NO_KEY;
	return tmin < tmax ? extract_byte(child_bits4, j) << extract_byte(bit_index4, j) : 0u;
}

struct HitRecord { float t, u, v; int mesh_id, triangle_id; };

// The same Moeller-Trumbore arithmetic, returning t, u, v and whether (u, v) is inside and t > 0; the
// comparison against the current best / the maximum distance is left to the caller (narrow mode).
RT_DEV bool triangle_test_values(float4 part_0, float4 part_1, float part_2_x, const Ray3 & ray, float & t, float & u, float & v) {
	f3 p0 = mk3(part_0.x, part_0.y, part_0.z);
	f3 e1 = mk3(part_0.w, part_1.x, part_1.y);
	f3 e2 = mk3(part_1.z, part_1.w, part_2_x);

	f3 h = cross_fma(ray.direction, e2);
	float a = dot_fma(e1, h);
	float f = 1.0f / a;
	f3 s = ray.origin - p0;
	u = f * dot_fma(s, h);
	if (u >= 0.0f && u <= 1.0f) {
		f3 q = cross_fma(s, e1);
		v = f * dot_fma(ray.direction, q);
		if (v >= 0.0f && u + v <= 1.0f) {
			t = f * dot_fma(e2, q);
			return t > 0.0f;
		}
	}
	return false;
}

// Moeller-Trumbore on an already fetched triangle (position_0, edge_1, edge_2 in 3 float4).
template<bool SHADOW>
RT_DEV bool triangle_test_loaded(float4 part_0, float4 part_1, float4 part_2, int mesh_id, int triangle_id, const Ray3 & ray, float max_distance, HitRecord & hit) {
	f3 p0 = mk3(part_0.x, part_0.y, part_0.z);
	f3 e1 = mk3(part_0.w, part_1.x, part_1.y);
	f3 e2 = mk3(part_1.z, part_1.w, part_2.x);

	f3 h = cross_fma(ray.direction, e2);
	float a = dot_fma(e1, h);
	float f = 1.0f / a;
	f3 s = ray.origin - p0;
	float u = f * dot_fma(s, h);
	if (u >= 0.0f && u <= 1.0f) {
		f3 q = cross_fma(s, e1);
		float v = f * dot_fma(ray.direction, q);
		if (v >= 0.0f && u + v <= 1.0f) {
			float t = f * dot_fma(e2, q);
			if (SHADOW) {
				if (t > 0.0f && t < max_distance) return true;
			} else if (t > 0.0f && t < hit.t) {
				hit.t = t; hit.u = u; hit.v = v;
				hit.mesh_id = mesh_id;
				hit.triangle_id = triangle_id;
			}
		}
	}
	return false;
}

// The same test with the ray kind decided per lane (the mixed engine of the merged wavefront); with a compile-time
// constant `shadow` it folds to triangle_test_loaded<SHADOW>.
// hit.t is the far limit of either kind: the closest hit so far, or a shadow ray's maximum distance.
RT_DEV bool triangle_test_kind(bool shadow, float4 part_0, float4 part_1, float4 part_2, int mesh_id, int triangle_id, const Ray3 & ray, HitRecord & hit) {
	f3 p0 = mk3(part_0.x, part_0.y, part_0.z);
	f3 e1 = mk3(part_0.w, part_1.x, part_1.y);
	f3 e2 = mk3(part_1.z, part_1.w, part_2.x);

	f3 h = cross_fma(ray.direction, e2);
	float a = dot_fma(e1, h);
	float f = 1.0f / a;
	f3 s = ray.origin - p0;
	float u = f * dot_fma(s, h);
	if (u >= 0.0f && u <= 1.0f) {
		f3 q = cross_fma(s, e1);
		float v = f * dot_fma(ray.direction, q);
		if (v >= 0.0f && u + v <= 1.0f) {
			float t = f * dot_fma(e2, q);
			if (t > 0.0f && t < hit.t) {
				if (shadow) return true;
				hit.t = t; hit.u = u; hit.v = v;
				hit.mesh_id = mesh_id;
				hit.triangle_id = triangle_id;
			}
		}
	}
	return false;
}

template<bool SHADOW>
RT_DEV bool triangle_test(const float4 * __restrict__ triangle_positions, int mesh_id, int triangle_id, const Ray3 & ray, float max_distance, HitRecord & hit) {
	// traversal reads the positions-only copy (48 B stride): half the cache footprint of the
	// 96-byte shading triangles, whose normals/uvs the trace kernels never touch
	const float4 * tri = triangle_positions + size_t(triangle_id) * 3;
	return triangle_test_loaded<SHADOW>(tri[0], tri[1], tri[2], mesh_id, triangle_id, ray, max_distance, hit);
}

// Per-lane stack: the first RT_LDS_STACK entries live in LDS ([entry][lane] stripes, see the
// file header), deeper entries spill to a context-owned HBM buffer laid out [entry][global lane]
// so that even the spill traffic is coalesced. No private arrays => no compiler scratch.
// The two pointers carry their address space in the type: with generic pointers the compiler emits
// FLAT instructions, which go through the texture-address path like a global access (they were
// 55 % of this kernel's VMEM issue, profiles/r01_pmc_trace_secondary_4M.txt) and tie the LDS
// accesses to vmcnt; as ds_read/ds_write_b64 they cost no vector-memory issue at all.
typedef __attribute__((address_space(3))) uint2 LdsUint2;
typedef __attribute__((address_space(1))) uint2 GlobalUint2;
struct TraversalStack {
	LdsUint2    * lds;       // this lane's column of the LDS stripe
	GlobalUint2 * spill;     // this lane's column of the HBM spill area
	int spill_stride;  // lanes in the grid
	int size;

	RT_DEV void push(uint2 item) {
		if (size < RT_LDS_STACK) { lds[size * RT_WAVE_SIZE].x = item.x; lds[size * RT_WAVE_SIZE].y = item.y; }
		else { GlobalUint2 * slot = spill + size_t(size - RT_LDS_STACK) * spill_stride; slot->x = item.x; slot->y = item.y; }
		size++;
	}
	RT_DEV uint2 pop() {
		size--;
		if (size < RT_LDS_STACK) return make_uint2(lds[size * RT_WAVE_SIZE].x, lds[size * RT_WAVE_SIZE].y);
		const GlobalUint2 * slot = spill + size_t(size - RT_LDS_STACK) * spill_stride;
		return make_uint2(slot->x, slot->y);
	}
};

// Ray distribution: rays are claimed from the shared cursor in BLOCKS, one returning atomic per
// `ray_block` rays, and dealt to the lanes of the claiming wave as they go idle; the claimed-but-
// undealt range [next, end) and the drained flag are per-wave words in LDS (refills happen under
// divergence, so any subset of lanes must be able to reach them). Why blocks: a same-address
// atomic retires at ~88 per microsecond in L2 on this chip (MI355X_MICROARCH.md, "dequeue"); with
// one atomic per refill (~23 rays for incoherent rays, ~57 for primary rays) the WHOLE GPU was
// capped at 88 x 23 = 2.0 Grays/s resp. 88 x 57 = 5.0 Grays/s whatever the traversal code did.

// Workgroup memory of every engine below (a kernel runs one of them): the traversal stacks and the
// per-wave ray-claim words. File scope, so that the wide and the narrow instantiation of the CWBVH
// engine inside one kernel share ONE allocation.
__shared__ uint2 shared_stack[(RT_TRACE_BLOCK / RT_WAVE_SIZE) * RT_LDS_STACK * RT_WAVE_SIZE];
__shared__ int   shared_fetch[RT_TRACE_BLOCK / RT_WAVE_SIZE][4]; // per wave: next, end, drained, which queue
// The fused launch of the merged wavefront: BLAS root (| identity flag) of every instance of a small scene. A ray enters ~8
// instances on Sponza, and the root index was a dependent global load (TLAS leaf -> root index -> root node) with a full
// wait behind it in the middle of a round; from LDS it costs a fraction of that latency.
#define RT_ROOTS_IN_LDS 1024
__shared__ int   shared_roots[RT_ROOTS_IN_LDS];
// The common traversal engine. RaySource supplies rays and consumes results so that the same
// code serves the wavefront queues and the stand-alone entry points.
// COUNT adds per-ray work counters (nodes fetched, triangles tested, instance entries) that are
// flushed with atomics when a ray retires; it exists to MEASURE the algorithmic bytes of a launch
// (rt_set_trace_statistics) and is never used in a timed frame.
// MODE: RT_TRACE_CLOSEST / RT_TRACE_SHADOW (one queue, the kind is a compile-time constant), or RT_TRACE_MIXED -- the fused
// launch of the merged wavefront: the closest-hit queue first, then the shadow queue (ray_count_2, cursor_2), with the kind
// of a ray a PER-LANE value: a lane that finds the closest-hit queue drained takes a shadow ray at once instead of idling
// until the longest closest-hit ray of its wave is done (two engine calls one after the other did that: every wave ended
// its first phase at a few lanes' occupancy). The source of a mixed launch takes the kind as first argument.
enum { RT_TRACE_CLOSEST = 0, RT_TRACE_SHADOW = 1, RT_TRACE_MIXED = 2 };
template<int MODE, typename Source> RT_DEV void source_load(const Source & src, bool shadow, int i, Ray3 & ray, float & max_distance) {
	if constexpr (MODE == RT_TRACE_MIXED) src.load(shadow, i, ray, max_distance); else src.load(i, ray, max_distance);
}
template<int MODE, typename Source> RT_DEV void source_finish(const Source & src, bool shadow, int i, const HitRecord & hit, bool occluded) {
	if constexpr (MODE == RT_TRACE_MIXED) src.finish(shadow, i, hit, occluded); else src.finish(i, hit, occluded);
}

// UNIFIED: the TLAS nodes have been copied into the slots [0, tlas_node_count) that the BLAS node array reserves for them
// (the merged wavefront does that whenever the TLAS changes, rt_api.hip: stream_sync_tlas), so a node is fetched from ONE
// base address: no compare / select of two 64-bit bases and no scalar load of the TLAS size in every round.
// FLAT: the whole scene is one world-space tree rooted in node 0 (rt_set_static_geometry): there is no TLAS to walk, no instance
// to enter or leave, no object-space ray -- the code for those and the three registers that track them are compiled out.
// SKIP: "skip behind the hit" (above; rt_set_skip_behind_hit) -- only for scenes that are ONE tree (p.entry_tlas_stack_size == 0): a stack entry
// is then always a group of inner children of that tree.
template<int MODE, bool COUNT, bool NARROW, bool UNIFIED = false, bool FLAT = false, bool SKIP = false, typename Source>
RT_DEV void bvh8_trace_engine(const RtParams & p, Source & src, int ray_count, int * xcd_counters, unsigned long long * stats = nullptr, int ray_count_2 = 0, int * cursor_2 = nullptr, int * regions = nullptr) {
	constexpr bool SHADOW = MODE == RT_TRACE_SHADOW;   // the kind of every ray, unless MODE == RT_TRACE_MIXED: then lane_shadow
	bool lane_shadow = SHADOW;
	#define RT_IS_SHADOW (MODE == RT_TRACE_MIXED ? lane_shadow : SHADOW)
	const float4 * __restrict__ nodes     = p.bvh8_nodes;
	const float4 * __restrict__ triangles = p.triangle_positions;

	unsigned lane = threadIdx.x & (RT_WAVE_SIZE - 1);
	unsigned wave = threadIdx.x / RT_WAVE_SIZE;
	const unsigned group_child = lane & 7u;   // narrow mode: this lane's child slot / triangle rank within its 8-lane group
	const unsigned group_base  = lane & ~7u;  //              first lane of the group

	TraversalStack stack;
	stack.lds   = (LdsUint2 *)&shared_stack[wave * (RT_LDS_STACK * RT_WAVE_SIZE) + lane];
	stack.spill_stride = int(gridDim.x * blockDim.x);
	stack.spill = (GlobalUint2 *)(p.stack_spill + (blockIdx.x * blockDim.x + threadIdx.x));
	stack.size  = 0;

	// Rays per cursor atomic: up to RT_FETCH_BLOCK_MAX for big launches, 64 for small ones so that every wave gets work
	// (narrow mode: 8, one ray per 8-lane group).
	const int waves_in_grid = int(gridDim.x) * (RT_TRACE_BLOCK / RT_WAVE_SIZE);
	const int rays_total = ray_count + (MODE == RT_TRACE_MIXED ? ray_count_2 : 0);
	// The END of a queue (`regions`: the merged wavefront's launches, round 6). With blocks of 256 rays a wave's exit is quantised by whole blocks -- a rank's launch of 5 M rays is 3.3
	// blocks per wave: the waves left between 0.56 and 1.0 of the launch's duration, 28 % of the wave slots x time stood empty (tools/wave_clock_probe.py, profiles/r06_endgame.txt) --
	// and smaller blocks from the shared cursor cost what blocks are for: lanes that hold NEIGHBOURS of the queue (profiles/r06_ray_blocks.txt). So the last `region_size` x waves
	// rays of a queue are not dealt by the shared cursor: wave w owns region w, takes it in pieces of 64 rays with an atomic on the region's own word (no same-word contention,
	// consecutive claims are neighbours), and a wave that has finished its region helps with the regions behind it, 64 rays at a time, until none is left. A mixed launch deals
	// its two queues as ONE (closest-hit rays first; the kind of a ray is where its index falls).
	const bool endgame = !NARROW && regions != nullptr && waves_in_grid <= RT_ENDGAME_MAX_WAVES;
	const int ray_block = NARROW ? 8 : endgame ? RT_FETCH_BLOCK_MAX : max(RT_WAVE_SIZE, min(RT_FETCH_BLOCK_MAX, (rays_total / (RT_FETCH_BLOCK_DIVISOR * waves_in_grid)) & ~(RT_WAVE_SIZE - 1)));
	const int region_size  = max(RT_WAVE_SIZE, min(RT_ENDGAME_REGION_MAX, ((rays_total + waves_in_grid - 1) / waves_in_grid + RT_WAVE_SIZE - 1) & ~(RT_WAVE_SIZE - 1)));
	const int region_count = min(waves_in_grid, (rays_total + region_size - 1) / region_size);
	const int main_limit   = endgame ? max(0, rays_total - region_count * region_size) : 0;   // the shared cursor deals [0, main_limit) in blocks, the regions the rest
	const int region_stride = max(1, region_count / RT_WAVE_SIZE) | 1;   // a scan's candidates: every region_stride-th region behind the one just finished
	const unsigned wave_in_grid = blockIdx.x * (RT_TRACE_BLOCK / RT_WAVE_SIZE) + wave;
	// Only as many waves as there are blocks (regions) take part; the rest of the (machine-sized) persistent
	// grid leaves without touching the shared cursor.
	if (endgame ? wave_in_grid >= unsigned(region_count) : wave_in_grid * unsigned(ray_block) >= unsigned(rays_total)) return;
	// volatile: the words are written by one lane and read by OTHER lanes of the same wave with no
	// barrier in between; without it the compiler forwards a lane's own last view of them.
	typedef volatile __attribute__((address_space(3))) int LdsFetchWord; // typed: ds_read/ds_write, not FLAT
	LdsFetchWord * fetch_state = (LdsFetchWord *)&shared_fetch[wave][0];
	if (lane == 0) { fetch_state[0] = 0; fetch_state[1] = 0; fetch_state[2] = 0; fetch_state[3] = 0; }
	// Called (converged) by the lanes that need a ray; returns its index or -1 once the launch is drained.
	// Narrow mode: whole groups call it, every lane of a group gets the group's ray.
	// Mixed launches: fetch_state[3] says which queue the wave is claiming from (0: closest hit, 1: shadow); a wave moves on
	// to the shadow queue when a claim on the first comes back empty, and the lanes served in one round all get that kind.
	auto fetch_ray = [&]() -> int {
		while (endgame) {
			if (fetch_state[2]) return -1;
			unsigned long long want = __ballot(1);
			int n_want = __popcll(want);
			unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(want >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(want), 0u));
			bool elected = rank == 0;
			int next = fetch_state[0], end = fetch_state[1];
			if (next >= end) { // the same for every lane of the ballot
				int region = fetch_state[3];   // 0: this wave still takes blocks from the shared cursor; r + 1: it works on region r
				if (region == 0) {
					int base = main_limit;
					if (main_limit > 0) { if (elected) base = atomicAdd(xcd_counters, ray_block); base = __builtin_amdgcn_readfirstlane(base); }
					if (base < main_limit) { next = base; end = min(base + ray_block, main_limit); }
					else region = int(wave_in_grid) + 1;
				}
				if (region != 0) {
					next = 0; end = 0;
					int scan_from = region, scanned = 0;   // (region index + 1: the first region behind this one)
					while (true) {
						int taken = 0;
						if (elected) taken = atomicAdd(&regions[region - 1], RT_WAVE_SIZE);
						taken = __builtin_amdgcn_readfirstlane(taken);
						int first = main_limit + (region - 1) * region_size + taken;
						if (taken < region_size && first < rays_total) { next = first; end = min(first + RT_WAVE_SIZE, rays_total); break; }
						// this region is finished: look at the regions behind it, as many at a time as lanes are asking, for one that is not (a look past the L1: the words are other waves' cursors)
						// (a scan stalls the lanes of this wave that are in the middle of a ray: at most RT_ENDGAME_SCAN_LIMIT of them per refill, each a sample of the whole ring)
						bool found = false;
						while (!found && scanned < RT_ENDGAME_SCAN_LIMIT) {
							int candidate = scan_from + (int(rank) + 1) * region_stride;   // (< 2 x region_count + 64: no division here, its reciprocal would live in two registers through the whole walk)
							while (candidate >= region_count) candidate -= region_count;
							int seen = __hip_atomic_load(&regions[candidate], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							bool open = seen < region_size && main_limit + candidate * region_size + seen < rays_total;
							unsigned long long open_lanes = __ballot(open);
							if (open_lanes) {
								int lane_of_first = __ffsll((long long)open_lanes) - 1;
								region = __builtin_amdgcn_readlane(candidate, lane_of_first) + 1; found = true;
							} else { scan_from = scan_from + 1 < region_count ? scan_from + 1 : 0; scanned++; }
						}
						if (!found) break;   // nothing left that this wave could help with
					}
				}
				if (elected) fetch_state[3] = region;
			}
			int give = min(n_want, end - next);
			if (elected) {
				fetch_state[0] = next + give;
				fetch_state[1] = end;
				if (next >= end) fetch_state[2] = 1;   // drained
			}
			if (int(rank) < give) {
				int index = next + int(rank);
				if (MODE == RT_TRACE_MIXED) { lane_shadow = index >= ray_count; if (lane_shadow) index -= ray_count; }
				return index;
			}
			// the piece did not cover every lane: the rest goes round again
		}
		while (true) {
			if (fetch_state[2]) return -1;
			const bool second_queue = MODE == RT_TRACE_MIXED && fetch_state[3] != 0;
			const int queue_count = second_queue ? ray_count_2 : ray_count;
			int * const queue_cursor = second_queue ? cursor_2 : xcd_counters;
			unsigned long long want = __ballot(1);
			unsigned long long askers = NARROW ? (want & 0x0101010101010101ull) : want; // one per ray wanted
			int n_want = __popcll(askers);
			bool elected = __builtin_amdgcn_mbcnt_hi(unsigned(want >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(want), 0u)) == 0;
			unsigned rank = NARROW ? unsigned(__popcll(askers & ((1ull << group_base) - 1ull)))
			                       : __builtin_amdgcn_mbcnt_hi(unsigned(want >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(want), 0u));
			int next = fetch_state[0], end = fetch_state[1];
			if (next >= end) { // the same for every lane of the ballot: claim the next block
				int base = 0;
				if (elected) base = atomicAdd(queue_cursor, ray_block);
				base = __builtin_amdgcn_readfirstlane(base);
				next = min(base, queue_count);
				end  = min(base + ray_block, queue_count);
			}
			int give = min(n_want, end - next);
			if (elected) {
				fetch_state[0] = next + give;
				fetch_state[1] = end;
				if (next >= end) { // this queue is drained
					if (MODE == RT_TRACE_MIXED && !second_queue) { fetch_state[0] = 0; fetch_state[1] = 0; fetch_state[3] = 1; }
					else fetch_state[2] = 1;
				}
			}
			if (int(rank) < give) { if (MODE == RT_TRACE_MIXED) lane_shadow = second_queue; return next + int(rank); }
			// the block did not cover every lane: the rest goes round again (and claims a new block)
		}
	};

	uint2 current_group = make_uint2(0, 0);
	uint2 triangle_group = make_uint2(0, 0);  // leaf work of this lane that is still to be tested

	int  ray_index = 0;
	Ray3 ray;
	f3   inv_dir;
	unsigned oct_inv4 = 0;
	// (a shadow ray's maximum distance is kept in hit.t: the node test's far limit and the bound of an accepted triangle are `t < hit.t` for both kinds --
	// a closest-hit ray lowers it with every hit, a shadow ray ends at its first; one register for the two)
	HitRecord hit;
	int  tlas_stack_size = RT_INVALID;
	int  mesh_id = 0;
	bool mesh_has_identity_transform = true;
	unsigned count_nodes = 0, count_triangles = 0, count_inst_xform = 0, count_inst_ident = 0;

	// COUNT: a lane sums the work of its rays and adds the sums to the launch's statistics when it leaves (an atomic per ray and counter was 400 M same-word
	// atomics for the burst's largest launch: 3.8 s of an untimed pass of bench.py)
	unsigned long long lane_totals[2][5] = { { 0, 0, 0, 0, 0 }, { 0, 0, 0, 0, 0 } };
	auto flush_counts = [&]() {
		#pragma unroll
		for (int kind = 0; kind < (MODE == RT_TRACE_MIXED ? 2 : 1); kind++) {
			#pragma unroll
			for (int k = 0; k < 5; k++) if (lane_totals[kind][k]) atomicAdd(&stats[kind * 5 + k], lane_totals[kind][k]);
		}
	};
	// A finished ray hands over its result at the next refill, together with the other lanes that finished since the last
	// one: inside the loop the hand-over ran in almost every round for 2-3 of 64 lanes, and its stores sat in front of the
	// next round's loads (one counter for both on this chip: a wait for a load is a wait for every store before it).
	int result_pending = 0;   // 1: finished (closest hit or a shadow ray that reached its light), 2: shadow ray occluded
	while (true) {
		bool inactive = stack.size == 0 && current_group.y == 0 && triangle_group.y == 0;

		if (result_pending) {
			if (!RT_IS_SHADOW && p.has_triangle_aliases && hit.triangle_id != RT_INVALID) { // a copy in the flattened static BLAS reports its original (rt_upload_triangle_aliases)
				float4 names = triangles[size_t(hit.triangle_id) * 3 + 2];
				if (__float_as_int(names.z) >= 0) { hit.mesh_id = __float_as_int(names.z); hit.triangle_id = __float_as_int(names.w); }
			}
			if (!NARROW || group_child == 0) source_finish<MODE>(src, RT_IS_SHADOW, ray_index, hit, result_pending == 2);
			result_pending = 0;
		}
		if (inactive) {
			ray_index = fetch_ray();
			if (ray_index < 0) { if constexpr (COUNT) flush_counts(); return; }

			float max_distance;
			source_load<MODE>(src, RT_IS_SHADOW, ray_index, ray, max_distance);   // (closest-hit sources: infinity)
			inv_dir  = reciprocal(ray.direction);
			oct_inv4 = ray_get_octant_inv4(ray.direction);

			hit.t = max_distance; hit.u = 0.0f; hit.v = 0.0f; hit.mesh_id = 0; hit.triangle_id = RT_INVALID;
			// A ray starts at node 0. That is the TLAS root -- or, when the whole scene is one flattened tree (rt_set_static_geometry),
			// that tree's root, and the ray is INSIDE an instance from the start (row 0, identity: the values mesh_id and
			// mesh_has_identity_transform hold until an instance is entered, which then never happens): no step on a TLAS root and no
			// instance entry, which were one of a ray's ~13 node steps and part of a round. (One uniform value instead of a
			// constant; anything more here -- a push, a root index from the parameters -- cost 16 B of scratch in the loop.)
			current_group   = make_uint2(0, 0x80000000u);
			tlas_stack_size = FLAT ? 0 : p.entry_tlas_stack_size;
		}

		int iterations_lost = 0;
		bool running = true;   // a lane that has finished its ray idles (masked) until the wave refills
		do {
			if (running) {
			// ---- TLAS leaf: enter the next instance -- BEFORE the node phase, so that the lane takes its step on the BLAS root in
			// this same round. A ray enters ~8 instances on Sponza and most of those leaf groups come off the stack: with the
			// entry behind the node phase (round 2) such a lane spent a whole round on the entry alone -- 3.8 of a bounce ray's
			// ~30 rounds. tools/wave_sim: 234 -> 202 wave-instructions per incoherent ray, lane utilisation 0.52 -> 0.59.
			if (!FLAT && triangle_group.y != 0 && tlas_stack_size == RT_INVALID) {
				int mesh_offset = int(msb(triangle_group.y));
				triangle_group.y &= ~(1u << mesh_offset);
				mesh_id = int(triangle_group.x) + mesh_offset;

				if (triangle_group.y != 0)         stack.push(triangle_group);
				if (current_group.y & 0xff000000u) stack.push(current_group);
				tlas_stack_size = stack.size;
				triangle_group.y = 0;

				unsigned root = unsigned(UNIFIED && p.mesh_count <= RT_ROOTS_IN_LDS ? shared_roots[mesh_id] : p.mesh_bvh_root_indices[mesh_id]);
				mesh_has_identity_transform = (root >> 31) != 0;
				if (!mesh_has_identity_transform) {
					const float4 * m = p.mesh_transforms_inv + size_t(mesh_id) * 3;
					ray.origin    = transform_position (m, ray.origin);
					ray.direction = transform_direction(m, ray.direction);
					inv_dir  = reciprocal(ray.direction);
					oct_inv4 = ray_get_octant_inv4(ray.direction);
					if (COUNT) count_inst_xform++;
				} else if (COUNT) count_inst_ident++;
				current_group = make_uint2(root & 0x7fffffffu, 0x80000000u);
			}

			// ---- node phase: lanes with no triangle work pending advance their traversal by one step
			RT_SETPRIO(RT_PHASE_TOP);
			if (triangle_group.y == 0) {
				if (current_group.y & 0xff000000u) {
					// take the closest pending child of current_group (pushing the rest) and fetch its node
					unsigned hits_imask = current_group.y;
					// the nearest child first (the highest bit: octant order), for shadow rays as well: the walk stays the reference's
					unsigned child_index_offset = msb(hits_imask);
					unsigned child_index_base   = current_group.x;

					current_group.y &= ~(1u << child_index_offset);
					if (current_group.y & 0xff000000u) stack.push(current_group);

					unsigned slot_index     = (child_index_offset - 24) ^ (oct_inv4 & 0xffu);
					unsigned relative_index = __popc(hits_imask & ~(0xffffffffu << slot_index));
					unsigned child_node_index = child_index_base + relative_index;

					// FLAT (launched only while both arrays stay below 4 GiB, RtParams::geometry_below_4gib): a 32-bit byte offset from the uniform base -- `global_load ...
					// v_offset, s[base]` -- where the general form builds a 64-bit address per lane (v_mad_u64_u32 + a 64-bit move per fetch); -0.7 % of the traversal
					// time (round 5, one box, the shipped build measured before and after: 0.9997 / 0.9928 / 1.0001 ms per step)
					const float4 * node = FLAT ? (const float4 *)((const char *)nodes + __umul24(child_node_index, 80u))
					                    : (!UNIFIED && child_node_index < unsigned(p.tlas_node_count) ? p.tlas_nodes : nodes) + size_t(child_node_index) * 5;
					float4 n0 = node[0], n1 = node[1], n2 = node[2], n3 = node[3], n4 = node[4];
					RT_SETPRIO(RT_PHASE_NODE_TEST);
					if (COUNT) count_nodes++;
					unsigned hitmask, bound_bits = 0;
					if (NARROW) {
						int key;
						hitmask = group8_or(bvh8_node_intersect_child(ray, inv_dir, oct_inv4, hit.t, n0, n1, n2, n3, n4, group_child, SKIP ? &key : nullptr));
						if (SKIP) {   // the two smallest keys of the group's eight lanes (keys of entered children are distinct: the bit index is part of them)
							int least = group8_min_signed(key);
							int second = group8_min_signed(key == least ? RT_SKIP_NO_KEY : key);
							bound_bits = skip_bound_bits(hitmask, least, second);
						}
					} else if (RT_FAST_NODE && UNIFIED && FLAT) hitmask = bvh8_node_intersect_fast<SKIP>(ray, inv_dir, oct_inv4, hit.t, n0, n1, n2, n3, n4, &bound_bits);
					else hitmask = bvh8_node_intersect<SKIP>(ray, inv_dir, oct_inv4, hit.t, n0, n1, n2, n3, n4, &bound_bits);
					unsigned imask = extract_byte(__float_as_uint(n0.w), 3);

					current_group .x = __float_as_uint(n1.x);
					triangle_group.x = __float_as_uint(n1.y);
					current_group .y = (hitmask & 0xff000000u) | imask | (SKIP ? bound_bits : 0u);
					triangle_group.y = (hitmask & 0x00ffffffu);
				}
			}
			RT_SETPRIO(RT_PHASE_TRI_ADDRESS);

			// ---- triangle phase: ONE batch per round, then back to the node phase. Lanes with more
			// triangles than a batch keep them in triangle_group and take part in the next rounds while
			// the other lanes already traverse on; the earlier form (an inner loop until the lane with
			// the most triangles was done, 3.2 rounds per node round at 6.6 of 64 lanes) cost +23 % time.
			// Holding triangle rounds back until more lanes have work ("postponing" without reordering)
			// was measured too and is slower at every threshold: the kernel is bound by the length of
			// each ray's dependent chain, not by issue slots (profiles/r01_trace_loop_structure.txt).
			bool occluded = false;
			{
				bool has_triangles = triangle_group.y != 0 && (FLAT || tlas_stack_size != RT_INVALID);
				if (has_triangles) {
					if (NARROW) {
						// lane c of the group takes the c-th triangle from the top; all 8 lanes compute the rest mask
						unsigned rest = triangle_group.y; int my_bit = -1;
						#pragma unroll
						for (unsigned k = 0; k < 8; k++) {
							if (rest != 0) { int b = int(msb(rest)); if (k == group_child) my_bit = b; rest &= ~(1u << b); }
						}
						triangle_group.y = rest;
						float t = 0.0f, u = 0.0f, v = 0.0f; bool valid = false; int my_triangle = RT_INVALID;
						if (my_bit >= 0) {
							my_triangle = int(triangle_group.x) + my_bit;
							const float4 * tri = triangles + size_t(my_triangle) * 3;
							valid = triangle_test_values(tri[0], tri[1], tri[2].x, ray, t, u, v) && t < hit.t;
						}
						if (RT_IS_SHADOW) {   // (a group's lanes share one ray, so the kind is uniform within the group)
							if ((__ballot(valid) >> group_base) & 0xffull) occluded = true;
						} else {
							// the sequential loop keeps the smallest t and, among equal t, the triangle tested first
							unsigned key  = valid ? __float_as_uint(t) : 0xffffffffu; // t > 0: bit patterns order like the values
							unsigned best = group8_min(key);
							if (best != 0xffffffffu) {
								unsigned winners = unsigned((__ballot(valid && key == best) >> group_base) & 0xffull);
								int from = int(group_base) + __ffs(int(winners)) - 1;
								hit.t = __uint_as_float(best);
								hit.u = __shfl(u, from); hit.v = __shfl(v, from);
								hit.triangle_id = __shfl(my_triangle, from);
								hit.mesh_id = FLAT ? 0 : mesh_id;
							}
						}
					} else {
					// up to RT_TRI_BATCH triangles per round: all their loads are issued before the first
					// test, the tests run in the sequential order (each sees the hit.t left by the previous)
					int    tri_id[RT_TRI_BATCH];
					float4 tri_a[RT_TRI_BATCH], tri_b[RT_TRI_BATCH];
					float  tri_c[RT_TRI_BATCH];
					#pragma unroll
					for (int k = 0; k < RT_TRI_BATCH; k++) {
						tri_id[k] = RT_INVALID;
						if (triangle_group.y != 0) {
							int triangle_index = int(msb(triangle_group.y));
							triangle_group.y &= ~(1u << triangle_index);
							tri_id[k] = int(triangle_group.x) + triangle_index;
							const float4 * tri = FLAT ? (const float4 *)((const char *)triangles + unsigned(tri_id[k]) * 48u) : triangles + size_t(tri_id[k]) * 3;
							tri_a[k] = tri[0]; tri_b[k] = tri[1]; tri_c[k] = tri[2].x; // position_0, edge_1, edge_2
						}
					}
					RT_SETPRIO(RT_PHASE_TRI_TEST);
					#pragma unroll
					for (int k = 0; k < RT_TRI_BATCH; k++) {
						if (tri_id[k] != RT_INVALID && !occluded) {
							if (COUNT) count_triangles++;
							if (triangle_test_kind(RT_IS_SHADOW, tri_a[k], tri_b[k], make_float4(tri_c[k], 0.0f, 0.0f, 0.0f), FLAT ? 0 : mesh_id, tri_id[k], ray, hit)) occluded = true;
						}
					}
				}
			}

					}
			RT_SETPRIO(RT_PHASE_END);
			if (RT_IS_SHADOW && occluded) {
				result_pending = 2;
				stack.size = 0;
				current_group.y = 0;
				triangle_group.y = 0;
				running = false;
			}

			if (running && triangle_group.y == 0 && (current_group.y & 0xff000000u) == 0) {
				if (stack.size == 0) {
					result_pending = 1;
					current_group.y = 0;
					running = false;
				} else if (SKIP) {
					// Pop until a group in front of the hit turns up (or the stack is empty: the ray is done). Measured against reading the two entries on top
					// together and sitting out a round when both lie behind (round 6, profiles/r06_skip_behind_hit.txt): the loop is 1.2 % faster -- a lane that
					// sits out a round loses what the dropped visit would have cost.
					do { current_group = stack.pop(); } while (skip_group_is_behind(current_group.y, hit.t) && stack.size > 0);
					if (skip_group_is_behind(current_group.y, hit.t)) { current_group.y = 0; result_pending = 1; running = false; }
				} else {
				if (!FLAT && stack.size == tlas_stack_size) {
					tlas_stack_size = RT_INVALID;
					if (!mesh_has_identity_transform) {
						float unused;
						source_load<MODE>(src, RT_IS_SHADOW, ray_index, ray, unused); // world-space ray again (kept in memory, not in registers)
						inv_dir  = reciprocal(ray.direction);
						oct_inv4 = ray_get_octant_inv4(ray.direction);
					}
				}
				current_group = stack.pop();
				if ((current_group.y & 0xff000000u) == 0) { // a leaf group: the rest of a TLAS node's instances, entered at the top of the next round
					triangle_group = current_group;
					current_group  = make_uint2(0, 0);
				}
				}
			}
			if constexpr (COUNT) if (result_pending) {   // (the ray has just retired: running was true at the top of this round) -- into the lane's totals; they leave with the lane (flush_counts)
				const int kind = MODE == RT_TRACE_MIXED && lane_shadow ? 1 : 0;   // mixed launches: {closest x5, shadow x5}
				lane_totals[kind][0] += count_nodes;      lane_totals[kind][1] += count_triangles;
				lane_totals[kind][2] += count_inst_xform; lane_totals[kind][3] += count_inst_ident;
				lane_totals[kind][4] += 1;
				count_nodes = count_triangles = count_inst_xform = count_inst_ident = 0;
			}
			}

			iterations_lost += RT_WAVE_SIZE - __popcll(__ballot(running)) - RT_N_D;
		} while (iterations_lost < RT_N_W);
	}
	#undef RT_IS_SHADOW
}


#define RT_TRACE_LAUNCH_WAVES RT_TRACE_WAVES_PER_SIMD
#ifndef RT_MIXED_MAX_RAYS
#define RT_MIXED_MAX_RAYS (10 * 1024 * 1024)   // fused launches up to this many rays mix closest-hit and shadow rays within a wave
#endif
#ifndef RT_NARROW_MAX_RAYS
#define RT_NARROW_MAX_RAYS 16384   // incoherent launches up to this many rays run 8 lanes per ray (measured cross-over ~20 k rays)
#endif

// One kernel, two instantiations: the ray count is only known on the device.
// `coherent`: primary rays keep one ray per lane at any count (neighbouring lanes walk the same nodes).
// (closest-hit rays of a one-tree scene take the skipping walk when the context asks for it -- rt_skip_walk, rt_types.h; uniform over the launch)
template<bool SHADOW, bool COUNT, typename Source>
RT_DEV void bvh8_trace_persistent(const RtParams & p, Source & src, int ray_count, int * cursor, unsigned long long * stats = nullptr, bool coherent = false) {
	const bool narrow = !COUNT && !coherent && ray_count <= RT_NARROW_MAX_RAYS;
	if (!SHADOW && rt_skip_walk(p)) {
		if (narrow) bvh8_trace_engine<RT_TRACE_CLOSEST, false, true, false, false, true>(p, src, ray_count, cursor);
		else bvh8_trace_engine<RT_TRACE_CLOSEST, COUNT, false, false, false, true>(p, src, ray_count, cursor, stats);
		return;
	}
	if (narrow) bvh8_trace_engine<SHADOW ? RT_TRACE_SHADOW : RT_TRACE_CLOSEST, false, true>(p, src, ray_count, cursor);
	else bvh8_trace_engine<SHADOW ? RT_TRACE_SHADOW : RT_TRACE_CLOSEST, COUNT, false>(p, src, ray_count, cursor, stats);
}
#define RT_TRACE_ENGINE bvh8_trace_persistent

RT_DEV uint4 pack_hit(const HitRecord & h) { // Buffers.h:25-32
	unsigned uv = unsigned(int(h.u * 65535.0f)) | (unsigned(int(h.v * 65535.0f)) << 16);
	return make_uint4(unsigned(h.mesh_id), unsigned(h.triangle_id), __float_as_uint(h.t), uv);
}

// ---- ray sources ------------------------------------------------------------------------------

struct ClosestHitSource {
	RtVec3SoA origin, direction;
	uint4 * hits;
	RT_DEV void load(int i, Ray3 & ray, float & max_distance) const { ray.origin = load3(origin, i); ray.direction = load3(direction, i); max_distance = RT_INFINITY; }
	RT_DEV void finish(int i, const HitRecord & hit, bool) const { hits[i] = pack_hit(hit); }
};

// Wavefront shadow rays: a MISS adds the pre-computed illumination to the AOVs
// (the lambda of kernel_trace_shadow_bvh8, Pathtracer.cu:183-196).
struct ShadowQueueSource {
	RtShadowBuffer buffer;
	RtAOV radiance, direct, indirect;
	int bounce;
	RT_DEV void load(int i, Ray3 & ray, float & max_distance) const { ray.origin = load3(buffer.origin, i); ray.direction = load3(buffer.direction, i); max_distance = buffer.max_distance[i]; }
	RT_DEV void finish(int i, const HitRecord &, bool occluded) const {
		if (occluded) return;
		float4 ip = buffer.illumination_and_pixel_index[i];
		int pixel_index = __float_as_int(ip.w);
		float4 value = make_float4(ip.x, ip.y, ip.z, 0.0f);
		if (radiance.framebuffer) { float4 c = radiance.framebuffer[pixel_index]; radiance.framebuffer[pixel_index] = make_float4(c.x + value.x, c.y + value.y, c.z + value.z, c.w + value.w); }
		if (bounce == 0) {
			if (direct.framebuffer) direct.framebuffer[pixel_index] = value;
		} else if (indirect.framebuffer) {
			float4 c = indirect.framebuffer[pixel_index];
			indirect.framebuffer[pixel_index] = make_float4(c.x + value.x, c.y + value.y, c.z + value.z, c.w + value.w);
		}
	}
};

// Merged wavefront: the shadow queue holds rays emitted at different bounces; the one distinction the miss lambda makes
// (bounce 0 sets RADIANCE_DIRECT, later bounces add to RADIANCE_INDIRECT) travels as a flag in the pixel word.
struct ShadowStreamSource {
	RtShadowBuffer buffer;
	RtAOV radiance, direct, indirect;
	RT_DEV void load(int i, Ray3 & ray, float & max_distance) const { ray.origin = load3(buffer.origin, i); ray.direction = load3(buffer.direction, i); max_distance = buffer.max_distance[i]; }
	RT_DEV void finish(int i, const HitRecord &, bool occluded) const {
		if (occluded) return;
		float4 ip = buffer.illumination_and_pixel_index[i];
		unsigned pixel_word = __float_as_uint(ip.w);
		int pixel_index = int(pixel_word & ~RT_SHADOW_FLAG_BOUNCE_0);
		float4 value = make_float4(ip.x, ip.y, ip.z, 0.0f);
		if (radiance.framebuffer) { float4 c = radiance.framebuffer[pixel_index]; radiance.framebuffer[pixel_index] = make_float4(c.x + value.x, c.y + value.y, c.z + value.z, c.w + value.w); }
		if (pixel_word & RT_SHADOW_FLAG_BOUNCE_0) {
			if (direct.framebuffer) direct.framebuffer[pixel_index] = value;
		} else if (indirect.framebuffer) {
			float4 c = indirect.framebuffer[pixel_index];
			indirect.framebuffer[pixel_index] = make_float4(c.x + value.x, c.y + value.y, c.z + value.z, c.w + value.w);
		}
	}
};

// AO integrator: an occlusion ray that escapes sets RADIANCE to 1 (CUDA/AO.cu:77-101)
struct ShadowAOSource {
	RtShadowBuffer buffer;
	RtAOV radiance;
	RT_DEV void load(int i, Ray3 & ray, float & max_distance) const { ray.origin = load3(buffer.origin, i); ray.direction = load3(buffer.direction, i); max_distance = buffer.max_distance[i]; }
	RT_DEV void finish(int i, const HitRecord &, bool occluded) const {
		if (occluded || !radiance.framebuffer) return;
		int pixel_index = __float_as_int(buffer.illumination_and_pixel_index[i].w);
		radiance.framebuffer[pixel_index] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
	}
};

struct ShadowExplicitSource {
	RtVec3SoA origin, direction;
	const float * max_dist;
	uint8_t * occluded_out;
	RT_DEV void load(int i, Ray3 & ray, float & max_distance) const { ray.origin = load3(origin, i); ray.direction = load3(direction, i); max_distance = max_dist[i]; }
	RT_DEV void finish(int i, const HitRecord &, bool occluded) const { occluded_out[i] = occluded ? 1 : 0; }
};


// =================================================================================================
// Binary BVH (BVH2.h:46-244): the reference's `bvh_type = BVH` configuration (BASELINE config #1).
// Same persistent-wave machinery (block-claimed rays, LDS stack with HBM spill, one node step or one
// triangle batch per round); a node is 32 B (AABB + left_or_first + count/axis), children are visited
// near-first by the sign of the ray direction on the split axis. Not the fast path -- it exists so that
// every BVH type the host can build can also be traced on the device, bit-exactly like the oracle.
// =================================================================================================
template<bool SHADOW, typename Source>
RT_DEV void bvh2_trace_persistent(const RtParams & p, Source & src, int ray_count, int * cursor) {
	const float4 * __restrict__ nodes     = p.bvh2_nodes;
	const float4 * __restrict__ triangles = p.triangle_positions;

	unsigned lane = threadIdx.x & (RT_WAVE_SIZE - 1);
	unsigned wave = threadIdx.x / RT_WAVE_SIZE;

	TraversalStack stack;
	stack.lds   = (LdsUint2 *)&shared_stack[wave * (RT_LDS_STACK * RT_WAVE_SIZE) + lane];
	stack.spill_stride = int(gridDim.x * blockDim.x);
	stack.spill = (GlobalUint2 *)(p.stack_spill + (blockIdx.x * blockDim.x + threadIdx.x));
	stack.size  = 0;

	const int waves_in_grid = int(gridDim.x) * (RT_TRACE_BLOCK / RT_WAVE_SIZE);
	const int ray_block = max(RT_WAVE_SIZE, min(RT_FETCH_BLOCK_MAX, (ray_count / (2 * waves_in_grid)) & ~(RT_WAVE_SIZE - 1)));
	if ((blockIdx.x * (RT_TRACE_BLOCK / RT_WAVE_SIZE) + wave) * unsigned(ray_block) >= unsigned(ray_count)) return;
	typedef volatile __attribute__((address_space(3))) int LdsFetchWord;
	LdsFetchWord * fetch_state = (LdsFetchWord *)&shared_fetch[wave][0];
	if (lane == 0) { fetch_state[0] = 0; fetch_state[1] = 0; fetch_state[2] = 0; }
	auto fetch_ray = [&]() -> int { // see bvh8_trace_persistent
		while (true) {
			if (fetch_state[2]) return -1;
			unsigned long long want = __ballot(1);
			int n_want = __popcll(want);
			unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(want >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(want), 0u));
			int next = fetch_state[0], end = fetch_state[1];
			if (next >= end) {
				int base = 0;
				if (rank == 0) base = atomicAdd(cursor, ray_block);
				base = __builtin_amdgcn_readfirstlane(base);
				next = min(base, ray_count);
				end  = min(base + ray_block, ray_count);
			}
			int give = min(n_want, end - next);
			if (rank == 0) {
				fetch_state[0] = next + give;
				fetch_state[1] = end;
				if (next >= end) fetch_state[2] = 1;
			}
			if (int(rank) < give) return next + int(rank);
		}
	};

	int  ray_index = 0;
	Ray3 ray;
	f3   inv_dir;
	float max_distance = 0.0f;
	HitRecord hit;
	int  tlas_stack_size = RT_INVALID;
	int  mesh_id = 0;
	bool mesh_has_identity_transform = true;
	int  tri_next = 0, tri_end = 0; // triangles of the current leaf still to be tested

	while (true) {
		bool inactive = stack.size == 0 && tri_next >= tri_end;
		if (inactive) {
			ray_index = fetch_ray();
			if (ray_index < 0) return;
			src.load(ray_index, ray, max_distance);
			inv_dir = reciprocal(ray.direction);
			hit.t = RT_INFINITY; hit.u = 0.0f; hit.v = 0.0f; hit.mesh_id = 0; hit.triangle_id = RT_INVALID;
			tlas_stack_size = RT_INVALID;
			stack.push(make_uint2(0, 0)); // root of the TLAS
		}

		int iterations_lost = 0;
		do {
			bool occluded = false;
			if (tri_next < tri_end) {
				#pragma unroll
				for (int k = 0; k < RT_TRI_BATCH; k++) {
					if (tri_next < tri_end && !occluded) {
						const float4 * tri = triangles + size_t(tri_next) * 3;
						if (triangle_test_loaded<SHADOW>(tri[0], tri[1], make_float4(tri[2].x, 0.0f, 0.0f, 0.0f), mesh_id, tri_next, ray, max_distance, hit)) occluded = true;
						tri_next++;
					}
				}
			} else {
				if (stack.size == tlas_stack_size) { // left the BLAS: back to the world-space ray
					tlas_stack_size = RT_INVALID;
					if (!mesh_has_identity_transform) {
						float unused;
						src.load(ray_index, ray, unused);
						inv_dir = reciprocal(ray.direction);
					}
				}
				unsigned node_index = stack.pop().x;
				const float4 * node = (node_index < unsigned(p.tlas_node_count) ? p.tlas_nodes : nodes) + size_t(node_index) * 2;
				float4 a = node[0], b = node[1];
				int      left_or_first = __float_as_int(b.z);
				unsigned count_axis    = __float_as_uint(b.w);
				unsigned count = count_axis & 0x3fffffffu, axis = count_axis >> 30;

				// AABB::intersects (BVH2.h:8-16) with t = (plane - origin) * inv_dir
				float t0x = (a.x - ray.origin.x) * inv_dir.x, t1x = (a.w - ray.origin.x) * inv_dir.x;
				float t0y = (a.y - ray.origin.y) * inv_dir.y, t1y = (b.x - ray.origin.y) * inv_dir.y;
				float t0z = (a.z - ray.origin.z) * inv_dir.z, t1z = (b.y - ray.origin.z) * inv_dir.z;
				float t_near = fmaxf(fminf(t0x, t1x), fmaxf(fminf(t0y, t1y), fmaxf(fminf(t0z, t1z), 0.0f)));
				float t_far  = fminf(fmaxf(t0x, t1x), fminf(fmaxf(t0y, t1y), fminf(fmaxf(t0z, t1z), SHADOW ? max_distance : hit.t)));

				if (t_near < t_far) {
					if (count > 0) {
						if (tlas_stack_size == RT_INVALID) { // TLAS leaf: enter the instance
							tlas_stack_size = stack.size;
							mesh_id = left_or_first;
							unsigned root = unsigned(p.mesh_bvh_root_indices[mesh_id]);
							mesh_has_identity_transform = (root >> 31) != 0;
							if (!mesh_has_identity_transform) {
								const float4 * m = p.mesh_transforms_inv + size_t(mesh_id) * 3;
								ray.origin    = transform_position (m, ray.origin);
								ray.direction = transform_direction(m, ray.direction);
								inv_dir = reciprocal(ray.direction);
							}
							stack.push(make_uint2(root & 0x7fffffffu, 0));
						} else {
							tri_next = left_or_first; tri_end = left_or_first + int(count);
						}
					} else {
						float d = axis == 0 ? ray.direction.x : (axis == 1 ? ray.direction.y : ray.direction.z);
						bool left_first = d > 0.0f;
						unsigned first  = unsigned(left_first ? left_or_first     : left_or_first + 1);
						unsigned second = unsigned(left_first ? left_or_first + 1 : left_or_first);
						stack.push(make_uint2(second, 0));
						stack.push(make_uint2(first, 0));
					}
				}
			}

			if (SHADOW && occluded) {
				src.finish(ray_index, hit, true);
				stack.size = 0; tri_next = tri_end = 0;
				break;
			}
			if (stack.size == 0 && tri_next >= tri_end) {
				src.finish(ray_index, hit, false);
				break;
			}
			iterations_lost += RT_WAVE_SIZE - __popcll(__ballot(1)) - RT_N_D;
		} while (iterations_lost < RT_N_W);
	}
}


// =================================================================================================
// 4-wide BVH (BVH4.h:4-295, `bvh_type = BVH4`). 128-B nodes: the boxes of up to four children in SoA
// form plus (index, count) per child. The reference's stack holds (node, child id) and re-reads the
// child's (index, count) when it pops; here the stack entry IS (index, count) -- the same values, one
// dependent load less per step. Children are pushed far-to-near (near distances tagged with the child
// id in two mantissa bits and sorted, as in bvh4_node_intersect), so the nearest is popped first.
// =================================================================================================
template<bool SHADOW, typename Source>
RT_DEV void bvh4_trace_persistent(const RtParams & p, Source & src, int ray_count, int * cursor) {
	const float4 * __restrict__ nodes     = p.bvh4_nodes;
	const float4 * __restrict__ triangles = p.triangle_positions;

	unsigned lane = threadIdx.x & (RT_WAVE_SIZE - 1);
	unsigned wave = threadIdx.x / RT_WAVE_SIZE;

	TraversalStack stack;
	stack.lds   = (LdsUint2 *)&shared_stack[wave * (RT_LDS_STACK * RT_WAVE_SIZE) + lane];
	stack.spill_stride = int(gridDim.x * blockDim.x);
	stack.spill = (GlobalUint2 *)(p.stack_spill + (blockIdx.x * blockDim.x + threadIdx.x));
	stack.size  = 0;

	const int waves_in_grid = int(gridDim.x) * (RT_TRACE_BLOCK / RT_WAVE_SIZE);
	const int ray_block = max(RT_WAVE_SIZE, min(RT_FETCH_BLOCK_MAX, (ray_count / (2 * waves_in_grid)) & ~(RT_WAVE_SIZE - 1)));
	if ((blockIdx.x * (RT_TRACE_BLOCK / RT_WAVE_SIZE) + wave) * unsigned(ray_block) >= unsigned(ray_count)) return;
	typedef volatile __attribute__((address_space(3))) int LdsFetchWord;
	LdsFetchWord * fetch_state = (LdsFetchWord *)&shared_fetch[wave][0];
	if (lane == 0) { fetch_state[0] = 0; fetch_state[1] = 0; fetch_state[2] = 0; }
	auto fetch_ray = [&]() -> int { // see bvh8_trace_persistent
		while (true) {
			if (fetch_state[2]) return -1;
			unsigned long long want = __ballot(1);
			int n_want = __popcll(want);
			unsigned rank = __builtin_amdgcn_mbcnt_hi(unsigned(want >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(want), 0u));
			int next = fetch_state[0], end = fetch_state[1];
			if (next >= end) {
				int base = 0;
				if (rank == 0) base = atomicAdd(cursor, ray_block);
				base = __builtin_amdgcn_readfirstlane(base);
				next = min(base, ray_count);
				end  = min(base + ray_block, ray_count);
			}
			int give = min(n_want, end - next);
			if (rank == 0) {
				fetch_state[0] = next + give;
				fetch_state[1] = end;
				if (next >= end) fetch_state[2] = 1;
			}
			if (int(rank) < give) return next + int(rank);
		}
	};

	int  ray_index = 0;
	Ray3 ray;
	f3   inv_dir;
	float max_distance = 0.0f;
	HitRecord hit;
	int  tlas_stack_size = RT_INVALID;
	int  mesh_id = 0;
	bool mesh_has_identity_transform = true;
	int  tri_next = 0, tri_end = 0;

	while (true) {
		bool inactive = stack.size == 0 && tri_next >= tri_end;
		if (inactive) {
			ray_index = fetch_ray();
			if (ray_index < 0) return;
			src.load(ray_index, ray, max_distance);
			inv_dir = reciprocal(ray.direction);
			hit.t = RT_INFINITY; hit.u = 0.0f; hit.v = 0.0f; hit.mesh_id = 0; hit.triangle_id = RT_INVALID;
			tlas_stack_size = RT_INVALID;
			stack.push(make_uint2(0, 0)); // child 0 of the entry node (node 1): the TLAS root, an inner node
		}

		int iterations_lost = 0;
		do {
			bool occluded = false;
			if (tri_next < tri_end) {
				#pragma unroll
				for (int k = 0; k < RT_TRI_BATCH; k++) {
					if (tri_next < tri_end && !occluded) {
						const float4 * tri = triangles + size_t(tri_next) * 3;
						if (triangle_test_loaded<SHADOW>(tri[0], tri[1], make_float4(tri[2].x, 0.0f, 0.0f, 0.0f), mesh_id, tri_next, ray, max_distance, hit)) occluded = true;
						tri_next++;
					}
				}
			} else {
				if (stack.size == tlas_stack_size) {
					tlas_stack_size = RT_INVALID;
					if (!mesh_has_identity_transform) {
						float unused;
						src.load(ray_index, ray, unused);
						inv_dir = reciprocal(ray.direction);
					}
				}
				uint2 entry = stack.pop();
				int index = int(entry.x), count = int(entry.y);

				if (count > 0) {
					if (tlas_stack_size == RT_INVALID) { // TLAS leaf: enter the instance through its entry node
						tlas_stack_size = stack.size;
						mesh_id = index;
						unsigned root = unsigned(p.mesh_bvh_root_indices[mesh_id]);
						mesh_has_identity_transform = (root >> 31) != 0;
						if (!mesh_has_identity_transform) {
							const float4 * m = p.mesh_transforms_inv + size_t(mesh_id) * 3;
							ray.origin    = transform_position (m, ray.origin);
							ray.direction = transform_direction(m, ray.direction);
							inv_dir = reciprocal(ray.direction);
						}
						stack.push(make_uint2(root & 0x7fffffffu, 0)); // = child 0 of node root + 1
					} else {
						tri_next = index; tri_end = index + count;
					}
				} else {
					const float4 * node = (index < p.tlas_node_count ? p.tlas_nodes : nodes) + size_t(index) * 8;
					float4 min_x = node[0], min_y = node[1], min_z = node[2], max_x = node[3], max_y = node[4], max_z = node[5], ic01 = node[6], ic23 = node[7];
					float limit = SHADOW ? max_distance : hit.t;
					float t_near[4]; unsigned hit_mask = 0;
					#define RT_BVH4_CHILD(i, c) { \
						float t0x = (min_x.c - ray.origin.x) * inv_dir.x, t1x = (max_x.c - ray.origin.x) * inv_dir.x; \
						float t0y = (min_y.c - ray.origin.y) * inv_dir.y, t1y = (max_y.c - ray.origin.y) * inv_dir.y; \
						float t0z = (min_z.c - ray.origin.z) * inv_dir.z, t1z = (max_z.c - ray.origin.z) * inv_dir.z; \
						float tn = fmaxf(fminf(t0x, t1x), fmaxf(fminf(t0y, t1y), fmaxf(fminf(t0z, t1z), 0.0f))); \
						float tf = fminf(fmaxf(t0x, t1x), fminf(fmaxf(t0y, t1y), fminf(fmaxf(t0z, t1z), limit))); \
						if (tn < tf) hit_mask |= 1u << i; \
						t_near[i] = __uint_as_float((__float_as_uint(tn) & 0xfffffffcu) | unsigned(i)); }
					RT_BVH4_CHILD(0, x) RT_BVH4_CHILD(1, y) RT_BVH4_CHILD(2, z) RT_BVH4_CHILD(3, w)
					#undef RT_BVH4_CHILD
					#pragma unroll
					for (int i = 1; i < 4; i++) {
						#pragma unroll
						for (int j = i - 1; j >= 0; j--) if (t_near[j] < t_near[j + 1]) { float t = t_near[j]; t_near[j] = t_near[j + 1]; t_near[j + 1] = t; }
					}
					#pragma unroll
					for (int i = 0; i < 4; i++) {
						unsigned id = __float_as_uint(t_near[i]) & 3u;
						if ((hit_mask >> id) & 1u) {
							float cx = id == 0 ? ic01.x : (id == 1 ? ic01.z : (id == 2 ? ic23.x : ic23.z));
							float cy = id == 0 ? ic01.y : (id == 1 ? ic01.w : (id == 2 ? ic23.y : ic23.w));
							stack.push(make_uint2(__float_as_uint(cx), __float_as_uint(cy)));
						}
					}
				}
			}

			if (SHADOW && occluded) {
				src.finish(ray_index, hit, true);
				stack.size = 0; tri_next = tri_end = 0;
				break;
			}
			if (stack.size == 0 && tri_next >= tri_end) {
				src.finish(ray_index, hit, false);
				break;
			}
			iterations_lost += RT_WAVE_SIZE - __popcll(__ballot(1)) - RT_N_D;
		} while (iterations_lost < RT_N_W);
	}
}

// ---- kernels ---------------------------------------------------------------------------------------

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_bvh8(RtParams p, int bounce) {
	ClosestHitSource src { p.trace[bounce & 1].origin, p.trace[bounce & 1].direction, p.trace[bounce & 1].hits };
	RT_TRACE_ENGINE<false, false>(p, src, p.sizes->trace[bounce], p.xcd_counters + (2 * bounce) * RT_NUM_XCD, nullptr, bounce == 0);
}

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh8(RtParams p, int bounce) {
	ShadowQueueSource src { p.shadow, p.aovs[RT_AOV_RADIANCE], p.aovs[RT_AOV_RADIANCE_DIRECT], p.aovs[RT_AOV_RADIANCE_INDIRECT], bounce };
	RT_TRACE_ENGINE<true, false>(p, src, p.sizes->shadow[bounce], p.xcd_counters + (2 * bounce + 1) * RT_NUM_XCD);
}

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_bvh2(RtParams p, int bounce) {
	ClosestHitSource src { p.trace[bounce & 1].origin, p.trace[bounce & 1].direction, p.trace[bounce & 1].hits };
	bvh2_trace_persistent<false>(p, src, p.sizes->trace[bounce], p.xcd_counters + (2 * bounce) * RT_NUM_XCD);
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh2(RtParams p, int bounce) {
	ShadowQueueSource src { p.shadow, p.aovs[RT_AOV_RADIANCE], p.aovs[RT_AOV_RADIANCE_DIRECT], p.aovs[RT_AOV_RADIANCE_INDIRECT], bounce };
	bvh2_trace_persistent<true>(p, src, p.sizes->shadow[bounce], p.xcd_counters + (2 * bounce + 1) * RT_NUM_XCD);
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh2_ao(RtParams p) {
	ShadowAOSource src { p.shadow, p.aovs[RT_AOV_RADIANCE] };
	bvh2_trace_persistent<true>(p, src, p.sizes->shadow[0], p.xcd_counters + RT_NUM_XCD);
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_bvh2_explicit(RtParams p, RtVec3SoA origin, RtVec3SoA direction, uint4 * hits, int ray_count, int * retired) {
	ClosestHitSource src { origin, direction, hits };
	bvh2_trace_persistent<false>(p, src, ray_count, retired);
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh2_explicit(RtParams p, RtVec3SoA origin, RtVec3SoA direction, const float * max_distance, uint8_t * occluded, int ray_count, int * retired) {
	ShadowExplicitSource src { origin, direction, max_distance, occluded };
	bvh2_trace_persistent<true>(p, src, ray_count, retired);
}

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_bvh4(RtParams p, int bounce) {
	ClosestHitSource src { p.trace[bounce & 1].origin, p.trace[bounce & 1].direction, p.trace[bounce & 1].hits };
	bvh4_trace_persistent<false>(p, src, p.sizes->trace[bounce], p.xcd_counters + (2 * bounce) * RT_NUM_XCD);
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh4(RtParams p, int bounce) {
	ShadowQueueSource src { p.shadow, p.aovs[RT_AOV_RADIANCE], p.aovs[RT_AOV_RADIANCE_DIRECT], p.aovs[RT_AOV_RADIANCE_INDIRECT], bounce };
	bvh4_trace_persistent<true>(p, src, p.sizes->shadow[bounce], p.xcd_counters + (2 * bounce + 1) * RT_NUM_XCD);
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh4_ao(RtParams p) {
	ShadowAOSource src { p.shadow, p.aovs[RT_AOV_RADIANCE] };
	bvh4_trace_persistent<true>(p, src, p.sizes->shadow[0], p.xcd_counters + RT_NUM_XCD);
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_bvh4_explicit(RtParams p, RtVec3SoA origin, RtVec3SoA direction, uint4 * hits, int ray_count, int * retired) {
	ClosestHitSource src { origin, direction, hits };
	bvh4_trace_persistent<false>(p, src, ray_count, retired);
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh4_explicit(RtParams p, RtVec3SoA origin, RtVec3SoA direction, const float * max_distance, uint8_t * occluded, int ray_count, int * retired) {
	ShadowExplicitSource src { origin, direction, max_distance, occluded };
	bvh4_trace_persistent<true>(p, src, ray_count, retired);
}

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh8_ao(RtParams p) {
	ShadowAOSource src { p.shadow, p.aovs[RT_AOV_RADIANCE] };
	RT_TRACE_ENGINE<true, false>(p, src, p.sizes->shadow[0], p.xcd_counters + RT_NUM_XCD);
}

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_bvh8_counting(RtParams p, int bounce, unsigned long long * stats) {
	ClosestHitSource src { p.trace[bounce & 1].origin, p.trace[bounce & 1].direction, p.trace[bounce & 1].hits };
	RT_TRACE_ENGINE<false, true>(p, src, p.sizes->trace[bounce], p.xcd_counters + (2 * bounce) * RT_NUM_XCD, stats);
}

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh8_counting(RtParams p, int bounce, unsigned long long * stats) {
	ShadowQueueSource src { p.shadow, p.aovs[RT_AOV_RADIANCE], p.aovs[RT_AOV_RADIANCE_DIRECT], p.aovs[RT_AOV_RADIANCE_INDIRECT], bounce };
	RT_TRACE_ENGINE<true, true>(p, src, p.sizes->shadow[bounce], p.xcd_counters + (2 * bounce + 1) * RT_NUM_XCD, stats + 5);
}

// The ONE traversal launch of an iteration of the merged wavefront: the closest-hit rays of the iteration (primary rays of
// the newest submission and the continuation rays of all others) and then, by the same persistent waves as they run out of
// those, the shadow rays the previous iteration's shade kernels emitted. Both queues are complete when the launch starts,
// both results are needed by the same next kernel (sort), so there is nothing to gain from two launches that would only
// compete for the wave slots -- and a wave that finds the closest-hit queue drained goes straight on to shadow rays instead
// of idling through the other waves' tails.
struct MixedStreamSource {
	ClosestHitSource closest; ShadowStreamSource shadow;
	RT_DEV void load(bool is_shadow, int i, Ray3 & ray, float & max_distance) const { if (is_shadow) shadow.load(i, ray, max_distance); else closest.load(i, ray, max_distance); }
	RT_DEV void finish(bool is_shadow, int i, const HitRecord & hit, bool occluded) const { if (is_shadow) shadow.finish(i, hit, occluded); else closest.finish(i, hit, occluded); }
};
#ifdef RT_WAVE_CLOCK
// Probe build (tools/build_trace_variant.sh waveclock "-DRT_WAVE_CLOCK=1"; tools/wave_clock_probe.py): every wave of the merged wavefront's traversal launch leaves the
// constant-rate clock (100 MHz) at which it started, left the closest-hit engine and left the launch -- the drain of a persistent launch, wave by wave.
#define RT_WAVE_CLOCK_SLOTS 32
#define RT_WAVE_CLOCK_WAVES 16384
__device__ unsigned long long grt_wave_clock[RT_WAVE_CLOCK_SLOTS][RT_WAVE_CLOCK_WAVES][3];
__device__ int grt_wave_clock_meta[RT_WAVE_CLOCK_SLOTS][4];   // iteration, closest-hit rays, shadow rays, waves of the grid
extern "C" int rt_debug_read_wave_clock(void * clocks, void * meta) {
	if (hipMemcpyFromSymbol(clocks, HIP_SYMBOL(grt_wave_clock), sizeof(grt_wave_clock)) != hipSuccess) return 1;
	return hipMemcpyFromSymbol(meta, HIP_SYMBOL(grt_wave_clock_meta), sizeof(grt_wave_clock_meta)) != hipSuccess;
}
template<bool COUNT, bool FLAT = false, bool SKIP = false> RT_DEV void trace_stream_probed(const RtParams & p, unsigned long long * stats, unsigned long long * mid);
template<bool COUNT, bool FLAT = false, bool SKIP = false>
RT_DEV void trace_stream(const RtParams & p, unsigned long long * stats) {
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	unsigned long long t1 = 0;
	trace_stream_probed<COUNT, FLAT, SKIP>(p, stats, &t1);
	const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
	if ((threadIdx.x & 63) == 0) {
		const int slot = p.stream_iteration % RT_WAVE_CLOCK_SLOTS, wave = blockIdx.x * (RT_TRACE_BLOCK / RT_WAVE_SIZE) + threadIdx.x / RT_WAVE_SIZE;
		if (wave < RT_WAVE_CLOCK_WAVES) { grt_wave_clock[slot][wave][0] = t0; grt_wave_clock[slot][wave][1] = t1 ? t1 : t2; grt_wave_clock[slot][wave][2] = t2; }
		if (wave == 0) { const int q = p.stream_iteration & 1; grt_wave_clock_meta[slot][0] = p.stream_iteration; grt_wave_clock_meta[slot][1] = p.stream->trace_count[q]; grt_wave_clock_meta[slot][2] = p.stream->shadow_count[q ^ 1]; grt_wave_clock_meta[slot][3] = gridDim.x * (RT_TRACE_BLOCK / RT_WAVE_SIZE); }
	}
}
#define trace_stream_body trace_stream_probed
#define RT_WAVE_CLOCK_MID , unsigned long long * mid
#define RT_WAVE_CLOCK_MARK *mid = __builtin_amdgcn_s_memrealtime();
#else
template<bool COUNT, bool FLAT = false, bool SKIP = false> RT_DEV void trace_stream(const RtParams & p, unsigned long long * stats);
#define trace_stream_body trace_stream
#define RT_WAVE_CLOCK_MID
#define RT_WAVE_CLOCK_MARK
#endif
template<bool COUNT, bool FLAT, bool SKIP>
RT_DEV void trace_stream_body(const RtParams & p, unsigned long long * stats RT_WAVE_CLOCK_MID) {
	const int q = p.stream_iteration & 1;
	MixedStreamSource src { { p.trace[q].origin, p.trace[q].direction, p.trace[q].hits },
	                        { p.shadow, p.aovs[RT_AOV_RADIANCE], p.aovs[RT_AOV_RADIANCE_DIRECT], p.aovs[RT_AOV_RADIANCE_INDIRECT] } };
	const int closest_count = p.stream->trace_count[q], shadow_count = p.stream->shadow_count[q ^ 1];
	if (!FLAT && p.mesh_count <= RT_ROOTS_IN_LDS) {   // (uniform over the launch; before any wave leaves the kernel)
		for (int i = threadIdx.x; i < p.mesh_count; i += RT_TRACE_BLOCK) shared_roots[i] = p.mesh_bvh_root_indices[i];
		__syncthreads();
	}
	// Which engine (the counts are only known on the device; the choice is uniform over the launch):
	//   * few rays (the fill and drain iterations of the wavefront): 8 lanes per ray, see the narrow mode;
	//   * up to RT_MIXED_MAX_RAYS: mixed kinds -- a lane that finds the closest-hit queue drained takes a shadow ray at once.
	//     On an eighth of a 1080p frame (3.3 M rays per launch) that is +11 % (0.60 -> 0.67 of the roofline): every wave
	//     used to end its closest-hit phase at a few lanes' occupancy;
	//   * beyond: the two kinds one after the other. Mixing costs ~4 % of the issue slots (the kind is a per-lane value
	//     where it was a compile-time constant) and a 25 M-ray launch has little to gain from it (0.87 -> 0.83 when mixed).
	//   profiles/r02_mixed_engine.txt
	if (!COUNT && closest_count + shadow_count <= RT_NARROW_MAX_RAYS)
		bvh8_trace_engine<RT_TRACE_MIXED, false, true, true, FLAT, SKIP>(p, src, closest_count, &p.stream->cursor[q][0], nullptr, shadow_count, &p.stream->cursor[q][1]);
	else if (COUNT || closest_count + shadow_count <= RT_MIXED_MAX_RAYS)
		bvh8_trace_engine<RT_TRACE_MIXED, COUNT, false, true, FLAT, SKIP>(p, src, closest_count, &p.stream->cursor[q][0], stats, shadow_count, &p.stream->cursor[q][1], RT_ENDGAME ? &p.stream->endgame[q][0][0] : nullptr);
	else {
		bvh8_trace_engine<RT_TRACE_CLOSEST, false, false, true, FLAT, SKIP>(p, src.closest, closest_count, &p.stream->cursor[q][0], nullptr, 0, nullptr, RT_ENDGAME ? &p.stream->endgame[q][0][0] : nullptr);
		RT_WAVE_CLOCK_MARK
		bvh8_trace_engine<RT_TRACE_SHADOW,  false, false, true, FLAT>(p, src.shadow,  shadow_count,  &p.stream->cursor[q][1], nullptr, 0, nullptr, RT_ENDGAME ? &p.stream->endgame[q][1][0] : nullptr);   // (a shadow ray's limit never moves: nothing to skip)
	}
}
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_stream_bvh8(RtParams p) { trace_stream<false>(p, nullptr); }
// The flattened scene's launch: without the TLAS / instance state the engine fits 80 registers with no scratch at 6 waves per SIMD with the two-pipe node
// test (the plain test: 72 registers at 7 waves; profiles/r03_flattened_static_geometry.txt, profiles/r04_traversal_experiments.txt item 7).
#ifndef RT_FLAT_WAVES
#define RT_FLAT_WAVES (RT_FAST_NODE ? 6 : 7)
#endif
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_FLAT_WAVES) kernel_trace_stream_bvh8_flat(RtParams p) { trace_stream<false, true>(p, nullptr); }
#ifndef RT_FLAT_SKIP_WAVES
#define RT_FLAT_SKIP_WAVES RT_FLAT_WAVES
#endif
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_FLAT_SKIP_WAVES) kernel_trace_stream_bvh8_flat_skip(RtParams p) { trace_stream<false, true, true>(p, nullptr); }
__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_stream_bvh8_counting(RtParams p, unsigned long long * stats) {
	if (rt_skip_walk(p)) trace_stream<true, false, true>(p, stats); else trace_stream<true>(p, stats);
}

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_bvh8_explicit(RtParams p, RtVec3SoA origin, RtVec3SoA direction, uint4 * hits, int ray_count, int * retired) {
	ClosestHitSource src { origin, direction, hits };
	RT_TRACE_ENGINE<false, false>(p, src, ray_count, retired);
}

__global__ void __launch_bounds__(RT_TRACE_BLOCK, RT_TRACE_LAUNCH_WAVES) kernel_trace_shadow_bvh8_explicit(RtParams p, RtVec3SoA origin, RtVec3SoA direction, const float * max_distance, uint8_t * occluded, int ray_count, int * retired) {
	ShadowExplicitSource src { origin, direction, max_distance, occluded };
	RT_TRACE_ENGINE<true, false>(p, src, ray_count, retired);
}

// Persistent grid: enough workgroups to fill every CU to the occupancy the kernel reaches,
// a multiple of 8 so that all XCDs get the same share (block b runs on XCD b % 8).
static int trace_grid_size(const void * kernel) {
	static const int cus = [] {   // (a function-local static: initialised once, also when several submitting threads arrive together -- FrameSplit)
		int device = 0, count = 0;
		(void)hipGetDevice(&device);
		(void)hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, device);
		return count > 0 ? count : 256;
	}();
	int blocks_per_cu = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, RT_TRACE_BLOCK, 0) != hipSuccess || blocks_per_cu <= 0) blocks_per_cu = 2;
	if (blocks_per_cu > 8) blocks_per_cu = 8;
	if (const char * e = getenv("GRT_TRACE_BLOCKS_PER_CU")) { int n = atoi(e); if (n >= 1 && n <= 16) blocks_per_cu = n; }   // (experiments: profiles/r06_occupancy.txt)
	return cus * blocks_per_cu;
}

void rt_launch_trace(const RtParams & p, int bounce, hipStream_t stream) {
	if (p.bvh_width == 2) {
		static int grid2 = trace_grid_size((const void *)kernel_trace_bvh2);
		hipLaunchKernelGGL(kernel_trace_bvh2, dim3(grid2), dim3(RT_TRACE_BLOCK), 0, stream, p, bounce);
		return;
	}
	if (p.bvh_width == 4) {
		static int grid4 = trace_grid_size((const void *)kernel_trace_bvh4);
		hipLaunchKernelGGL(kernel_trace_bvh4, dim3(grid4), dim3(RT_TRACE_BLOCK), 0, stream, p, bounce);
		return;
	}
	static int grid = trace_grid_size((const void *)kernel_trace_bvh8);
	hipLaunchKernelGGL(kernel_trace_bvh8, dim3(grid), dim3(RT_TRACE_BLOCK), 0, stream, p, bounce);
}
void rt_launch_trace_shadow(const RtParams & p, int bounce, hipStream_t stream) {
	if (p.bvh_width == 2) {
		static int grid2 = trace_grid_size((const void *)kernel_trace_shadow_bvh2);
		hipLaunchKernelGGL(kernel_trace_shadow_bvh2, dim3(grid2), dim3(RT_TRACE_BLOCK), 0, stream, p, bounce);
		return;
	}
	if (p.bvh_width == 4) {
		static int grid4 = trace_grid_size((const void *)kernel_trace_shadow_bvh4);
		hipLaunchKernelGGL(kernel_trace_shadow_bvh4, dim3(grid4), dim3(RT_TRACE_BLOCK), 0, stream, p, bounce);
		return;
	}
	static int grid = trace_grid_size((const void *)kernel_trace_shadow_bvh8);
	hipLaunchKernelGGL(kernel_trace_shadow_bvh8, dim3(grid), dim3(RT_TRACE_BLOCK), 0, stream, p, bounce);
}
void rt_launch_trace_shadow_ao(const RtParams & p, hipStream_t stream) {
	if (p.bvh_width == 2) {
		static int grid2 = trace_grid_size((const void *)kernel_trace_shadow_bvh2_ao);
		hipLaunchKernelGGL(kernel_trace_shadow_bvh2_ao, dim3(grid2), dim3(RT_TRACE_BLOCK), 0, stream, p);
		return;
	}
	if (p.bvh_width == 4) {
		static int grid4 = trace_grid_size((const void *)kernel_trace_shadow_bvh4_ao);
		hipLaunchKernelGGL(kernel_trace_shadow_bvh4_ao, dim3(grid4), dim3(RT_TRACE_BLOCK), 0, stream, p);
		return;
	}
	static int grid = trace_grid_size((const void *)kernel_trace_shadow_bvh8_ao);
	hipLaunchKernelGGL(kernel_trace_shadow_bvh8_ao, dim3(grid), dim3(RT_TRACE_BLOCK), 0, stream, p);
}
void rt_launch_trace_stream(const RtParams & p, unsigned long long * stats, hipStream_t stream) {
	if (stats) {
		static int grid_counting = trace_grid_size((const void *)kernel_trace_stream_bvh8_counting);
		hipLaunchKernelGGL(kernel_trace_stream_bvh8_counting, dim3(grid_counting), dim3(RT_TRACE_BLOCK), 0, stream, p, stats);
		return;
	}
	if (p.entry_tlas_stack_size == 0 && p.geometry_below_4gib) {   // the whole scene is one world-space tree: the engine without the TLAS / instance code (32-bit offsets; a larger scene walks the general engine from node 0)
		if (rt_skip_walk(p)) {
			static int grid_flat_skip = trace_grid_size((const void *)kernel_trace_stream_bvh8_flat_skip);
			hipLaunchKernelGGL(kernel_trace_stream_bvh8_flat_skip, dim3(grid_flat_skip), dim3(RT_TRACE_BLOCK), 0, stream, p);
			return;
		}
		static int grid_flat = trace_grid_size((const void *)kernel_trace_stream_bvh8_flat);
		hipLaunchKernelGGL(kernel_trace_stream_bvh8_flat, dim3(grid_flat), dim3(RT_TRACE_BLOCK), 0, stream, p);
		return;
	}
	static int grid = trace_grid_size((const void *)kernel_trace_stream_bvh8);
	hipLaunchKernelGGL(kernel_trace_stream_bvh8, dim3(grid), dim3(RT_TRACE_BLOCK), 0, stream, p);
}
void rt_launch_trace_counting(const RtParams & p, int bounce, unsigned long long * stats, hipStream_t stream) {
	static int grid = trace_grid_size((const void *)kernel_trace_bvh8_counting);
	hipLaunchKernelGGL(kernel_trace_bvh8_counting, dim3(grid), dim3(RT_TRACE_BLOCK), 0, stream, p, bounce, stats);
}
void rt_launch_trace_shadow_counting(const RtParams & p, int bounce, unsigned long long * stats, hipStream_t stream) {
	static int grid = trace_grid_size((const void *)kernel_trace_shadow_bvh8_counting);
	hipLaunchKernelGGL(kernel_trace_shadow_bvh8_counting, dim3(grid), dim3(RT_TRACE_BLOCK), 0, stream, p, bounce, stats);
}
void rt_launch_trace_explicit(const RtParams & p, RtVec3SoA origin, RtVec3SoA direction, uint4 * hits, int ray_count, int * retired_counter, hipStream_t stream) {
	if (p.bvh_width == 2) {
		static int grid2 = trace_grid_size((const void *)kernel_trace_bvh2_explicit);
		hipLaunchKernelGGL(kernel_trace_bvh2_explicit, dim3(grid2), dim3(RT_TRACE_BLOCK), 0, stream, p, origin, direction, hits, ray_count, retired_counter);
		return;
	}
	if (p.bvh_width == 4) {
		static int grid4 = trace_grid_size((const void *)kernel_trace_bvh4_explicit);
		hipLaunchKernelGGL(kernel_trace_bvh4_explicit, dim3(grid4), dim3(RT_TRACE_BLOCK), 0, stream, p, origin, direction, hits, ray_count, retired_counter);
		return;
	}
	static int grid = trace_grid_size((const void *)kernel_trace_bvh8_explicit);
	hipLaunchKernelGGL(kernel_trace_bvh8_explicit, dim3(grid), dim3(RT_TRACE_BLOCK), 0, stream, p, origin, direction, hits, ray_count, retired_counter);
}
void rt_launch_trace_shadow_explicit(const RtParams & p, RtVec3SoA origin, RtVec3SoA direction, const float * max_distance, uint8_t * occluded, int ray_count, int * retired_counter, hipStream_t stream) {
	if (p.bvh_width == 2) {
		static int grid2 = trace_grid_size((const void *)kernel_trace_shadow_bvh2_explicit);
		hipLaunchKernelGGL(kernel_trace_shadow_bvh2_explicit, dim3(grid2), dim3(RT_TRACE_BLOCK), 0, stream, p, origin, direction, max_distance, occluded, ray_count, retired_counter);
		return;
	}
	if (p.bvh_width == 4) {
		static int grid4 = trace_grid_size((const void *)kernel_trace_shadow_bvh4_explicit);
		hipLaunchKernelGGL(kernel_trace_shadow_bvh4_explicit, dim3(grid4), dim3(RT_TRACE_BLOCK), 0, stream, p, origin, direction, max_distance, occluded, ray_count, retired_counter);
		return;
	}
	static int grid = trace_grid_size((const void *)kernel_trace_shadow_bvh8_explicit);
	hipLaunchKernelGGL(kernel_trace_shadow_bvh8_explicit, dim3(grid), dim3(RT_TRACE_BLOCK), 0, stream, p, origin, direction, max_distance, occluded, ray_count, retired_counter);
}
