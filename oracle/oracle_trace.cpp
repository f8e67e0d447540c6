// oracle_trace.cpp -- CPU restatement of the reference's ray traversal. TEST INFRASTRUCTURE ONLY.
//
// Follows, function by function:
//   ray_get_octant_inv4, bvh8_node_intersect, bvh8_trace, bvh8_trace_shadow  CUDA/Raytracing/BVH8.h:5-444
//   AABB::intersects, bvh2_trace, bvh2_trace_shadow                         CUDA/Raytracing/BVH2.h:4-244
//   triangle_get_positions, triangle_intersect(_shadow)                     CUDA/Raytracing/Triangle.h:21-33,148-198
//   matrix3x4_transform_*, mesh_get_transform_inv, bvh_get_mesh_root_index   CUDA/Raytracing/Mesh.h:9-55, BVH.h:49-55
//   HitBuffer::set                                                           CUDA/Buffers.h:25-32
//
// One ray at a time, in index order. The reference's persistent-thread scheduling
// (dynamic fetch, triangle postponing, BVH8.h:110-111,200,234-240,271-272) only reorders
// work between lanes of a warp; it is not part of the per-ray algorithm and is omitted.
#include "oracle.h"
#include "oracle_math.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <omp.h>

namespace {

struct Ray { float3 origin, direction; };

struct RayHit {
	float t, u, v;
	int mesh_id, triangle_id;
};

// ---- arithmetic contract (see oracle.h): explicit fused forms used by traversal -----------
inline float dot_fma(float3 a, float3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline float3 cross_fma(float3 a, float3 b) {
	return make_float3(
		fmaf(a.y, b.z, -(a.z * b.y)),
		fmaf(a.z, b.x, -(a.x * b.z)),
		fmaf(a.x, b.y, -(a.y * b.x)));
}

// Mesh.h:9-27
inline float3 transform_position(const float * m, float3 p) {
	return make_float3(
		fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[ 2], p.z, m[ 3]))),
		fmaf(m[4], p.x, fmaf(m[5], p.y, fmaf(m[ 6], p.z, m[ 7]))),
		fmaf(m[8], p.x, fmaf(m[9], p.y, fmaf(m[10], p.z, m[11]))));
}
inline float3 transform_direction(const float * m, float3 d) {
	return make_float3(
		fmaf(m[0], d.x, fmaf(m[1], d.y, m[ 2] * d.z)),
		fmaf(m[4], d.x, fmaf(m[5], d.y, m[ 6] * d.z)),
		fmaf(m[8], d.x, fmaf(m[9], d.y, m[10] * d.z)));
}

inline unsigned msb(unsigned x) { return 31u - unsigned(__builtin_clz(x)); } // Util.h:295-299 (bfind)
inline unsigned extract_byte(unsigned x, unsigned i) { return (x >> (i * 8)) & 0xff; }
inline unsigned sign_extend_s8x4(unsigned x) { // Util.h:280-284 (prmt 0xBA98): byte MSB -> 0xff / 0x00
	return ((x >> 7) & 0x01010101u) * 0xffu;
}

// BVH8.h:5-10
inline unsigned ray_get_octant_inv4(float3 d) {
	return (d.x < 0.0f ? 0 : 0x04040404) | (d.y < 0.0f ? 0 : 0x02020202) | (d.z < 0.0f ? 0 : 0x01010101);
}

struct Counters { uint64_t nodes = 0, triangles = 0, inst_xform = 0, inst_ident = 0, groups_skipped = 0; };

// Triangle.h:148-174. f = 1/a is an IEEE division (the reference's fast-math reciprocal
// is not specified); acceptance tests are literal: 0<=u<=1, v>=0, u+v<=1, 0<t<t_best.
inline void triangle_intersect(const oracle_scene & s, int mesh_id, int triangle_id, const Ray & ray, RayHit & hit) {
	const float * tri = s.triangles + size_t(triangle_id) * 24;
	float3 p0 = make_float3(tri[0], tri[1], tri[2]);
	float3 e1 = make_float3(tri[3], tri[4], tri[5]);
	float3 e2 = make_float3(tri[6], tri[7], tri[8]);

	float3 h = cross_fma(ray.direction, e2);
	float  a = dot_fma(e1, h);
	float  f = 1.0f / a;
	float3 sv = ray.origin - p0;
	float  u = f * dot_fma(sv, h);
	if (u >= 0.0f && u <= 1.0f) {
		float3 q = cross_fma(sv, e1);
		float  v = f * dot_fma(ray.direction, q);
		if (v >= 0.0f && u + v <= 1.0f) {
			float t = f * dot_fma(e2, q);
			if (t > 0.0f && t < hit.t) {
				hit.t = t; hit.u = u; hit.v = v;
				hit.mesh_id = mesh_id;
				hit.triangle_id = triangle_id;
			}
		}
	}
}

// Triangle.h:176-198
inline bool triangle_intersect_shadow(const oracle_scene & s, int triangle_id, const Ray & ray, float max_distance) {
	const float * tri = s.triangles + size_t(triangle_id) * 24;
	float3 p0 = make_float3(tri[0], tri[1], tri[2]);
	float3 e1 = make_float3(tri[3], tri[4], tri[5]);
	float3 e2 = make_float3(tri[6], tri[7], tri[8]);

	float3 h = cross_fma(ray.direction, e2);
	float  a = dot_fma(e1, h);
	float  f = 1.0f / a;
	float3 sv = ray.origin - p0;
	float  u = f * dot_fma(sv, h);
	if (u >= 0.0f && u <= 1.0f) {
		float3 q = cross_fma(sv, e1);
		float  v = f * dot_fma(ray.direction, q);
		if (v >= 0.0f && u + v <= 1.0f) {
			float t = f * dot_fma(e2, q);
			if (t > 0.0f && t < max_distance) return true;
		}
	}
	return false;
}

// BVH8.h:29-107. The three "/ direction" of the reference become one IEEE reciprocal per ray
// (inv_dir) times exact power-of-two scales; min/max use IEEE maxNum/minNum (NaN slabs are
// ignored) where the reference compares float bit patterns as integers (Util.h:303-341).
// bound_bits (optional; not in the reference): the 16-bit lower bound of the entry distances of the children the ray enters EXCEPT the one it visits first,
// in bits 8..23 -- "skip behind the hit", kernels_trace.hip (skip_bound_bits) restated: keys = tmin with the low byte replaced by the child's bit index,
// compared as signed integers; the two smallest are kept.
inline unsigned bvh8_node_intersect(const Ray & ray, float3 inv_dir, unsigned oct_inv4, float max_distance, const uint8_t * node, unsigned * bound_bits = nullptr) {
	uint32_t w[20];
	memcpy(w, node, 80);

	float3 p = make_float3(uint_as_float(w[0]), uint_as_float(w[1]), uint_as_float(w[2]));
	unsigned e_imask = w[3];
	unsigned e_x = extract_byte(e_imask, 0), e_y = extract_byte(e_imask, 1), e_z = extract_byte(e_imask, 2);

	float3 adjusted_dir_inv = make_float3(
		uint_as_float(e_x << 23) * inv_dir.x,
		uint_as_float(e_y << 23) * inv_dir.y,
		uint_as_float(e_z << 23) * inv_dir.z);
	float3 adjusted_origin = (p - ray.origin) * inv_dir;

	unsigned hit_mask = 0;
	int32_t least = 0x7fffffff, second = 0x7fffffff;
	static const int variant = getenv("ORACLE_SKIP_VARIANT") ? atoi(getenv("ORACLE_SKIP_VARIANT")) : 2;
	int32_t half_min[2] = { 0x7fffffff, 0x7fffffff };
	for (int i = 0; i < 2; i++) {
		unsigned meta4 = w[6 + i];

		unsigned is_inner4   = (meta4 & (meta4 << 1)) & 0x10101010;
		unsigned inner_mask4 = sign_extend_s8x4(is_inner4 << 3);
		unsigned bit_index4  = (meta4 ^ (oct_inv4 & inner_mask4)) & 0x1f1f1f1f;
		unsigned child_bits4 = (meta4 >> 5) & 0x07070707;

		unsigned q_lo_x = w[ 8 + i], q_hi_x = w[10 + i];
		unsigned q_lo_y = w[12 + i], q_hi_y = w[14 + i];
		unsigned q_lo_z = w[16 + i], q_hi_z = w[18 + i];

		unsigned x_min = ray.direction.x < 0.0f ? q_hi_x : q_lo_x, x_max = ray.direction.x < 0.0f ? q_lo_x : q_hi_x;
		unsigned y_min = ray.direction.y < 0.0f ? q_hi_y : q_lo_y, y_max = ray.direction.y < 0.0f ? q_lo_y : q_hi_y;
		unsigned z_min = ray.direction.z < 0.0f ? q_hi_z : q_lo_z, z_max = ray.direction.z < 0.0f ? q_lo_z : q_hi_z;

		for (int j = 0; j < 4; j++) {
			float tx0 = fmaf(float(extract_byte(x_min, j)), adjusted_dir_inv.x, adjusted_origin.x);
			float ty0 = fmaf(float(extract_byte(y_min, j)), adjusted_dir_inv.y, adjusted_origin.y);
			float tz0 = fmaf(float(extract_byte(z_min, j)), adjusted_dir_inv.z, adjusted_origin.z);
			float tx1 = fmaf(float(extract_byte(x_max, j)), adjusted_dir_inv.x, adjusted_origin.x);
			float ty1 = fmaf(float(extract_byte(y_max, j)), adjusted_dir_inv.y, adjusted_origin.y);
			float tz1 = fmaf(float(extract_byte(z_max, j)), adjusted_dir_inv.z, adjusted_origin.z);

			float tmin = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, 0.0f));
			float tmax = fminf(fminf(tx1, ty1), fminf(tz1, max_distance));

			if (tmin < tmax) {
				unsigned child_bits = extract_byte(child_bits4, j);
				unsigned bit_index  = extract_byte(bit_index4,  j);
				hit_mask |= child_bits << bit_index;
				if (bound_bits) {
					int32_t key = int32_t((float_as_uint(tmin) & 0xffffff00u) | bit_index);   // tmin >= 0: the bit patterns order like the values (a -0 sorts first: conservative)
					second = key < least ? least : (key < second ? key : second);           // (the median of the three)
					least  = key < least ? key : least;
					if (variant == 4 ? bit_index >= 24 : true) half_min[i] = key < half_min[i] ? key : half_min[i];
				}
			}
		}
	}
	if (bound_bits) {
		// the smallest key that is not the key of the child visited first (the highest bit of the mask)
		int32_t key = (uint32_t(least) & 0x1fu) == (hit_mask ? msb(hit_mask) : 0xffu) ? second : least;
		*bound_bits = (uint32_t(key) >> 8) & 0x00ffff00u;
		if (variant >= 3) { int far = (oct_inv4 >> 2) & 1; *bound_bits = (uint32_t(half_min[far]) >> 8) & 0x00ffff00u; }
	}
	return hit_mask;
}

struct Group { unsigned x, y; };

constexpr int ORACLE_STACK_SIZE = 128; // the reference has 32 entries and no overflow check (Common.h:103)

inline float3 reciprocal(float3 d) { return make_float3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

// Shared body of bvh8_trace (closest hit) and bvh8_trace_shadow (any hit).
template<bool SHADOW>
inline bool bvh8_traverse(const oracle_scene & s, Ray ray, float max_distance, RayHit & ray_hit, Counters & c) {
	Group stack[ORACLE_STACK_SIZE];
	int stack_size = 0;

	Ray ray_untransformed = ray;
	float3 inv_dir = reciprocal(ray.direction);
	unsigned oct_inv4 = ray_get_octant_inv4(ray.direction);

	Group current_group = { 0, 0x80000000u };
	// The walk of the flattened scene's engine (kernels_trace.hip, rt_set_skip_behind_hit): closest-hit rays in a one-tree scene only.
	const bool skip = !SHADOW && s.skip_behind_hit != 0 && s.static_whole_scene;

	int  tlas_stack_size = RT_INVALID;
	int  mesh_id = 0;
	bool mesh_has_identity_transform = true;

	auto push = [&](Group g) { if (stack_size >= ORACLE_STACK_SIZE) { fprintf(stderr, "oracle: traversal stack overflow\n"); abort(); } stack[stack_size++] = g; };

	if (s.static_whole_scene) tlas_stack_size = 0; // rt_set_static_geometry: node 0 is the root of the one world-space tree, the ray is inside it (row 0) from the start

	while (true) {
		Group triangle_group;

		if (current_group.y & 0xff000000u) {
			unsigned hits_imask = current_group.y;
			unsigned child_index_offset = msb(hits_imask);
			unsigned child_index_base   = current_group.x;

			current_group.y &= ~(1u << child_index_offset);
			if (current_group.y & 0xff000000u) push(current_group);

			unsigned slot_index     = (child_index_offset - 24) ^ (oct_inv4 & 0xff);
			unsigned relative_index = unsigned(__builtin_popcount(hits_imask & ~(0xffffffffu << slot_index)));
			unsigned child_node_index = child_index_base + relative_index;

			const uint8_t * node = s.bvh8_nodes + size_t(child_node_index) * 80;
			c.nodes++;

			float limit = SHADOW ? max_distance : ray_hit.t;
			unsigned bound_bits = 0;
			unsigned hitmask = bvh8_node_intersect(ray, inv_dir, oct_inv4, limit, node, skip ? &bound_bits : nullptr);

			uint32_t node_w3, node_w4, node_w5;
			memcpy(&node_w3, node + 12, 4); memcpy(&node_w4, node + 16, 4); memcpy(&node_w5, node + 20, 4);
			unsigned imask = extract_byte(node_w3, 3);

			current_group .x = node_w4; // child    base offset
			triangle_group.x = node_w5; // triangle base offset
			current_group .y = (hitmask & 0xff000000u) | imask;
			if (skip) current_group.y |= bound_bits;   // (the 16 bits of the mask word that the reference leaves empty)
			triangle_group.y = (hitmask & 0x00ffffffu);
		} else {
			triangle_group = current_group;
			current_group  = { 0, 0 };
		}

		while (triangle_group.y != 0) {
			if (tlas_stack_size == RT_INVALID) {
				// In the TLAS a "triangle" is a mesh: descend into its BLAS (BVH8.h:204-232)
				int mesh_offset = int(msb(triangle_group.y));
				triangle_group.y &= ~(1u << mesh_offset);
				mesh_id = int(triangle_group.x) + mesh_offset;

				if (triangle_group.y != 0)          push(triangle_group);
				if (current_group.y & 0xff000000u)  push(current_group);
				tlas_stack_size = stack_size;

				unsigned root = unsigned(s.mesh_bvh_root_indices[mesh_id]);
				mesh_has_identity_transform = root >> 31;
				unsigned root_index = root & 0x7fffffffu;

				if (!mesh_has_identity_transform) {
					const float * m = s.mesh_transforms_inv + size_t(mesh_id) * 12;
					ray.origin    = transform_position (m, ray.origin);
					ray.direction = transform_direction(m, ray.direction); // not renormalised: t stays in world units
					inv_dir  = reciprocal(ray.direction);
					oct_inv4 = ray_get_octant_inv4(ray.direction);
					c.inst_xform++;
				} else {
					c.inst_ident++;
				}
				current_group = { root_index, 0x80000000u };
				break;
			} else {
				int triangle_index = int(msb(triangle_group.y));
				triangle_group.y &= ~(1u << triangle_index);
				c.triangles++;
				if (SHADOW) {
					if (triangle_intersect_shadow(s, int(triangle_group.x) + triangle_index, ray, max_distance)) return true;
				} else {
					triangle_intersect(s, mesh_id, int(triangle_group.x) + triangle_index, ray, ray_hit);
				}
			}
		}

		if ((current_group.y & 0xff000000u) == 0) {
			if (stack_size == 0) return false; // closest: ray_hit holds the result; shadow: nothing hit

			if (stack_size == tlas_stack_size) {
				tlas_stack_size = RT_INVALID;
				if (!mesh_has_identity_transform) {
					ray = ray_untransformed;
					inv_dir  = reciprocal(ray.direction);
					oct_inv4 = ray_get_octant_inv4(ray.direction);
				}
			}
			current_group = stack[--stack_size];
			// Skip behind the hit: every child left in the group is entered at or beyond the group's bound; at or beyond the hit
			// already held none of them can be entered (their test would be tmin < tmax <= hit.t), so the group is dropped unvisited.
			static const int variant = getenv("ORACLE_SKIP_VARIANT") ? atoi(getenv("ORACLE_SKIP_VARIANT")) : 2;
			while (skip && variant >= 3) {
				if (int32_t((current_group.y << 8) & 0xffff0000u) >= int32_t(float_as_uint(ray_hit.t))) current_group.y &= ~0x0f000000u;
				if (current_group.y & 0xff000000u) break;
				c.groups_skipped++;
				if (stack_size == 0) return false;
				current_group = stack[--stack_size];
			}
			while (skip && variant < 3 && int32_t((current_group.y << 8) & 0xffff0000u) >= int32_t(float_as_uint(ray_hit.t))) {
				c.groups_skipped++;
				if (stack_size == 0) return false;
				current_group = stack[--stack_size];
			}
		}
	}
}

// ---- BVH2 ------------------------------------------------------------------------------------

struct Node2 { float min[3], max[3]; int left_or_first; unsigned count_axis; };

// BVH2.h:8-16 with t = (plane - origin) * inv_dir
inline bool aabb_intersects(const Node2 & n, const Ray & ray, float3 inv_dir, float max_distance) {
	float t0x = (n.min[0] - ray.origin.x) * inv_dir.x, t1x = (n.max[0] - ray.origin.x) * inv_dir.x;
	float t0y = (n.min[1] - ray.origin.y) * inv_dir.y, t1y = (n.max[1] - ray.origin.y) * inv_dir.y;
	float t0z = (n.min[2] - ray.origin.z) * inv_dir.z, t1z = (n.max[2] - ray.origin.z) * inv_dir.z;
	float t_near = fmaxf(fminf(t0x, t1x), fmaxf(fminf(t0y, t1y), fmaxf(fminf(t0z, t1z), 0.0f)));
	float t_far  = fminf(fmaxf(t0x, t1x), fminf(fmaxf(t0y, t1y), fminf(fmaxf(t0z, t1z), max_distance)));
	return t_near < t_far;
}

template<bool SHADOW>
inline bool bvh2_traverse(const oracle_scene & s, Ray ray, float max_distance, RayHit & ray_hit, Counters & c) {
	int stack[ORACLE_STACK_SIZE];
	int stack_size = 1;
	stack[0] = 0;

	Ray ray_untransformed = ray;
	float3 inv_dir = reciprocal(ray.direction);

	int  tlas_stack_size = RT_INVALID;
	int  mesh_id = 0;
	bool mesh_has_identity_transform = true;

	const Node2 * nodes = reinterpret_cast<const Node2 *>(s.bvh2_nodes);

	while (true) {
		if (stack_size == tlas_stack_size) {
			tlas_stack_size = RT_INVALID;
			if (!mesh_has_identity_transform) { ray = ray_untransformed; inv_dir = reciprocal(ray.direction); }
		}
		int node_index = stack[--stack_size];
		Node2 node;
		memcpy(&node, &nodes[node_index], sizeof(Node2));
		unsigned count = node.count_axis & 0x3fffffffu, axis = node.count_axis >> 30;
		c.nodes++;

		if (aabb_intersects(node, ray, inv_dir, SHADOW ? max_distance : ray_hit.t)) {
			if (count > 0) {
				if (tlas_stack_size == RT_INVALID) {
					tlas_stack_size = stack_size;
					mesh_id = node.left_or_first;
					unsigned root = unsigned(s.mesh_bvh_root_indices[mesh_id]);
					mesh_has_identity_transform = root >> 31;
					if (!mesh_has_identity_transform) {
						const float * m = s.mesh_transforms_inv + size_t(mesh_id) * 12;
						ray.origin    = transform_position (m, ray.origin);
						ray.direction = transform_direction(m, ray.direction);
						inv_dir = reciprocal(ray.direction);
						c.inst_xform++;
					} else c.inst_ident++;
					stack[stack_size++] = int(root & 0x7fffffffu);
				} else {
					for (int i = node.left_or_first; i < node.left_or_first + int(count); i++) {
						c.triangles++;
						if (SHADOW) { if (triangle_intersect_shadow(s, i, ray, max_distance)) return true; }
						else triangle_intersect(s, mesh_id, i, ray, ray_hit);
					}
				}
			} else {
				float d = axis == 0 ? ray.direction.x : (axis == 1 ? ray.direction.y : ray.direction.z);
				bool left_first = d > 0.0f;
				int first  = left_first ? node.left_or_first     : node.left_or_first + 1;
				int second = left_first ? node.left_or_first + 1 : node.left_or_first;
				if (stack_size + 2 > ORACLE_STACK_SIZE) { fprintf(stderr, "oracle: traversal stack overflow\n"); abort(); }
				stack[stack_size++] = second;
				stack[stack_size++] = first;
			}
		}
		if (stack_size == 0) return false;
	}
}


// ---- BVH4 (CUDA/Raytracing/BVH4.h:4-295) ------------------------------------------------------

struct Node4 { float min_x[4], min_y[4], min_z[4], max_x[4], max_y[4], max_z[4]; int index_and_count[4][2]; };

struct AABBHits4 { float t_near[4]; bool hit[4]; };

// bvh4_node_intersect (BVH4.h:22-67) with t = (plane - origin) * inv_dir (the arithmetic contract of
// this repository): slab test of the four children, then the near distances are tagged with
// the child id in their two low mantissa bits and sorted in descending order.
inline AABBHits4 bvh4_node_intersect(const Node4 & n, const Ray & ray, float3 inv_dir, float max_distance) {
	AABBHits4 r;
	for (int i = 0; i < 4; i++) {
		float t0x = (n.min_x[i] - ray.origin.x) * inv_dir.x, t1x = (n.max_x[i] - ray.origin.x) * inv_dir.x;
		float t0y = (n.min_y[i] - ray.origin.y) * inv_dir.y, t1y = (n.max_y[i] - ray.origin.y) * inv_dir.y;
		float t0z = (n.min_z[i] - ray.origin.z) * inv_dir.z, t1z = (n.max_z[i] - ray.origin.z) * inv_dir.z;
		r.t_near[i] = fmaxf(fminf(t0x, t1x), fmaxf(fminf(t0y, t1y), fmaxf(fminf(t0z, t1z), 0.0f)));
		float t_far = fminf(fmaxf(t0x, t1x), fminf(fmaxf(t0y, t1y), fminf(fmaxf(t0z, t1z), max_distance)));
		r.hit[i] = r.t_near[i] < t_far;
	}
	for (int i = 0; i < 4; i++) r.t_near[i] = uint_as_float((float_as_uint(r.t_near[i]) & 0xfffffffcu) | unsigned(i));
	for (int i = 1; i < 4; i++) for (int j = i - 1; j >= 0; j--) if (r.t_near[j] < r.t_near[j + 1]) { float t = r.t_near[j]; r.t_near[j] = r.t_near[j + 1]; r.t_near[j + 1] = t; }
	return r;
}

template<bool SHADOW>
inline bool bvh4_traverse(const oracle_scene & s, Ray ray, float max_distance, RayHit & ray_hit, Counters & c) {
	unsigned stack[ORACLE_STACK_SIZE];
	int stack_size = 1;
	stack[0] = 1; // node 1, child 0: the entry point whose child is the root (BVH4Converter.cpp:8-12)

	Ray ray_untransformed = ray;
	float3 inv_dir = reciprocal(ray.direction);

	int  tlas_stack_size = RT_INVALID;
	int  mesh_id = 0;
	bool mesh_has_identity_transform = true;

	const Node4 * nodes = reinterpret_cast<const Node4 *>(s.bvh4_nodes);

	while (true) {
		if (stack_size == tlas_stack_size) {
			tlas_stack_size = RT_INVALID;
			if (!mesh_has_identity_transform) { ray = ray_untransformed; inv_dir = reciprocal(ray.direction); }
		}
		unsigned packed = stack[--stack_size];
		int node_index = int(packed & 0x3fffffffu), node_id = int(packed >> 30);
		int index = nodes[node_index].index_and_count[node_id][0];
		int count = nodes[node_index].index_and_count[node_id][1];

		if (count > 0) {
			if (tlas_stack_size == RT_INVALID) {
				tlas_stack_size = stack_size;
				mesh_id = index;
				unsigned root = unsigned(s.mesh_bvh_root_indices[mesh_id]);
				mesh_has_identity_transform = root >> 31;
				if (!mesh_has_identity_transform) {
					const float * m = s.mesh_transforms_inv + size_t(mesh_id) * 12;
					ray.origin    = transform_position (m, ray.origin);
					ray.direction = transform_direction(m, ray.direction);
					inv_dir = reciprocal(ray.direction);
					c.inst_xform++;
				} else c.inst_ident++;
				stack[stack_size++] = (root & 0x7fffffffu) + 1u; // the BLAS's own entry node
			} else {
				for (int j = index; j < index + count; j++) {
					c.triangles++;
					if (SHADOW) { if (triangle_intersect_shadow(s, j, ray, max_distance)) return true; }
					else triangle_intersect(s, mesh_id, j, ray, ray_hit);
				}
			}
		} else {
			int child = index;
			Node4 node;
			memcpy(&node, &nodes[child], sizeof(Node4));
			c.nodes++;
			AABBHits4 hits = bvh4_node_intersect(node, ray, inv_dir, SHADOW ? max_distance : ray_hit.t);
			for (int i = 0; i < 4; i++) {
				int id = int(float_as_uint(hits.t_near[i]) & 3u);
				if (hits.hit[id]) {
					if (stack_size + 1 > ORACLE_STACK_SIZE) { fprintf(stderr, "oracle: traversal stack overflow\n"); abort(); }
					stack[stack_size++] = (unsigned(id) << 30) | unsigned(child);
				}
			}
		}
		if (stack_size == 0) return false;
	}
}

inline void store_hit(uint32_t * out, const RayHit & h) { // Buffers.h:25-32
	uint32_t uv = uint32_t(int(h.u * 65535.0f)) | (uint32_t(int(h.v * 65535.0f)) << 16);
	out[0] = uint32_t(h.mesh_id); out[1] = uint32_t(h.triangle_id); out[2] = float_as_uint(h.t); out[3] = uv;
}

} // namespace

// Exposed to the path tracing restatement (oracle_pathtrace.cpp)
void oracle_trace_one(const oracle_scene & s, float3 origin, float3 direction, uint32_t * hit4, oracle_trace_stats * stats) {
	Ray ray = { origin, direction };
	RayHit hit; hit.t = INFINITY; hit.u = 0.0f; hit.v = 0.0f; hit.mesh_id = 0; hit.triangle_id = RT_INVALID;
	Counters c;
	if (s.bvh_type == 2) bvh2_traverse<false>(s, ray, 0.0f, hit, c); else if (s.bvh_type == 4) bvh4_traverse<false>(s, ray, 0.0f, hit, c); else bvh8_traverse<false>(s, ray, 0.0f, hit, c);
	if (s.alias_mesh_ids && hit.triangle_id != RT_INVALID && s.alias_mesh_ids[hit.triangle_id] >= 0) { // a copy in the flattened static BLAS names its original
		int copy = hit.triangle_id;
		hit.mesh_id = s.alias_mesh_ids[copy]; hit.triangle_id = s.alias_triangle_ids[copy];
	}
	store_hit(hit4, hit);
	if (stats) { stats->nodes += c.nodes; stats->triangles += c.triangles; stats->instances_transformed += c.inst_xform; stats->instances_identity += c.inst_ident; stats->rays++; }
}

bool oracle_trace_shadow_one(const oracle_scene & s, float3 origin, float3 direction, float max_distance, oracle_trace_stats * stats) {
	Ray ray = { origin, direction };
	RayHit hit; hit.t = INFINITY; hit.u = hit.v = 0.0f; hit.mesh_id = 0; hit.triangle_id = RT_INVALID;
	Counters c;
	bool occluded = s.bvh_type == 2 ? bvh2_traverse<true>(s, ray, max_distance, hit, c) : (s.bvh_type == 4 ? bvh4_traverse<true>(s, ray, max_distance, hit, c) : bvh8_traverse<true>(s, ray, max_distance, hit, c));
	if (stats) { stats->nodes += c.nodes; stats->triangles += c.triangles; stats->instances_transformed += c.inst_xform; stats->instances_identity += c.inst_ident; stats->rays++; }
	return occluded;
}

extern "C" {

const char * oracle_version(void) { return "gpu-raytracer oracle 0.1 (CPU restatement; test infrastructure)"; }

void oracle_trace(const oracle_scene * scene, const float * ox, const float * oy, const float * oz,
                  const float * dx, const float * dy, const float * dz, size_t ray_count,
                  uint32_t * hits, oracle_trace_stats * stats, int threads) {
	if (threads <= 0) threads = oracle_default_threads();
	oracle_trace_stats total = { 0, 0, 0, 0, 0 };
	#pragma omp parallel num_threads(threads)
	{
		oracle_trace_stats local = { 0, 0, 0, 0, 0 };
		#pragma omp for schedule(dynamic, 1024)
		for (long long i = 0; i < (long long)ray_count; i++) {
			oracle_trace_one(*scene, make_float3(ox[i], oy[i], oz[i]), make_float3(dx[i], dy[i], dz[i]), hits + 4 * i, &local);
		}
		#pragma omp critical
		{ total.nodes += local.nodes; total.triangles += local.triangles; total.instances_transformed += local.instances_transformed; total.instances_identity += local.instances_identity; total.rays += local.rays; }
	}
	if (stats) *stats = total;
}

void oracle_trace_shadow(const oracle_scene * scene, const float * ox, const float * oy, const float * oz,
                         const float * dx, const float * dy, const float * dz, const float * max_distance,
                         size_t ray_count, uint8_t * occluded, oracle_trace_stats * stats, int threads) {
	if (threads <= 0) threads = oracle_default_threads();
	oracle_trace_stats total = { 0, 0, 0, 0, 0 };
	#pragma omp parallel num_threads(threads)
	{
		oracle_trace_stats local = { 0, 0, 0, 0, 0 };
		#pragma omp for schedule(dynamic, 1024)
		for (long long i = 0; i < (long long)ray_count; i++) {
			occluded[i] = oracle_trace_shadow_one(*scene, make_float3(ox[i], oy[i], oz[i]), make_float3(dx[i], dy[i], dz[i]), max_distance[i], &local) ? 1 : 0;
		}
		#pragma omp critical
		{ total.nodes += local.nodes; total.triangles += local.triangles; total.instances_transformed += local.instances_transformed; total.instances_identity += local.instances_identity; total.rays += local.rays; }
	}
	if (stats) *stats = total;
}

} // extern "C"
