// SlotOrder.cpp -- the order in which rays visit a CWBVH node's children, learned from sample rays.
//
// The traversal kernels (csrc/kernels_trace.hip, like the reference's BVH8.h:113-274) do not sort a node's children by distance: a ray
// whose direction signs are v visits the inner children by falling (slot ^ v), so WHICH slot a child sits in decides how early a ray
// meets it. The reference's converter deals the slots greedily by where the children's centres lie (BVH8Converter.cpp:146-205); the
// flattened tree of DESIGN.md 4.6 is this renderer's own layout and may seat them any way it likes. What a bad seat costs: a ray that
// ends inside child H of a node walks every other child that is seated in front of H for its octant and whose box it enters -- also
// those whose box begins BEHIND the hit, which it would have skipped had it been to H first. That is counted here, on the tree itself:
//   1  sample rays (seeded, a pure function of the input): from area-weighted points on the triangles, cosine-distributed about the
//      normal (what a bounce is) or towards another such point and no further (what a shadow ray is); half of the budget from points of
//      the FREE space (somewhere along such a bounce) in uniformly random directions (what a camera standing there would send) -- and, when
//      the caller hands over its camera, a quarter as the paths it is about to trace: a ray through a random pixel and up to three bounces;
//   2  each is traced for its closest hit; along the path from the root to the leaf that holds the hit, every inner child c that the
//      ray enters only BEHIND the hit scores one for the pair (c in front of H) in the ray's octant;
//   3  per node, the children trade slots while that lowers the score summed over the eight octants; the records of its inner children
//      are re-ordered to match (a child's index is its parent's base + its rank among the inner slots).
// Boxes, triangles and leaf contents do not change: closest hits stay what they were (up to exact ties in t), tests/test_static_geometry.py.
// Measured: profiles/r05_slot_assignment.txt.
#include "BVH.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

namespace {

inline float bits_to_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

inline uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
struct Random {   // counter-based: sample i is the same on every thread count
	uint64_t state;
	explicit Random(uint64_t seed) : state(seed) { }
	float next() { state = mix(state); return float(state >> 40) * (1.0f / 16777216.0f); }   // [0, 1)
};

struct SampleRay { float o[3], d[3], inv[3], tmax; int octant; };   // octant: bit 2 / 1 / 0 set where d.x / d.y / d.z is not negative (the kernels' oct_inv)

struct ChildBox { float lo[3], hi[3]; };
inline ChildBox child_box(const BVHNode8 & n, int slot) {
	const float scale[3] = { bits_to_float(unsigned(n.e[0]) << 23), bits_to_float(unsigned(n.e[1]) << 23), bits_to_float(unsigned(n.e[2]) << 23) };
	ChildBox b;
	b.lo[0] = n.p.x + float(n.quantized_min_x[slot]) * scale[0]; b.hi[0] = n.p.x + float(n.quantized_max_x[slot]) * scale[0];
	b.lo[1] = n.p.y + float(n.quantized_min_y[slot]) * scale[1]; b.hi[1] = n.p.y + float(n.quantized_max_y[slot]) * scale[1];
	b.lo[2] = n.p.z + float(n.quantized_min_z[slot]) * scale[2]; b.hi[2] = n.p.z + float(n.quantized_max_z[slot]) * scale[2];
	return b;
}
// where the ray enters the box, if it does before `limit`
inline bool enters(const SampleRay & r, const ChildBox & b, float limit, float & t_enter) {
	float t0 = 0.0f, t1 = limit;
	for (int d = 0; d < 3; d++) {
		float a = (b.lo[d] - r.o[d]) * r.inv[d], c = (b.hi[d] - r.o[d]) * r.inv[d];
		if (a > c) std::swap(a, c);
		t0 = std::max(t0, a); t1 = std::min(t1, c);
	}
	t_enter = t0;
	return t0 < t1;
}

struct Learner {
	const BVH8 & bvh;
	const std::vector<Triangle> & triangles;
	// Who holds a leaf position: per node its parent, per leaf position the node whose leaf it is in. (Rounds 5's form -- "a subtree's leaves are one run of
	// positions" -- holds for the host's depth-first builders only; the device's builder emits leaves level by level, a subtree's positions are scattered.)
	std::vector<unsigned> parent, leaf_owner;
	std::unique_ptr<std::atomic<unsigned>[]> score;   // [score_row[node]][octant][c][h]: rays of that octant that ended in child h and enter child c only behind their hit
	std::vector<int> score_row;                       // per node: its row of 512 counters, or -1 (a tree too large to score whole keeps rows for its top levels: memory_limit)
	std::vector<float> area_cdf; float scene_size = 1.0f;
	bool weigh_by_work = true;
	const SlotLearningView * view = nullptr;

	Learner(const BVH8 & bvh, const std::vector<Triangle> & triangles) : bvh(bvh), triangles(triangles) { }

	void index_leaves() {
		parent.assign(bvh.nodes.size(), ~0u); leaf_owner.assign(bvh.indices.size(), ~0u);
		std::vector<unsigned> order; order.reserve(bvh.nodes.size()); order.push_back(0u);
		for (size_t k = 0; k < order.size(); k++) {
			const unsigned node = order[k]; const BVHNode8 & n = bvh.nodes[node];
			for (int s = 0; s < 8; s++) if (!((n.imask >> s) & 1) && n.meta[s]) {
				const unsigned first = n.base_index_triangle + unsigned(n.meta[s] & 31u), count = unsigned(__builtin_popcount(unsigned(n.meta[s]) >> 5));
				for (unsigned t = first; t < first + count && t < leaf_owner.size(); t++) leaf_owner[t] = node;
			}
			for (int c = 0; c < __builtin_popcount(unsigned(n.imask)); c++) { const unsigned child = n.base_index_child + unsigned(c); if (child < bvh.nodes.size() && parent[child] == ~0u && child != 0) { parent[child] = node; order.push_back(child); } }
		}
	}

	// closest hit below `root`; returns the leaf position (index into bvh.indices) or -1, and what the walk cost (2 per node step, 1 per triangle test)
	int trace(const SampleRay & r, float & t_hit, unsigned root = 0, unsigned * work = nullptr) const {
		unsigned stack[256]; int sp = 0; stack[sp++] = root;
		int hit = -1; t_hit = r.tmax;
		unsigned spent = 0;
		while (sp) {
			const BVHNode8 & n = bvh.nodes[stack[--sp]];
			spent += 2;
			unsigned inner_hit = 0;   // bit (slot ^ octant): visiting order = falling bit
			for (int s = 0; s < 8; s++) {
				if (!n.meta[s]) continue;
				float t_enter;
				if (!enters(r, child_box(n, s), t_hit, t_enter)) continue;
				if ((n.imask >> s) & 1) { inner_hit |= 1u << (s ^ r.octant); continue; }
				const unsigned first = n.base_index_triangle + unsigned(n.meta[s] & 31u), count = unsigned(__builtin_popcount(unsigned(n.meta[s]) >> 5));
				spent += count;
				for (unsigned k = 0; k < count; k++) {
					const Triangle & tri = triangles[size_t(bvh.indices[first + k])];
					const Vector3 e1 = tri.position_1 - tri.position_0, e2 = tri.position_2 - tri.position_0;
					const Vector3 dir(r.d[0], r.d[1], r.d[2]), org(r.o[0], r.o[1], r.o[2]);
					const Vector3 h = Vector3::cross(dir, e2);
					const float a = Vector3::dot(e1, h), f = 1.0f / a;
					const Vector3 s0 = org - tri.position_0;
					const float u = f * Vector3::dot(s0, h);
					if (!(u >= 0.0f && u <= 1.0f)) continue;
					const Vector3 q = Vector3::cross(s0, e1);
					const float v = f * Vector3::dot(dir, q);
					if (!(v >= 0.0f && u + v <= 1.0f)) continue;
					const float t = f * Vector3::dot(e2, q);
					if (t > 0.0f && t < t_hit) { t_hit = t; hit = int(first + k); }
				}
			}
			// push so that the child with the highest bit is popped first
			for (int bit = 0; bit < 8; bit++) if ((inner_hit >> bit) & 1u) {
				const int s = bit ^ r.octant;
				if (sp < 256) stack[sp++] = n.base_index_child + unsigned(__builtin_popcount(unsigned(n.imask) & ((1u << s) - 1u)));
			}
		}
		if (work) *work = spent;
		return hit;
	}

	// returns the leaf position of the ray's closest hit (or -1) and its distance
	int learn_from(const SampleRay & r, float & t_hit) {
		t_hit = r.tmax;
		if (!std::isfinite(r.o[0] + r.o[1] + r.o[2]) || !std::isfinite(r.d[0] + r.d[1] + r.d[2])) return -1;   // (a ray of NaNs enters every box: non-finite vertices must not cost a walk of the whole tree per sample)
		const int hit = trace(r, t_hit);
		if (hit < 0 || size_t(hit) >= leaf_owner.size() || leaf_owner[size_t(hit)] == ~0u) return hit;
		// the nodes from the root to the one that holds the hit
		unsigned path[65]; int length = 0;
		for (unsigned node = leaf_owner[size_t(hit)]; node != ~0u && length < 65; node = parent[node]) path[length++] = node;
		if (length == 0 || length >= 65 || path[length - 1] != 0u) return hit;
		for (int depth = 0; depth + 1 < length; depth++) {
			const unsigned node = path[length - 1 - depth], holder_node = path[length - 2 - depth];
			const BVHNode8 & n = bvh.nodes[node];
			int holder = -1; unsigned rank = 0;
			for (int s = 0; s < 8; s++) if ((n.imask >> s) & 1) { if (n.base_index_child + rank == holder_node) holder = s; rank++; }
			if (holder < 0) return hit;
			for (int s = 0; s < 8; s++) if (((n.imask >> s) & 1) && s != holder) {
				float t_enter;
				if (score_row[node] >= 0 && enters(r, child_box(n, s), r.tmax, t_enter) && t_enter >= t_hit) {
					// what the detour costs: the walk of that child's subtree by a ray that has not found its hit yet
					unsigned work = 2; float unused;
					if (weigh_by_work) (void)trace(r, unused, n.base_index_child + unsigned(__builtin_popcount(unsigned(n.imask) & ((1u << s) - 1u))), &work);
					// (a counter is 32 bits wide and the root's cells see every ray: at most 8 M rays of at most 511 each cannot wrap it -- a detour that costs more than
					// 255 node steps weighs as one that costs 255: advisor finding, round 5)
					score[((size_t(score_row[node]) * 8 + size_t(r.octant)) * 8 + size_t(s)) * 8 + size_t(holder)].fetch_add(std::min(work, 511u), std::memory_order_relaxed);
				}
			}
		}
		return hit;
	}

	void surface_point(Random & rng, Vector3 & point, Vector3 & normal) const {
		const float pick = rng.next() * area_cdf.back();
		const size_t i = std::min(size_t(std::lower_bound(area_cdf.begin(), area_cdf.end(), pick) - area_cdf.begin()), triangles.size() - 1);
		const Triangle & t = triangles[i];
		float u = rng.next(), v = rng.next();
		if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
		point = t.position_0 + (t.position_1 - t.position_0) * u + (t.position_2 - t.position_0) * v;
		normal = Vector3::cross(t.position_1 - t.position_0, t.position_2 - t.position_0);
		const float length = Vector3::length(normal);
		normal = length > 0.0f ? normal * (1.0f / length) : Vector3(0.0f, 1.0f, 0.0f);
	}

	bool make_ray(uint64_t index, SampleRay & r) const {
		Random rng(mix(index * 2654435761ull + 12345ull));
		Vector3 from, normal; surface_point(rng, from, normal);
		if (rng.next() < 0.5f) normal = normal * -1.0f;   // either side of the triangle
		Vector3 dir; float tmax = INFINITY;
		const int kind = int(index & 1ull);
		if (kind == 1) {   // towards another point of the surface, and no further
			Vector3 to, unused; surface_point(rng, to, unused);
			dir = to - from;
			const float distance = Vector3::length(dir);
			if (!(distance > 1.0e-4f * scene_size)) return false;
			dir = dir * (1.0f / distance); tmax = distance * (1.0f - 1.0e-3f);
			if (Vector3::dot(dir, normal) < 0.0f) normal = normal * -1.0f;
		} else {              // cosine-distributed about the normal
			dir = cosine_direction(rng, normal);
		}
		set_ray(r, from + normal * (1.0e-4f * scene_size), dir, tmax);
		return true;
	}
	static void set_ray(SampleRay & r, const Vector3 & origin, const Vector3 & dir, float tmax) {
		r.o[0] = origin.x; r.o[1] = origin.y; r.o[2] = origin.z; r.d[0] = dir.x; r.d[1] = dir.y; r.d[2] = dir.z;
		for (int d = 0; d < 3; d++) r.inv[d] = 1.0f / r.d[d];
		r.tmax = tmax;
		r.octant = (r.d[0] < 0.0f ? 0 : 4) | (r.d[1] < 0.0f ? 0 : 2) | (r.d[2] < 0.0f ? 0 : 1);
	}
	static Vector3 cosine_direction(Random & rng, const Vector3 & normal) {
		const float r1 = rng.next(), r2 = rng.next(), radius = sqrtf(r1), phi = 6.2831853f * r2;
		const Vector3 helper = fabsf(normal.x) < 0.9f ? Vector3(1.0f, 0.0f, 0.0f) : Vector3(0.0f, 1.0f, 0.0f);
		const Vector3 tangent = Vector3::normalize(Vector3::cross(helper, normal)), bitangent = Vector3::cross(normal, tangent);
		return tangent * (radius * cosf(phi)) + bitangent * (radius * sinf(phi)) + normal * sqrtf(std::max(0.0f, 1.0f - r1));
	}

	// A ray from a point of the scene's free space in a uniformly random direction: what a camera standing THERE would send. The point: somewhere along a
	// cosine bounce off the surface, before its hit.
	void learn_from_free_point(uint64_t index) {
		Random rng(mix(index * 0xD1342543DE82EF95ull + 4242ull));
		Vector3 from, normal; surface_point(rng, from, normal);
		if (rng.next() < 0.5f) normal = normal * -1.0f;
		SampleRay probe; set_ray(probe, from + normal * (1.0e-4f * scene_size), cosine_direction(rng, normal), INFINITY);
		if (!std::isfinite(probe.o[0] + probe.o[1] + probe.o[2]) || !std::isfinite(probe.d[0] + probe.d[1] + probe.d[2])) return;
		float reach; if (trace(probe, reach) < 0) reach = 0.25f * scene_size;
		const float along = reach * (0.1f + 0.8f * rng.next());
		const Vector3 eye = Vector3(probe.o[0], probe.o[1], probe.o[2]) + Vector3(probe.d[0], probe.d[1], probe.d[2]) * along;
		const float z = 1.0f - 2.0f * rng.next(), phi = 6.2831853f * rng.next(), rho = sqrtf(std::max(0.0f, 1.0f - z * z));
		SampleRay r; set_ray(r, eye, Vector3(rho * cosf(phi), rho * sinf(phi), z), INFINITY);
		float unused; (void)learn_from(r, unused);
	}

	// One path as the integrator is about to trace them (Pathtracer.cu:122-139, 557-773 in outline): a camera ray through a random pixel, then up to `bounces`
	// cosine-distributed bounces; every ray is a sample. (Shadow rays towards the emitting meshes were sampled too and taken out again: scored like closest-hit rays
	// they made the real ones' walks LONGER, 12.9 -> 13.6 node steps -- an any-hit ray does not care for the nearest occluder; the surface-to-surface segments of
	// make_ray serve them better. Scoring segments as what they are -- the child that finds ANY occluder soonest first -- was tried as well: shadow rays 12.9 -> 12.0
	// node steps, closest-hit rays 13.1 -> 13.3, the launch as long as before; not kept. profiles/r05_slot_assignment.txt.) Returns how many rays it traced.
	int learn_from_path(uint64_t index, int bounces) {
		Random rng(mix(index * 0x9E3779B1ull + 777ull));
		SampleRay r;
		const float px = rng.next() * float(view->width), py = rng.next() * float(view->height);
		Vector3 dir = Vector3::normalize(view->bottom_left_corner + view->x_axis * px + view->y_axis * py), origin = view->position;
		if (!std::isfinite(dir.x) || !std::isfinite(dir.y) || !std::isfinite(dir.z) || !std::isfinite(origin.x + origin.y + origin.z)) return 0;   // (a ray of NaNs enters every box)
		set_ray(r, origin, dir, INFINITY);
		int traced = 0;
		for (int bounce = 0; bounce <= bounces; bounce++) {
			float t_hit; traced++;
			const int hit = learn_from(r, t_hit);
			if (hit < 0 || bounce == bounces) break;
			const Triangle & tri = triangles[size_t(bvh.indices[size_t(hit)])];
			Vector3 normal = Vector3::cross(tri.position_1 - tri.position_0, tri.position_2 - tri.position_0);
			const float length = Vector3::length(normal);
			if (!(length > 0.0f)) break;
			normal = normal * (1.0f / length);
			if (Vector3::dot(normal, dir) > 0.0f) normal = normal * -1.0f;
			const Vector3 point = origin + dir * t_hit + normal * (1.0e-4f * scene_size);
			origin = point; dir = cosine_direction(rng, normal);
			set_ray(r, origin, dir, INFINITY);
		}
		return traced;
	}
};

}   // namespace

void bvh8_learn_slot_order(BVH8 & bvh, const std::vector<Triangle> & triangles, int rays, int thread_count, const SlotLearningView * view) {
	if (bvh.nodes.empty() || triangles.empty() || bvh.indices.empty() || rays <= 0) return;
	rays = std::min(rays, 8 << 20);   // (the counters are 32 bits wide, a detour weighs at most 511 (learn_from): 8 M x 511 < 2^32)
	Learner learner(bvh, triangles);
	if (view && view->width > 0 && view->height > 0) learner.view = view;
	if (const char * c = getenv("GRT_SLOT_LEARNING_UNWEIGHTED")) learner.weigh_by_work = atoi(c) == 0;
	learner.index_leaves();
	// 2 KB of counters per scored node, at most 1 GB of them: a tree of more than half a million nodes is scored down to the depth that fits (the levels every ray walks)
	{
		const size_t row_limit = (size_t(1) << 30) / 2048;
		std::vector<int> depth(bvh.nodes.size(), 0); std::vector<size_t> at_depth(64, 0);
		std::vector<unsigned> order; order.reserve(bvh.nodes.size()); order.push_back(0u);
		for (size_t k = 0; k < order.size(); k++) {   // (parents before children)
			const BVHNode8 & n = bvh.nodes[order[k]];
			at_depth[size_t(std::min(depth[order[k]], 63))]++;
			for (int c = 0; c < __builtin_popcount(unsigned(n.imask)); c++) { const unsigned child = n.base_index_child + unsigned(c); if (child < bvh.nodes.size()) { depth[child] = depth[order[k]] + 1; order.push_back(child); } }
		}
		int deepest = 0; size_t rows = 0;
		while (deepest < 64 && rows + at_depth[size_t(deepest)] <= row_limit) rows += at_depth[size_t(deepest++)];
		learner.score_row.assign(bvh.nodes.size(), -1);
		int next_row = 0;
		for (unsigned node : order) if (depth[node] < deepest) learner.score_row[node] = next_row++;
		learner.score.reset(new std::atomic<unsigned>[size_t(next_row) * 512 + 1]);
		for (size_t i = 0; i < size_t(next_row) * 512; i++) learner.score[i].store(0u, std::memory_order_relaxed);
	}
	learner.area_cdf.resize(triangles.size());
	Vector3 lo(+INFINITY), hi(-INFINITY); double running = 0.0;
	for (size_t i = 0; i < triangles.size(); i++) {
		const Triangle & t = triangles[i];
		const double area = 0.5 * double(Vector3::length(Vector3::cross(t.position_1 - t.position_0, t.position_2 - t.position_0)));
		if (std::isfinite(area)) running += area;   // (a triangle with non-finite vertices is never picked)
		learner.area_cdf[i] = float(running);
		if (std::isfinite(area)) for (const Vector3 * p : { &t.position_0, &t.position_1, &t.position_2 }) { lo = Vector3::min(lo, *p); hi = Vector3::max(hi, *p); }
	}
	if (!(learner.area_cdf.back() > 0.0f)) return;
	learner.scene_size = Vector3::length(hi - lo);
	if (!std::isfinite(learner.scene_size) || !(learner.scene_size > 0.0f)) return;

	if (thread_count <= 0) thread_count = int(std::max(1u, std::thread::hardware_concurrency()));
	thread_count = std::min(thread_count, 64);
	{
		std::atomic<int> next(0);
		// with a view: three quarters of the rays come as paths from the camera (up to 4 rays each), the rest from the surface itself
		// the budget: a quarter as paths from the caller's camera (if it has one), half from points of the free space, the rest from the surface itself. Measured
		// (profiles/r05_slot_assignment.txt, 4.): a quarter of camera paths already seats the tree for that camera as well as three quarters do (1.308 against 1.301 ms
		// per step), and the free-space rays are what other viewpoints gain from (the reference's nine: 1.286 against 1.291; no camera paths at all: 1.344 / 1.303)
		const float camera_share = learner.view ? 0.25f : 0.0f, free_share = 0.5f;
		const int paths = int(float(rays) * camera_share) / 4, free_rays = int(float(rays) * free_share) / 2, surface_rays = std::max(0, rays - paths * 4 - free_rays * 2), items = surface_rays + paths + free_rays;
		auto work = [&] { for (int begin; (begin = next.fetch_add(512)) < items; ) for (int i = begin; i < std::min(items, begin + 512); i++) {
			if (i < surface_rays) { SampleRay r; float unused; if (learner.make_ray(uint64_t(i), r)) (void)learner.learn_from(r, unused); }
			else if (i < surface_rays + paths) (void)learner.learn_from_path(uint64_t(i - surface_rays), 3);
			else learner.learn_from_free_point(uint64_t(i - surface_rays - paths));
		} };
		std::vector<std::thread> helpers;
		for (int t = 1; t < thread_count; t++) helpers.emplace_back(work);
		work();
		for (std::thread & t : helpers) t.join();
	}

	// per node: the seating of least score, by exchanges; then the re-ordered records
	std::vector<BVHNode8> seated(bvh.nodes.size());
	std::vector<unsigned> place(bvh.nodes.size());
	for (size_t i = 0; i < place.size(); i++) place[i] = unsigned(i);
	{
		std::atomic<size_t> next(0);
		auto work = [&] {
			for (size_t begin; (begin = next.fetch_add(256)) < bvh.nodes.size(); ) for (size_t node = begin; node < std::min(bvh.nodes.size(), begin + 256); node++) {
				const BVHNode8 & n = bvh.nodes[node];
				seated[node] = n;
				if (__builtin_popcount(unsigned(n.imask)) < 2 || learner.score_row[node] < 0) continue;
				const std::atomic<unsigned> * counts = &learner.score[size_t(learner.score_row[node]) * 512];
				unsigned table[8][8][8]; unsigned long long total = 0;
				for (int v = 0; v < 8; v++) for (int c = 0; c < 8; c++) for (int h = 0; h < 8; h++) { table[v][c][h] = counts[(v * 8 + c) * 8 + h].load(std::memory_order_relaxed); total += table[v][c][h]; }
				if (total < 16) continue;   // nothing to go by: the converter's seating stays
				// what seating child c in slot a and child h in slot b costs: the rays that end in h and enter c behind their hit, over the octants that walk a before b
				static thread_local unsigned long long pair_cost[8][8][8][8];
				int inner[8], inner_count = 0;
				for (int c = 0; c < 8; c++) if ((n.imask >> c) & 1) inner[inner_count++] = c;
				for (int i = 0; i < inner_count; i++) for (int j = 0; j < inner_count; j++) {
					const int c = inner[i], h = inner[j];
					for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) {
						unsigned long long sum = 0;
						if (c != h && a != b) for (int v = 0; v < 8; v++) if ((a ^ v) > (b ^ v)) sum += table[v][c][h];
						pair_cost[c][h][a][b] = sum;
					}
				}
				int seat[8]; for (int s = 0; s < 8; s++) seat[s] = s;   // old slot -> new slot
				auto cost = [&](const int * at) {
					unsigned long long sum = 0;
					for (int i = 0; i < inner_count; i++) for (int j = 0; j < inner_count; j++) sum += pair_cost[inner[i]][inner[j]][at[inner[i]]][at[inner[j]]];
					return sum;
				};
				// a first bound by exchanges from the converter's seating ...
				unsigned long long current = cost(seat);
				for (bool improved = true; improved && current > 0; ) {
					improved = false;
					int best_a = -1, best_b = -1; unsigned long long best = current;
					for (int a = 0; a < 8; a++) for (int b = a + 1; b < 8; b++) {
						if (!((n.imask >> a) & 1) && !((n.imask >> b) & 1)) continue;   // (two seats without an inner child: nothing changes)
						std::swap(seat[a], seat[b]);
						const unsigned long long v = cost(seat);
						std::swap(seat[a], seat[b]);
						if (v < best) { best = v; best_a = a; best_b = b; }
					}
					if (best_a >= 0) { std::swap(seat[best_a], seat[best_b]); current = best; improved = true; }
				}
				// ... then every seating of the inner children that can still beat it (depth first, a branch is left as soon as its pairs so far cost as much as the best;
				// at most 8! / (8 - k)! leaves, in practice a few thousand steps; bounded all the same)
				if (current > 0 && inner_count >= 2) {
					int chosen[8], best_chosen[8]; bool found = false; unsigned taken = 0; long steps = 0;
					unsigned long long best = current;
					auto place = [&](auto && self, int i, unsigned long long so_far) -> void {
						if (so_far >= best || steps > 400000) return;
						if (i == inner_count) { best = so_far; found = true; for (int k = 0; k < inner_count; k++) best_chosen[k] = chosen[k]; return; }
						for (int slot = 0; slot < 8; slot++) if (!((taken >> slot) & 1u)) {
							steps++;
							unsigned long long add = 0;
							for (int k = 0; k < i; k++) add += pair_cost[inner[i]][inner[k]][slot][chosen[k]] + pair_cost[inner[k]][inner[i]][chosen[k]][slot];
							chosen[i] = slot; taken |= 1u << slot;
							self(self, i + 1, so_far + add);
							taken &= ~(1u << slot);
						}
					};
					place(place, 0, 0ull);
					if (found) {
						unsigned used = 0;
						for (int k = 0; k < inner_count; k++) { seat[inner[k]] = best_chosen[k]; used |= 1u << best_chosen[k]; }
						int next_free = 0;
						for (int s = 0; s < 8; s++) if (!((n.imask >> s) & 1)) { while ((used >> next_free) & 1u) next_free++; seat[s] = next_free; used |= 1u << next_free; }
						current = best;
					}
				}
				bool moved = false; for (int s = 0; s < 8; s++) if (seat[s] != s) moved = true;
				if (!moved) continue;
				BVHNode8 out = n;
				out.imask = 0;
				for (int s = 0; s < 8; s++) {
					const int to = seat[s];
					out.quantized_min_x[to] = n.quantized_min_x[s]; out.quantized_max_x[to] = n.quantized_max_x[s];
					out.quantized_min_y[to] = n.quantized_min_y[s]; out.quantized_max_y[to] = n.quantized_max_y[s];
					out.quantized_min_z[to] = n.quantized_min_z[s]; out.quantized_max_z[to] = n.quantized_max_z[s];
					if ((n.imask >> s) & 1) { out.meta[to] = byte(0x20 | (24 + to)); out.imask |= byte(1u << to); }
					else out.meta[to] = n.meta[s];
				}
				int rank = 0;
				for (int s = 0; s < 8; s++) if ((n.imask >> s) & 1) {
					const int new_rank = __builtin_popcount(unsigned(out.imask) & ((1u << seat[s]) - 1u));
					place[n.base_index_child + unsigned(rank++)] = n.base_index_child + unsigned(new_rank);
				}
				seated[node] = out;
			}
		};
		std::vector<std::thread> helpers;
		for (int t = 1; t < thread_count; t++) helpers.emplace_back(work);
		work();
		for (std::thread & t : helpers) t.join();
	}
	std::vector<BVHNode8> moved(bvh.nodes.size());
	for (size_t node = 0; node < bvh.nodes.size(); node++) moved[place[node]] = seated[node];
	bvh.nodes.swap(moved);
}
