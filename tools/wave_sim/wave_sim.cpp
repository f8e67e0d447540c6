// wave_sim.cpp -- a CPU model of how kernels_trace.hip's persistent 64-lane waves spend their issue slots.
//
// Development tool (not product, not a checker): it replays the ROUND STRUCTURE of bvh8_trace_engine -- which lanes
// of a wave take part in the node step, the instance entry, the triangle batch, the pop, the refill of every round --
// over real rays of a real scene, and prices each block with the VALU instruction count read off the ISA. GPU minutes
// are scarce; this answers "what would schedule X do to lane utilisation and issue slots per ray" on the CPU first.
// The traversal arithmetic is restated here (self-contained: nothing under oracle/ is used).
//
//   g++ -O2 -std=c++17 -fopenmp -o /tmp/wave_sim tools/wave_sim/wave_sim.cpp
//   python tools/wave_sim/export_scene.py sponza && /tmp/wave_sim /tmp/wave_sim_sponza.bin
//   python tools/wave_sim/export_merged.py sponza /tmp/flat.bin 2 && /tmp/wave_sim /tmp/flat.bin      (one tree over everything: DESIGN.md 4.6)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

struct f3 { float x, y, z; };
static inline f3 mk3(float x, float y, float z) { return { x, y, z }; }
static inline f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline f3 operator+(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline f3 operator*(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
static inline f3 operator*(f3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static inline float dot_fma(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline f3 cross_fma(f3 a, f3 b) { return { fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)) }; }
static inline f3 normalize(f3 a) { float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); return { a.x / l, a.y / l, a.z / l }; }
static inline unsigned msb(unsigned x) { return 31u - unsigned(__builtin_clz(x)); }
static inline unsigned extract_byte(unsigned x, unsigned i) { return (x >> (i * 8)) & 0xffu; }
static inline unsigned sign_extend_s8x4(unsigned x) { return ((x >> 7) & 0x01010101u) * 0xffu; }
static inline float as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

struct Scene {
	std::vector<int> depth;   // per node (single-tree scenes): levels below the root
	int node_count, tri_count, mesh_count, tlas_count, width, height;
	std::vector<uint8_t> nodes; std::vector<float> tris; std::vector<int> roots; std::vector<float> xinv; float cam[15];
};
struct Ray { f3 o, d; float tmax; };

static unsigned octant_inv4(f3 d) { return (d.x < 0 ? 0u : 0x04040404u) | (d.y < 0 ? 0u : 0x02020202u) | (d.z < 0 ? 0u : 0x01010101u); }

static unsigned node_intersect(const Ray & ray, f3 inv_dir, unsigned oct_inv4, float max_distance, const uint8_t * node) {
	uint32_t w[20]; memcpy(w, node, 80);
	f3 p = mk3(as_float(w[0]), as_float(w[1]), as_float(w[2]));
	unsigned e_imask = w[3];
	f3 adi = mk3(as_float(extract_byte(e_imask, 0) << 23) * inv_dir.x, as_float(extract_byte(e_imask, 1) << 23) * inv_dir.y, as_float(extract_byte(e_imask, 2) << 23) * inv_dir.z);
	f3 ao = (p - ray.o) * inv_dir;
	bool nx = ray.d.x < 0, ny = ray.d.y < 0, nz = ray.d.z < 0;
	unsigned hit_mask = 0;
	for (int i = 0; i < 2; i++) {
		unsigned meta4 = w[6 + i];
		unsigned is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
		unsigned inner_mask4 = sign_extend_s8x4(is_inner4 << 3);
		unsigned bit_index4 = (meta4 ^ (oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
		unsigned child_bits4 = (meta4 >> 5) & 0x07070707u;
		unsigned qlx = w[8 + i], qhx = w[10 + i], qly = w[12 + i], qhy = w[14 + i], qlz = w[16 + i], qhz = w[18 + i];
		unsigned xmin = nx ? qhx : qlx, xmax = nx ? qlx : qhx, ymin = ny ? qhy : qly, ymax = ny ? qly : qhy, zmin = nz ? qhz : qlz, zmax = nz ? qlz : qhz;
		for (int j = 0; j < 4; j++) {
			float tx0 = fmaf(float(extract_byte(xmin, j)), adi.x, ao.x), ty0 = fmaf(float(extract_byte(ymin, j)), adi.y, ao.y), tz0 = fmaf(float(extract_byte(zmin, j)), adi.z, ao.z);
			float tx1 = fmaf(float(extract_byte(xmax, j)), adi.x, ao.x), ty1 = fmaf(float(extract_byte(ymax, j)), adi.y, ao.y), tz1 = fmaf(float(extract_byte(zmax, j)), adi.z, ao.z);
			float tmin = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, 0.0f)), tmax = fminf(fminf(tx1, ty1), fminf(tz1, max_distance));
			if (tmin < tmax) hit_mask |= extract_byte(child_bits4, j) << extract_byte(bit_index4, j);
		}
	}
	return hit_mask;
}

struct Hit { float t = INFINITY; int tri = -1; };
// returns true for a shadow ray that is occluded
static bool triangle_test(const Scene & s, bool shadow, int id, const Ray & ray, float max_distance, Hit & hit) {
	const float * tr = &s.tris[size_t(id) * 9];
	f3 p0 = mk3(tr[0], tr[1], tr[2]), e1 = mk3(tr[3], tr[4], tr[5]), e2 = mk3(tr[6], tr[7], tr[8]);
	f3 h = cross_fma(ray.d, e2); float a = dot_fma(e1, h), f = 1.0f / a; f3 sv = ray.o - p0; float u = f * dot_fma(sv, h);
	if (u >= 0 && u <= 1) { f3 q = cross_fma(sv, e1); float v = f * dot_fma(ray.d, q);
		if (v >= 0 && u + v <= 1) { float t = f * dot_fma(e2, q);
			if (shadow) { if (t > 0 && t < max_distance) return true; } else if (t > 0 && t < hit.t) { hit.t = t; hit.tri = id; } } }
	return false;
}

// ---- the schedule under test ----------------------------------------------------------------------------------
struct Policy {
	const char * name = "current";
	int tri_batch = 2;          // triangles a lane tests per round
	bool early_instance = false; // enter the instance BEFORE the node step of a round; popped TLAS leaf groups become pending at once
	int n_d = 4, n_w = 16;      // dynamic fetch thresholds
	int ray_block = 128;
	int tri_hold = 0;           // postponing: run the triangle phase only with >= this many lanes wanting it (or nobody can step)
	bool pop_in_node_phase = false; // a lane whose round ends with nothing pending pops and may take its node step in the SAME... (unused)
	int second_min = 0;         // the k-th (k >= 1) test of a batch runs only when at least this many lanes have a k-th triangle; else those triangles wait for the next round
	bool overlap = false;       // a lane takes its node step while it still holds <= tri_batch pending triangles (they are tested in the same round, after the step); single-tree scenes only
	bool coop = false;          // cooperative triangle phase: every pending (ray, triangle) pair of the wave gets a lane of its own (hand-over through LDS), all pending triangles of a lane in one round
	double coop_owner = 43, coop_helper = 58, coop_readback = 8;   // instruction prices: prefix + slot writes (all running lanes), fetch + test + result write per pass of 64 pairs, read-back per triangle trip
};

// instruction prices per block (VALU wave-instructions, kernel_trace_stream_bvh8 wide closest engine, round 2 ISA)
struct Cost { double header = 17 + 2, push = 3, node = 220, leafgroup = 4, inst_head = 14, inst_push = 4, inst_xform = 88, tri_head = 18, tri_test = 52.5, end = 49, pop = 12, leave_reload = 45, refill = 170; };

struct LaneState {
	bool has_ray = false; int ray_index = -1; bool shadow = false;
	Ray ray, world; f3 inv; unsigned oct = 0; float max_distance = 0; Hit hit;
	uint32_t cg_x = 0, cg_y = 0, tg_x = 0, tg_y = 0, ng_x = 0, ng_y = 0; bool stepped = false; int tlas_stack = -1; int mesh = 0; bool ident = true;
	std::vector<std::pair<uint32_t, uint32_t>> stack;
	bool running = false;
};

struct Stats {
	double rounds = 0, instr = 0, useful = 0; // wave-instructions issued, lane-instructions useful / 64
	double node_exec = 0, node_lanes = 0, tri_exec[8] = {}, tri_lanes[8] = {}, inst_exec = 0, inst_lanes = 0, end_pop_lanes = 0, refills = 0, refill_lanes = 0;
	double nodes = 0, tris = 0, rays = 0, insts = 0, leafpops = 0;
	double depth_visits[24] = {};   // node steps by depth of the node below the root of a single flattened tree (what an LDS-resident top of the tree would serve)
	void add(const Stats & o) { const double * a = &o.rounds; double * b = &rounds; for (size_t i = 0; i < sizeof(Stats) / sizeof(double); i++) b[i] += a[i]; }
};

static void run_wave(const Scene & s, const std::vector<Ray> & rays, bool shadow_rays, size_t & cursor, const Policy & pol, const Cost & c, Stats & st, std::vector<Hit> * hits_out) {
	LaneState L[64];
	size_t blk_next = 0, blk_end = 0; bool drained = false;
	auto fetch = [&]() -> long { if (blk_next >= blk_end) { if (cursor >= rays.size()) { drained = true; return -1; } blk_next = cursor; blk_end = std::min(rays.size(), cursor + size_t(pol.ray_block)); cursor = blk_end; } return long(blk_next++); };
	while (true) {
		// refill
		int want = 0, got = 0;
		for (auto & l : L) if (!l.has_ray) want++;
		if (want) {
			for (auto & l : L) if (!l.has_ray && !drained) { long i = fetch(); if (i < 0) break; got++;
				l.has_ray = true; l.ray_index = int(i); l.shadow = shadow_rays; l.world = rays[i]; l.ray = l.world; l.max_distance = shadow_rays ? l.world.tmax : INFINITY;
				l.inv = mk3(1.0f / l.ray.d.x, 1.0f / l.ray.d.y, 1.0f / l.ray.d.z); l.oct = octant_inv4(l.ray.d); l.cg_x = 0; l.cg_y = 0x80000000u; l.tg_x = l.tg_y = 0; l.hit = Hit(); l.tlas_stack = s.tlas_count == 0 ? -2 : -1; /* no top level: the ray starts inside the one bottom-level tree */ l.stack.clear(); }
			st.refills++; st.refill_lanes += got; st.instr += c.refill; st.useful += c.refill * got / 64.0;
		}
		int alive = 0; for (auto & l : L) if (l.has_ray) alive++;
		if (!alive) return;
		for (auto & l : L) l.running = l.has_ray;
		int lost = 0;
		do {
			st.rounds++; st.instr += c.header; int running = 0; for (auto & l : L) if (l.running) running++; st.useful += c.header * running / 64.0;
			auto enter_instances = [&]() {
				int n = 0, npush = 0, nx = 0;
				for (auto & l : L) if (l.running && l.tg_y != 0 && l.tlas_stack == -1) { n++;
					int off = int(msb(l.tg_y)); l.tg_y &= ~(1u << off); l.mesh = int(l.tg_x) + off;
					if (l.tg_y) { l.stack.push_back({ l.tg_x, l.tg_y }); npush++; }
					if (l.cg_y & 0xff000000u) { l.stack.push_back({ l.cg_x, l.cg_y }); npush++; }
					l.tlas_stack = int(l.stack.size()); l.tg_y = 0;
					unsigned root = unsigned(s.roots[l.mesh]); l.ident = (root >> 31) != 0;
					if (!l.ident) { nx++; const float * m = &s.xinv[size_t(l.mesh) * 12];
						f3 o = l.ray.o, d = l.ray.d;
						l.ray.o = mk3(fmaf(m[0], o.x, fmaf(m[1], o.y, fmaf(m[2], o.z, m[3]))), fmaf(m[4], o.x, fmaf(m[5], o.y, fmaf(m[6], o.z, m[7]))), fmaf(m[8], o.x, fmaf(m[9], o.y, fmaf(m[10], o.z, m[11]))));
						l.ray.d = mk3(fmaf(m[0], d.x, fmaf(m[1], d.y, m[2] * d.z)), fmaf(m[4], d.x, fmaf(m[5], d.y, m[6] * d.z)), fmaf(m[8], d.x, fmaf(m[9], d.y, m[10] * d.z)));
						l.inv = mk3(1.0f / l.ray.d.x, 1.0f / l.ray.d.y, 1.0f / l.ray.d.z); l.oct = octant_inv4(l.ray.d); }
					l.cg_x = root & 0x7fffffffu; l.cg_y = 0x80000000u; st.insts++; }
				if (n) { st.inst_exec++; st.inst_lanes += n; st.instr += c.inst_head; st.useful += c.inst_head * n / 64.0; }
				if (npush) { st.instr += c.inst_push; st.useful += c.inst_push * npush / 64.0; }
				if (nx) { st.instr += c.inst_xform; st.useful += c.inst_xform * nx / 64.0; }
			};
			if (pol.early_instance) enter_instances();
			// node phase
			{ int n = 0, npush = 0, nleaf = 0;
				for (auto & l : L) l.stepped = false;
				for (auto & l : L) if (l.running && (pol.overlap ? __builtin_popcount(l.tg_y) <= pol.tri_batch : l.tg_y == 0)) {
					if (l.cg_y & 0xff000000u) { n++;
						unsigned hits_imask = l.cg_y, off = msb(hits_imask), base = l.cg_x; l.cg_y &= ~(1u << off);
						if (l.cg_y & 0xff000000u) { l.stack.push_back({ l.cg_x, l.cg_y }); npush++; }
						unsigned slot = (off - 24) ^ (l.oct & 0xffu), rel = __builtin_popcount(hits_imask & ~(0xffffffffu << slot));
						const uint8_t * node = &s.nodes[size_t(base + rel) * 80];
						if (!s.depth.empty()) st.depth_visits[std::min(23, s.depth[size_t(base + rel)])]++;
						unsigned hm = node_intersect(l.ray, l.inv, l.oct, l.shadow ? l.max_distance : l.hit.t, node);
						uint32_t w[8]; memcpy(w, node, 32);
						l.cg_x = w[4]; l.cg_y = (hm & 0xff000000u) | (w[3] >> 24); st.nodes++;
						if (pol.overlap) { l.ng_x = w[5]; l.ng_y = hm & 0x00ffffffu; l.stepped = true; } else { l.tg_x = w[5]; l.tg_y = hm & 0x00ffffffu; }
					} else if (!pol.early_instance) { if (l.cg_y) { nleaf++; l.tg_x = l.cg_x; l.tg_y = l.cg_y; l.cg_x = l.cg_y = 0; } }
				}
				if (n) { st.node_exec++; st.node_lanes += n; st.instr += c.node; st.useful += c.node * n / 64.0; }
				if (npush) { st.instr += c.push; st.useful += c.push * npush / 64.0; }
				if (nleaf) { st.instr += c.leafgroup; st.useful += c.leafgroup * nleaf / 64.0; st.leafpops += nleaf; }
			}
			if (!pol.early_instance) enter_instances();
			// triangle phase
			{ int wanting = 0, stepping = 0; for (auto & l : L) if (l.running) { if (l.tg_y != 0 && l.tlas_stack != -1) wanting++; else if (l.tg_y == 0 && (l.cg_y & 0xff000000u)) stepping++; }
				bool go = wanting > 0 && (wanting >= pol.tri_hold || stepping == 0);
				if (go && pol.coop) {
					int pairs = 0, max_n = 0;
					for (auto & l : L) if (l.running && l.tg_y != 0 && l.tlas_stack != -1) { bool occluded = false; int n = 0;
						while (l.tg_y != 0) { int ti = int(msb(l.tg_y)); l.tg_y &= ~(1u << ti); n++; pairs++;
							if (!occluded) { st.tris++; if (triangle_test(s, l.shadow, int(l.tg_x) + ti, l.ray, l.max_distance, l.hit)) occluded = true; } }
						max_n = std::max(max_n, n);
						if (occluded) { l.stack.clear(); l.cg_y = 0; l.tg_y = 0; l.running = false; l.hit.tri = -2; } }
					int running = 0; for (auto & l : L) if (l.running) running++;
					int passes = (pairs + 63) / 64;
					st.instr += pol.coop_owner + passes * pol.coop_helper + pol.coop_readback * max_n;
					st.useful += pol.coop_owner * wanting / 64.0 + pol.coop_helper * pairs / 64.0 + pol.coop_readback * pairs / 64.0;
					st.tri_exec[0] += passes; st.tri_lanes[0] += pairs;
				} else
				if (go) { st.instr += c.tri_head; st.useful += c.tri_head * wanting / 64.0;
					int per_k[8] = {};
					int want_k[8] = {};
					if (pol.second_min > 0) for (auto & l : L) if (l.running && l.tg_y != 0 && l.tlas_stack != -1) { int n = __builtin_popcount(l.tg_y); for (int k = 0; k < pol.tri_batch && k < n; k++) want_k[k]++; }
					for (auto & l : L) if (l.running && l.tg_y != 0 && l.tlas_stack != -1) { bool occluded = false;
						for (int k = 0; k < pol.tri_batch && l.tg_y != 0; k++) { if (k >= 1 && pol.second_min > 0 && want_k[k] < pol.second_min) break; int ti = int(msb(l.tg_y)); l.tg_y &= ~(1u << ti);
							if (!occluded) { per_k[k]++; st.tris++; if (triangle_test(s, l.shadow, int(l.tg_x) + ti, l.ray, l.max_distance, l.hit)) occluded = true; } }
						if (occluded) { l.stack.clear(); l.cg_y = 0; l.tg_y = 0; l.running = false; l.hit.tri = -2; } }
					for (int k = 0; k < pol.tri_batch; k++) if (per_k[k]) { st.tri_exec[k]++; st.tri_lanes[k] += per_k[k]; st.instr += c.tri_test; st.useful += c.tri_test * per_k[k] / 64.0; }
				}
			}
			if (pol.overlap) for (auto & l : L) if (l.stepped) { if (l.running) { l.tg_x = l.ng_x; l.tg_y = l.ng_y; } l.stepped = false; }   // (the pending ones have just been tested: at most tri_batch of them)
			// end of round: finish / leave instance / pop
			{ int npop = 0, nreload = 0; int still = 0; for (auto & l : L) if (l.running) still++;
				st.instr += c.end; st.useful += c.end * still / 64.0;
				for (auto & l : L) if (l.running && (pol.overlap ? __builtin_popcount(l.tg_y) <= pol.tri_batch : l.tg_y == 0) && (l.cg_y & 0xff000000u) == 0) {
					if (l.stack.empty()) { if (l.tg_y == 0) { l.cg_y = 0; l.running = false; } }
					else { if (int(l.stack.size()) == l.tlas_stack) { l.tlas_stack = -1; if (!l.ident) { nreload++; l.ray = l.world; l.inv = mk3(1.0f / l.ray.d.x, 1.0f / l.ray.d.y, 1.0f / l.ray.d.z); l.oct = octant_inv4(l.ray.d); } }
						auto e = l.stack.back(); l.stack.pop_back(); npop++; l.cg_x = e.first; l.cg_y = e.second;
						if (pol.early_instance && (l.cg_y & 0xff000000u) == 0) { l.tg_x = l.cg_x; l.tg_y = l.cg_y; l.cg_x = l.cg_y = 0; st.leafpops++; } }
				}
				if (npop) { st.instr += c.pop; st.useful += c.pop * npop / 64.0; st.end_pop_lanes += npop; }
				if (nreload) { st.instr += c.leave_reload; st.useful += c.leave_reload * nreload / 64.0; }
			}
			int running_now = 0; for (auto & l : L) if (l.running) running_now++;
			lost += 64 - running_now - pol.n_d;
		} while (lost < pol.n_w);
		for (auto & l : L) if (l.has_ray && !l.running) { if (hits_out) (*hits_out)[l.ray_index] = l.hit; l.has_ray = false; st.rays++; }
	}
}

static Stats simulate(const Scene & s, const std::vector<Ray> & rays, bool shadow, const Policy & pol, std::vector<Hit> * hits_out = nullptr) {
	// waves are independent given the block they claim; model the shared cursor by giving each simulated wave a contiguous
	// super-chunk (what the persistent grid does in effect over a launch: ~6100 waves x blocks of 128 round robin).
	const int n_waves = std::max(1, int(rays.size() / 4096)); Cost cost; Stats total;   // ~4000 rays per wave, as in a 25 M-ray launch on 6144 resident waves
	// round-robin blocks: wave w gets blocks w, w + n_waves, ... -> build per-wave ray lists
	std::vector<Stats> per(n_waves);
	#pragma omp parallel for schedule(dynamic, 16)
	for (int w = 0; w < n_waves; w++) {
		std::vector<Ray> mine; std::vector<int> idx;
		for (size_t b = size_t(w) * pol.ray_block; b < rays.size(); b += size_t(n_waves) * pol.ray_block)
			for (size_t i = b; i < std::min(rays.size(), b + size_t(pol.ray_block)); i++) { mine.push_back(rays[i]); idx.push_back(int(i)); }
		if (mine.empty()) continue;
		size_t cursor = 0; std::vector<Hit> local(mine.size());
		run_wave(s, mine, shadow, cursor, pol, cost, per[w], &local);
		if (hits_out) for (size_t i = 0; i < idx.size(); i++) (*hits_out)[idx[i]] = local[i];
	}
	for (auto & p : per) total.add(p);
	return total;
}

static void report(const char * label, const Policy & pol, const Stats & st) {
	printf("%-34s %-22s rays %8.0f | rounds/ray %5.2f nodes %5.2f tris %5.2f inst %4.2f leafpop %4.2f | wave-instr/ray %6.1f util %.3f | node exec/round %.2f lanes %.1f | tri", label, pol.name, st.rays,
		st.rounds * 64 / st.rays / 1.0 / 64 * 64 / 64 * 64 / 64, st.nodes / st.rays, st.tris / st.rays, st.insts / st.rays, st.leafpops / st.rays, st.instr / st.rays, st.useful / st.instr, st.node_exec / st.rounds, st.node_lanes / std::max(1.0, st.node_exec));
	for (int k = 0; k < pol.tri_batch && k < 4; k++) printf(" [%d] %.2f x %.1f", k, st.tri_exec[k] / st.rounds, st.tri_lanes[k] / std::max(1.0, st.tri_exec[k]));
	if (st.depth_visits[0] > 0) { printf(" | node steps per ray by depth:"); for (int d = 0; d < 12; d++) printf(" %.2f", st.depth_visits[d] / st.rays); }
	printf(" | refill every %.1f rounds x %.1f lanes\n", st.rounds / std::max(1.0, st.refills), st.refill_lanes / std::max(1.0, st.refills));
}

int main(int argc, char ** argv) {
	if (argc < 2) { fprintf(stderr, "usage: wave_sim scene.bin [stride]\n"); return 1; }
	Scene s; FILE * f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 1; }
	int hdr[6]; if (fread(hdr, 4, 6, f) != 6) return 1; s.node_count = hdr[0]; s.tri_count = hdr[1]; s.mesh_count = hdr[2]; s.tlas_count = hdr[3]; s.width = hdr[4]; s.height = hdr[5];
	s.nodes.resize(size_t(s.node_count) * 80); s.tris.resize(size_t(s.tri_count) * 9); s.roots.resize(s.mesh_count); s.xinv.resize(size_t(s.mesh_count) * 12);
	if (fread(s.nodes.data(), 1, s.nodes.size(), f) != s.nodes.size() || fread(s.tris.data(), 4, s.tris.size(), f) != s.tris.size() || fread(s.roots.data(), 4, s.roots.size(), f) != s.roots.size() || fread(s.xinv.data(), 4, s.xinv.size(), f) != s.xinv.size() || fread(s.cam, 4, 15, f) != 15) return 1;
	fclose(f);
	if (s.tlas_count == 0) { // one tree: depth of every node (children of a node are consecutive from its child base, one per bit of its inner mask)
		s.depth.assign(size_t(s.node_count), 0);
		std::vector<int> queue = { 0 };
		for (size_t q = 0; q < queue.size(); q++) {
			uint32_t w[5]; memcpy(w, &s.nodes[size_t(queue[q]) * 80], 20);
			int children = __builtin_popcount(w[3] >> 24);
			for (int c = 0; c < children; c++) { s.depth[size_t(w[4]) + c] = s.depth[size_t(queue[q])] + 1; queue.push_back(int(w[4]) + c); }
		}
		std::vector<int> per_level(24, 0); for (int d : s.depth) per_level[std::min(23, d)]++;
		printf("nodes per level:"); for (int d = 0; d < 12; d++) printf(" %d", per_level[d]); printf("\n");
	}
	int stride = argc > 2 ? atoi(argv[2]) : 2;   // every stride-th pixel row/column block keeps wave coherence: we subsample whole 64-pixel runs
	// primary rays (pinhole, pixel centres), in scan order, subsampled by whole rows
	std::vector<Ray> primary;
	f3 pos = mk3(s.cam[0], s.cam[1], s.cam[2]), blc = mk3(s.cam[3], s.cam[4], s.cam[5]), xa = mk3(s.cam[6], s.cam[7], s.cam[8]), ya = mk3(s.cam[9], s.cam[10], s.cam[11]);
	for (int y = 0; y < s.height; y += stride) for (int x = 0; x < s.width; x++) primary.push_back({ pos, normalize(blc + xa * (x + 0.5f) + ya * (y + 0.5f)), INFINITY });
	Policy cur; std::vector<Hit> hits(primary.size());
	Stats sp = simulate(s, primary, false, cur, &hits);
	report("primary", cur, sp);
	// bounce rays: cosine-free uniform directions from the hit points (as tools/trace_bench.py), in queue (= pixel) order
	std::mt19937 rng(1); std::normal_distribution<float> nd; std::vector<Ray> bounce, shadow;
	f3 light = mk3(0.0f, 12.0f, 0.0f);
	for (size_t i = 0; i < primary.size(); i++) if (hits[i].tri >= 0) { f3 p = primary[i].o + primary[i].d * (hits[i].t * 0.999f); f3 d = normalize(mk3(nd(rng), nd(rng), nd(rng))); bounce.push_back({ p, d, INFINITY });
		f3 to = light - p; float dist = sqrtf(dot_fma(to, to)); shadow.push_back({ p, to * (1.0f / dist), dist }); }
	std::vector<Hit> h2(bounce.size());
	Stats sb = simulate(s, bounce, false, cur, &h2); report("bounce (queue order)", cur, sb);
	// second-bounce rays: less coherent still
	std::vector<Ray> bounce2; for (size_t i = 0; i < bounce.size(); i++) if (h2[i].tri >= 0) { f3 p = bounce[i].o + bounce[i].d * (h2[i].t * 0.999f); bounce2.push_back({ p, normalize(mk3(nd(rng), nd(rng), nd(rng))), INFINITY }); }
	auto bucket = [&](const std::vector<Ray> & in, int block) { std::vector<Ray> out; out.reserve(in.size());
		for (size_t b = 0; b < in.size(); b += block) { size_t e = std::min(in.size(), b + size_t(block)); for (int o = 0; o < 8; o++) for (size_t i = b; i < e; i++) { const f3 & d = in[i].d; int oc = (d.x < 0 ? 4 : 0) | (d.y < 0 ? 2 : 0) | (d.z < 0 ? 1 : 0); if (oc == o) out.push_back(in[i]); } }
		return out; };
	std::vector<Policy> pols;
	{ Policy p; pols.push_back(p); }
	{ Policy p; p.name = "early_instance"; p.early_instance = true; pols.push_back(p); }
	{ Policy p; p.name = "tri_batch 1"; p.tri_batch = 1; pols.push_back(p); }
	{ Policy p; p.name = "tri_batch 3"; p.tri_batch = 3; pols.push_back(p); }
	{ Policy p; p.name = "early+batch3"; p.early_instance = true; p.tri_batch = 3; pols.push_back(p); }
	{ Policy p; p.name = "early+hold8"; p.early_instance = true; p.tri_hold = 8; pols.push_back(p); }
	{ Policy p; p.name = "early+hold16"; p.early_instance = true; p.tri_hold = 16; pols.push_back(p); }
	{ Policy p; p.name = "early nd2 nw8"; p.early_instance = true; p.n_d = 2; p.n_w = 8; pols.push_back(p); }
	{ Policy p; p.name = "early nd8 nw32"; p.early_instance = true; p.n_d = 8; p.n_w = 32; pols.push_back(p); }
	if (argc > 3) {   // the triangle-phase experiments of round 5 only
		std::vector<Policy> q;
		{ Policy p; q.push_back(p); }
		{ Policy p; p.name = "overlap (step with <= 2 pending)"; p.overlap = true; q.push_back(p); }
		{ Policy p; p.name = "overlap, batch 3"; p.overlap = true; p.tri_batch = 3; q.push_back(p); }
		if (argc > 4) { for (auto & p : q) { report("bounce2", p, simulate(s, bounce2, false, p)); report("shadow ", p, simulate(s, shadow, true, p)); report("primary", p, simulate(s, primary, false, p)); } return 0; }
		for (int m : { 4, 8, 12, 16, 24 }) { Policy p; p.name = "second slot if >= m lanes"; p.second_min = m; q.push_back(p); }
		for (int m : { 8, 16 }) { Policy p; p.name = "batch 3, slots 2,3 if >= m"; p.tri_batch = 3; p.second_min = m; q.push_back(p); }
		{ Policy p; p.name = "coop (43/58/8)"; p.coop = true; q.push_back(p); }
		{ Policy p; p.name = "coop cheap (25/55/6)"; p.coop = true; p.coop_owner = 25; p.coop_helper = 55; p.coop_readback = 6; q.push_back(p); }
		{ Policy p; p.name = "coop free (0/52/0)"; p.coop = true; p.coop_owner = 0; p.coop_helper = 52.5; p.coop_readback = 0; q.push_back(p); }
		for (auto & p : q) { report("bounce2", p, simulate(s, bounce2, false, p)); report("shadow ", p, simulate(s, shadow, true, p)); report("primary", p, simulate(s, primary, false, p)); }
		return 0;
	}
	for (auto & p : pols) { report("bounce2 (queue order)", p, simulate(s, bounce2, false, p)); }
	for (int block : { 64, 128, 256, 512, 1024 }) { auto r = bucket(bounce2, block); char l[64]; snprintf(l, 64, "bounce2 octant buckets of %d", block); report(l, pols[0], simulate(s, r, false, pols[0])); report(l, pols[1], simulate(s, r, false, pols[1])); }
	{ auto r = bucket(bounce2, 1 << 30); report("bounce2 octant-major (global)", pols[0], simulate(s, r, false, pols[0])); }
	// rays sorted by where they start (Morton code of the origin in a grid over the scene), within blocks of N rays and globally; with and without the octant as the top key
	{ f3 lo = mk3(1e30f, 1e30f, 1e30f), hi = mk3(-1e30f, -1e30f, -1e30f);
	  for (auto & r : bounce2) { lo.x = std::min(lo.x, r.o.x); lo.y = std::min(lo.y, r.o.y); lo.z = std::min(lo.z, r.o.z); hi.x = std::max(hi.x, r.o.x); hi.y = std::max(hi.y, r.o.y); hi.z = std::max(hi.z, r.o.z); }
	  auto expand = [](uint32_t v) { v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v; };
	  auto key = [&](const Ray & r, int bits, bool octant) { uint32_t q = (1u << bits) - 1; uint32_t x = uint32_t((r.o.x - lo.x) / (hi.x - lo.x) * q), y = uint32_t((r.o.y - lo.y) / (hi.y - lo.y) * q), z = uint32_t((r.o.z - lo.z) / (hi.z - lo.z) * q);
	      uint64_t m = (uint64_t(expand(x)) << 2) | (uint64_t(expand(y)) << 1) | expand(z); int oc = (r.d.x < 0 ? 4 : 0) | (r.d.y < 0 ? 2 : 0) | (r.d.z < 0 ? 1 : 0); return octant ? (uint64_t(oc) << 40) | m : (m << 3) | uint64_t(oc); };
	  for (int bits : { 4, 6, 10 }) for (int octant_first = 0; octant_first < 2; octant_first++) for (size_t block : { size_t(4096), size_t(65536), size_t(1) << 30 }) {
	      std::vector<Ray> r = bounce2;
	      for (size_t b = 0; b < r.size(); b += block) { size_t e = std::min(r.size(), b + block); std::stable_sort(r.begin() + b, r.begin() + e, [&](const Ray & a, const Ray & c) { return key(a, bits, octant_first) < key(c, bits, octant_first); }); }
	      char l[96]; snprintf(l, 96, "bounce2 morton%d %s blocks of %zu", bits, octant_first ? "octant-major" : "cell-major", block > (1u << 29) ? size_t(0) : block);
	      report(l, pols[1], simulate(s, r, false, pols[1])); } }
	for (auto & p : { pols[0], pols[1] }) report("shadow (point light, queue order)", p, simulate(s, shadow, true, p));
	for (auto & p : { pols[0], pols[1] }) report("primary", p, simulate(s, primary, false, p));
	return 0;
}
