// morton_sah.cpp -- CPU prototype for the device BLAS builder (kernels_blas.hip): how much of the gap between the linear BVH (cuts at the highest
// differing Morton bit) and a SAH tree closes when the SAME level-by-level 8-wide build over the Morton order chooses its cuts by surface-area cost?
// Development tool (not product, not a checker). Builds 8-wide trees over the triangles of a wave_sim scene file with several cut policies and
// counts node steps / triangle tests of closest-hit rays (primary + random bounce rays), with exact child boxes.
//   g++ -O2 -std=c++17 -fopenmp -o /tmp/morton_sah tools/blas_proto/morton_sah.cpp
//   python tools/wave_sim/export_merged.py sponza /tmp/merged.bin 0 && /tmp/morton_sah /tmp/merged.bin
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

struct V { float x, y, z; };
static inline V operator-(V a, V b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline V operator+(V a, V b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline V operator*(V a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static inline float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V cross(V a, V b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline V normalize(V a) { float l = sqrtf(dot(a, a)); return a * (1.0f / l); }
struct Box { float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
	void grow(const Box & b) { for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], b.lo[d]); hi[d] = std::max(hi[d], b.hi[d]); } }
	float area() const { float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2]; return x < 0 ? 0.0f : 2.0f * (x * y + y * z + z * x); } };

struct Tri { V p0, e1, e2; };
struct Node { Box child_box[8]; int child[8]; int count[8]; int n = 0; };   // child >= 0: inner node index; else leaf: first triangle = ~child, count[]
struct Tree { std::vector<Node> nodes; std::vector<int> order; double sah = 0; };

static std::vector<Tri> tris; static std::vector<Box> tbox; static float cam[15]; static int W, H;

static uint32_t expand(uint32_t v) { v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v; }

enum Policy { LBVH, SAH_EXACT, SAH_CANDIDATES, SAH_ALIGNED, RECORDED };
static std::vector<int> cut_depth;   // RECORDED: depth at which the boundary in front of position p was cut by the binary build (INT_MAX: never)
struct Build { const std::vector<uint64_t> & keys; const std::vector<Box> & sbox; Policy policy; bool area_first; int candidates; std::vector<Box> prefix, suffix; };

static Box range_box(const std::vector<Box> & sbox, int a, int b) { Box r; for (int i = a; i < b; i++) r.grow(sbox[i]); return r; }

static int lbvh_split(const std::vector<uint64_t> & keys, int lo, int hi) {
	uint64_t first = keys[lo], last = keys[hi - 1];
	if (first == last) return (lo + hi) / 2;
	uint64_t bit = 0x8000000000000000ull >> __builtin_clzll(first ^ last);
	int below = lo, above = hi - 1;
	while (above - below > 1) { int mid = (below + above) / 2; if (keys[mid] & bit) above = mid; else below = mid; }
	return above;
}
// best cut of [lo, hi) by SAH over the given candidate positions (exclusive ends)
static int sah_split(const Build & b, int lo, int hi, int rounds) {
	int n = hi - lo;
	std::vector<float> left(n + 1), right(n + 1);
	{ Box acc; for (int i = 0; i < n; i++) { acc.grow(b.sbox[lo + i]); left[i + 1] = acc.area(); } }
	{ Box acc; for (int i = n - 1; i >= 0; i--) { acc.grow(b.sbox[lo + i]); right[i] = acc.area(); } }
	auto cost = [&](int c) { return left[c] * c + right[c] * (n - c); };
	if (b.policy == SAH_EXACT || n <= b.candidates + 1) { int best = 1; float bc = cost(1); for (int c = 2; c < n; c++) { float v = cost(c); if (v < bc) { bc = v; best = c; } } return lo + best; }
	// K evenly spaced candidates, then refine around the best (what one wave would do: a lane per candidate, range boxes from a box pyramid)
	int a = 1, z = n - 1, best = n / 2; float bc = 1e38f;
	for (int r = 0; r < rounds && z - a >= 1; r++) {
		int K = b.candidates; double step = double(z - a) / (K + 1);
		int bi = best;
		for (int k = 1; k <= K; k++) { int c = a + int(step * k + 0.5); c = std::max(1, std::min(n - 1, c)); float v = cost(c); if (v < bc) { bc = v; bi = c; } }
		best = bi; int half = std::max(1, int(step + 1)); a = std::max(1, best - half); z = std::min(n - 1, best + half);
	}
	return lo + best;
}

// cuts at the cell boundaries of the next `levels` Morton bits below the common prefix of [lo, hi): the highest differing bit gives ONE cut (the
// linear BVH's), the bit below it up to two more, ...; the cheapest by SAH wins
static int aligned_split(const Build & b, int lo, int hi, int levels) {
	uint64_t first = b.keys[lo], last = b.keys[hi - 1];
	if (first == last) return (lo + hi) / 2;
	int top = 63 - __builtin_clzll(first ^ last);
	int n = hi - lo;
	std::vector<float> left(n + 1), right(n + 1);
	{ Box acc; for (int i = 0; i < n; i++) { acc.grow(b.sbox[lo + i]); left[i + 1] = acc.area(); } }
	{ Box acc; for (int i = n - 1; i >= 0; i--) { acc.grow(b.sbox[lo + i]); right[i] = acc.area(); } }
	int best = -1; float bc = 1e38f;
	uint64_t mask = ~((1ull << std::max(0, top - levels + 1)) - 1);   // the prefix down to the lowest considered bit
	for (int c = 1; c < n; c++) if ((b.keys[lo + c - 1] & mask) != (b.keys[lo + c] & mask)) { float v = left[c] * c + right[c] * (n - c); if (v < bc) { bc = v; best = c; } }
	return best < 0 ? (lo + hi) / 2 : lo + best;
}

static int recorded_split(int lo, int hi) { int best = (lo + hi) / 2, depth = 0x7fffffff; for (int p = lo + 1; p < hi; p++) if (cut_depth[p] < depth) { depth = cut_depth[p]; best = p; } return best; }

// top-down binned SAH (16 bins per axis over the centroid box, object splits only) down to <= 3 triangles: re-orders ids[lo, hi) and records the cuts
static void binned_sah(std::vector<int> & ids, const std::vector<Box> & boxes, int lo, int hi, int depth, int bins) {
	if (hi - lo <= 3) return;
	Box cb; for (int i = lo; i < hi; i++) { const Box & b = boxes[ids[i]]; Box c; for (int d = 0; d < 3; d++) c.lo[d] = c.hi[d] = 0.5f * (b.lo[d] + b.hi[d]); cb.grow(c); }
	int best_axis = -1, best_plane = 0; float best_cost = 1e38f;
	for (int axis = 0; axis < 3; axis++) {
		float extent = cb.hi[axis] - cb.lo[axis]; if (!(extent > 0)) continue;
		std::vector<Box> bb(bins); std::vector<int> bc(bins, 0);
		for (int i = lo; i < hi; i++) { const Box & b = boxes[ids[i]]; float c = 0.5f * (b.lo[axis] + b.hi[axis]); int k = std::min(bins - 1, int((c - cb.lo[axis]) / extent * bins)); bb[k].grow(b); bc[k]++; }
		std::vector<float> la(bins), ra(bins); std::vector<int> ln(bins), rn(bins);
		{ Box acc; int n = 0; for (int k = 0; k < bins; k++) { acc.grow(bb[k]); n += bc[k]; la[k] = n ? acc.area() : 0; ln[k] = n; } }
		{ Box acc; int n = 0; for (int k = bins - 1; k >= 0; k--) { acc.grow(bb[k]); n += bc[k]; ra[k] = n ? acc.area() : 0; rn[k] = n; } }
		for (int k = 0; k + 1 < bins; k++) if (ln[k] > 0 && rn[k + 1] > 0) { float cost = la[k] * ln[k] + ra[k + 1] * rn[k + 1]; if (cost < best_cost) { best_cost = cost; best_axis = axis; best_plane = k; } }
	}
	int cut;
	if (best_axis < 0) cut = (lo + hi) / 2;
	else { float extent = cb.hi[best_axis] - cb.lo[best_axis];
		auto left = [&](int id) { const Box & b = boxes[id]; float c = 0.5f * (b.lo[best_axis] + b.hi[best_axis]); return std::min(bins - 1, int((c - cb.lo[best_axis]) / extent * bins)) <= best_plane; };
		cut = int(std::stable_partition(ids.begin() + lo, ids.begin() + hi, left) - ids.begin()); }
	cut_depth[cut] = depth;
	binned_sah(ids, boxes, lo, cut, depth + 1, bins); binned_sah(ids, boxes, cut, hi, depth + 1, bins);
}

// Hybrid (round 5): top-down binned SAH (object splits, `bins` per axis) only down to pieces of <= K references or `max_depth` levels; a piece's path (0 = left,
// 1 = right, most significant first) becomes the high word of its references' keys, their Morton code inside the piece's box the low word: the level-by-level 8-wide
// build over the sorted keys then cuts at the highest differing bit as before -- which is the SAH tree's own cut as long as a run spans several pieces.
// extents: 0 = the bins span the centroid box (the classic), 1 = they span the node's box (what a device build knows without a pass of its own).
struct TopLeaf { uint32_t path; int depth; Box box; std::vector<int> ids; };
static void hybrid_top(std::vector<int> & ids, const std::vector<Box> & boxes, uint32_t path, int depth, int K, int max_depth, int bins, int extents, std::vector<TopLeaf> & out) {
	Box nb, cb; for (int id : ids) { const Box & b = boxes[id]; nb.grow(b); Box c; for (int d = 0; d < 3; d++) c.lo[d] = c.hi[d] = 0.5f * (b.lo[d] + b.hi[d]); cb.grow(c); }
	if (int(ids.size()) <= K || depth >= max_depth) { out.push_back({ path, depth, nb, ids }); return; }
	const Box & eb = extents ? nb : cb;
	int best_axis = -1, best_plane = 0; float best_cost = 1e38f;
	for (int axis = 0; axis < 3; axis++) {
		float extent = eb.hi[axis] - eb.lo[axis]; if (!(extent > 0)) continue;
		std::vector<Box> bb(bins); std::vector<int> bc(bins, 0);
		for (int id : ids) { const Box & b = boxes[id]; float c = 0.5f * (b.lo[axis] + b.hi[axis]); int k = std::max(0, std::min(bins - 1, int((c - eb.lo[axis]) / extent * bins))); bb[k].grow(b); bc[k]++; }
		std::vector<float> la(bins), ra(bins); std::vector<int> ln(bins), rn(bins);
		{ Box acc; int n = 0; for (int k = 0; k < bins; k++) { acc.grow(bb[k]); n += bc[k]; la[k] = n ? acc.area() : 0; ln[k] = n; } }
		{ Box acc; int n = 0; for (int k = bins - 1; k >= 0; k--) { acc.grow(bb[k]); n += bc[k]; ra[k] = n ? acc.area() : 0; rn[k] = n; } }
		for (int k = 0; k + 1 < bins; k++) if (ln[k] > 0 && rn[k + 1] > 0) { float cost = la[k] * ln[k] + ra[k + 1] * rn[k + 1]; if (cost < best_cost) { best_cost = cost; best_axis = axis; best_plane = k; } }
	}
	if (best_axis < 0) { out.push_back({ path, depth, nb, ids }); return; }   // every centre in one bin on every axis: the Morton order takes it from here
	float extent = eb.hi[best_axis] - eb.lo[best_axis];
	std::vector<int> l, r;
	for (int id : ids) { const Box & b = boxes[id]; float c = 0.5f * (b.lo[best_axis] + b.hi[best_axis]); int k = std::max(0, std::min(bins - 1, int((c - eb.lo[best_axis]) / extent * bins))); (k <= best_plane ? l : r).push_back(id); }
	std::vector<int>().swap(ids);
	hybrid_top(l, boxes, path << 1, depth + 1, K, max_depth, bins, extents, out); hybrid_top(r, boxes, (path << 1) | 1u, depth + 1, K, max_depth, bins, extents, out);
}

static int build_node(Tree & t, const Build & b, int lo, int hi) {
	int index = int(t.nodes.size()); t.nodes.emplace_back();
	int begin[9]; begin[0] = lo; for (int c = 1; c < 9; c++) begin[c] = hi; int count = 1;
	for (int round = 0; round < 7; round++) {
		int pick = -1; float key = -1.0f;
		for (int c = 0; c < count; c++) { int n = begin[c + 1] - begin[c]; if (n <= 3) continue;
			float area = range_box(b.sbox, begin[c], begin[c + 1]).area();
			float k = !b.area_first ? float(n) : b.candidates == -1 ? area : b.candidates == -2 ? area * sqrtf(float(n)) : b.candidates == -3 ? area * log2f(float(n)) : area * n;
			// -4 / -5 / -6: largest area first, but a SMALL piece (4 .. 6 / 9 / 12 triangles) -- which would otherwise become a whole node of its own for a handful of triangles -- is
			// cut first as long as the node has slots for its halves (round 5: node steps are what the device tree has too many of)
			if (b.candidates <= -4) { int small = b.candidates == -4 ? 6 : b.candidates == -5 ? 9 : 12; k = area; if (n <= small) k += 1e30f; }
			if (k > key) { key = k; pick = c; } }
		if (pick < 0) break;
		int cut = b.policy == RECORDED ? recorded_split(begin[pick], begin[pick + 1]) : b.policy == LBVH ? lbvh_split(b.keys, begin[pick], begin[pick + 1]) : b.policy == SAH_ALIGNED ? aligned_split(b, begin[pick], begin[pick + 1], b.candidates) : sah_split(b, begin[pick], begin[pick + 1], 3);
		for (int c = 8; c >= 1; c--) { if (c > pick + 1) begin[c] = begin[c - 1]; else if (c == pick + 1) begin[c] = cut; }
		count++;
	}
	Node node; node.n = count;
	for (int c = 0; c < count; c++) { node.child_box[c] = range_box(b.sbox, begin[c], begin[c + 1]); node.count[c] = begin[c + 1] - begin[c]; }
	for (int c = 0; c < count; c++) node.child[c] = node.count[c] <= 3 ? ~begin[c] : build_node(t, b, begin[c], begin[c + 1]);
	t.nodes[index] = node;
	return index;
}

struct Ray { V o, d; };
static bool slab(const Box & b, const Ray & r, V inv, float tmax, float & tnear) {
	float t0 = 0.0f, t1 = tmax;
	const float o[3] = { r.o.x, r.o.y, r.o.z }, id[3] = { inv.x, inv.y, inv.z };
	for (int d = 0; d < 3; d++) { float a = (b.lo[d] - o[d]) * id[d], c = (b.hi[d] - o[d]) * id[d]; if (a > c) std::swap(a, c); t0 = std::max(t0, a); t1 = std::min(t1, c); }
	tnear = t0; return t0 <= t1;
}
static void trace(const Tree & t, const Ray & r, float & best, int & hit, long & nodes, long & tests) {
	V inv = { 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };
	struct E { int node; float tn; }; E stack[256]; int sp = 0; stack[sp++] = { 0, 0.0f };
	while (sp) { E e = stack[--sp]; if (e.tn >= best) continue; const Node & n = t.nodes[e.node]; nodes++;
		int idx[8]; float tn[8]; int m = 0;
		for (int c = 0; c < n.n; c++) { float x; if (slab(n.child_box[c], r, inv, best, x)) { idx[m] = c; tn[m] = x; m++; } }
		for (int i = 1; i < m; i++) for (int j = i; j > 0 && tn[j] > tn[j - 1]; j--) { std::swap(tn[j], tn[j - 1]); std::swap(idx[j], idx[j - 1]); }   // far first on the stack
		for (int i = 0; i < m; i++) { int c = idx[i];
			if (n.child[c] >= 0) stack[sp++] = { n.child[c], tn[i] };
			else { int first = ~n.child[c]; for (int k = 0; k < n.count[c]; k++) { tests++; const Tri & tr = tris[t.order[first + k]];
				V h = cross(r.d, tr.e2); float a = dot(tr.e1, h), f = 1.0f / a; V s = r.o - tr.p0; float u = f * dot(s, h);
				if (u >= 0 && u <= 1) { V q = cross(s, tr.e1); float v = f * dot(r.d, q); if (v >= 0 && u + v <= 1) { float tt = f * dot(tr.e2, q); if (tt > 0 && tt < best) { best = tt; hit = t.order[first + k]; } } } } } }
	}
}

int main(int argc, char ** argv) {
	if (argc < 2) { fprintf(stderr, "usage: morton_sah scene.bin\n"); return 1; }
	FILE * f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 1; }
	int hdr[6]; if (fread(hdr, 4, 6, f) != 6) return 1; int node_count = hdr[0], tri_count = hdr[1], mesh_count = hdr[2]; W = hdr[4]; H = hdr[5];
	fseek(f, long(node_count) * 80, SEEK_CUR);
	std::vector<float> raw(size_t(tri_count) * 9); if (fread(raw.data(), 4, raw.size(), f) != raw.size()) return 1;
	fseek(f, long(mesh_count) * 4 + long(mesh_count) * 48, SEEK_CUR); if (fread(cam, 4, 15, f) != 15) return 1; fclose(f);
	tris.resize(tri_count); tbox.resize(tri_count); Box scene;
	for (int i = 0; i < tri_count; i++) { const float * r = &raw[size_t(i) * 9]; tris[i] = { { r[0], r[1], r[2] }, { r[3], r[4], r[5] }, { r[6], r[7], r[8] } };
		V v[3] = { tris[i].p0, tris[i].p0 + tris[i].e1, tris[i].p0 + tris[i].e2 }; Box b; for (auto & p : v) { Box q; q.lo[0] = q.hi[0] = p.x; q.lo[1] = q.hi[1] = p.y; q.lo[2] = q.hi[2] = p.z; b.grow(q); } tbox[i] = b; scene.grow(b); }
	// Early split clipping (round 5): a triangle whose box is longer than `presplit` x the scene's longest side along some axis is cut in two at the middle of
	// that axis, the pieces clipped (Sutherland-Hodgman) and boxed tightly, and so on; every piece becomes a REFERENCE with the triangle's data and its own box --
	// what the flattened tree's copies already are. Blind (no cost function), one thread per triangle on a device: the question is how close it brings the Morton
	// build to the host's SAH + spatial-split tree.
	const float presplit = argc > 5 ? float(atof(argv[5])) : 0.0f;
	if (presplit > 0.0f) {
		float longest = 0; for (int d = 0; d < 3; d++) longest = std::max(longest, scene.hi[d] - scene.lo[d]);
		const float limit = presplit * longest;
		std::vector<Tri> rt; std::vector<Box> rb;
		struct Piece { std::vector<V> poly; };
		for (int i = 0; i < tri_count; i++) {
			std::vector<Piece> work; work.push_back({ { tris[i].p0, tris[i].p0 + tris[i].e1, tris[i].p0 + tris[i].e2 } });
			int made = 0;
			while (!work.empty()) {
				Piece pc = work.back(); work.pop_back();
				Box b; for (auto & q : pc.poly) { Box c; c.lo[0] = c.hi[0] = q.x; c.lo[1] = c.hi[1] = q.y; c.lo[2] = c.hi[2] = q.z; b.grow(c); }
				int axis = 0; float ext = 0; for (int d = 0; d < 3; d++) if (b.hi[d] - b.lo[d] > ext) { ext = b.hi[d] - b.lo[d]; axis = d; }
				if (ext <= limit || made + int(work.size()) >= 63) { rt.push_back(tris[i]); rb.push_back(b); made++; continue; }
				float mid = 0.5f * (b.lo[axis] + b.hi[axis]);
				auto comp = [&](const V & v) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; };
				Piece lo_p, hi_p; size_t n = pc.poly.size();
				for (size_t k = 0; k < n; k++) { V a = pc.poly[k], c = pc.poly[(k + 1) % n]; float fa = comp(a) - mid, fc = comp(c) - mid;
					if (fa <= 0) lo_p.poly.push_back(a); if (fa >= 0) hi_p.poly.push_back(a);
					if ((fa < 0 && fc > 0) || (fa > 0 && fc < 0)) { float t = fa / (fa - fc); V x = a + (c - a) * t; lo_p.poly.push_back(x); hi_p.poly.push_back(x); } }
				if (lo_p.poly.size() >= 3) work.push_back(lo_p); if (hi_p.poly.size() >= 3) work.push_back(hi_p);
			}
		}
		printf("early split clipping at %.4f of the scene's longest side: %d triangles -> %zu references (+%.1f %%)\n", presplit, tri_count, rt.size(), 100.0 * (double(rt.size()) / tri_count - 1.0));
		tris.swap(rt); tbox.swap(rb); tri_count = int(tris.size());
	}
	const int bits = argc > 2 ? atoi(argv[2]) : 10;   // Morton bits per axis (the device build: 10)
	const int size_every = argc > 3 ? atoi(argv[3]) : 0;   // extended Morton codes (Vinkler et al. 2017): a bit of the triangle's SIZE after every so many position bits (0: none)
	const int size_first = argc > 4 ? atoi(argv[4]) : 0;   // position bits in front of the first size bit
	auto expand64 = [](uint64_t v) { uint64_t r = 0; for (int i = 0; i < 21; i++) r |= ((v >> i) & 1ull) << (3 * i); return r; };
	std::vector<uint64_t> keys(tri_count); std::vector<int> order(tri_count);
	for (int i = 0; i < tri_count; i++) { uint64_t q[3]; for (int d = 0; d < 3; d++) { float c = 0.5f * (tbox[i].lo[d] + tbox[i].hi[d]); double g = double(1u << bits); q[d] = uint64_t(std::min(g - 1.0, std::max(0.0, double(c - scene.lo[d]) / double(scene.hi[d] - scene.lo[d]) * g))); }
		keys[i] = (expand64(q[0]) << 2) | (expand64(q[1]) << 1) | expand64(q[2]); order[i] = i;
		if (size_every < 0) {   // level bits: after the position bits of octree level k (3 k of them) one bit "the triangle is no larger than factor x that level's cell"
			const float factor = size_first > 0 ? size_first / 100.0f : 1.0f; const int max_level = -size_every;
			float diag = 0; for (int d = 0; d < 3; d++) diag = std::max(diag, (tbox[i].hi[d] - tbox[i].lo[d]) / (scene.hi[d] - scene.lo[d]));
			uint64_t out = 0; int emitted = 0;
			for (int level = 1; level <= bits && emitted < 60; level++) {
				for (int a = 0; a < 3; a++) { int pb = 3 * (bits - level) + (2 - a); out = (out << 1) | ((keys[i] >> pb) & 1ull); emitted++; }
				if (level <= max_level) { out = (out << 1) | (diag <= factor * ldexpf(1.0f, -level) ? 1ull : 0ull); emitted++; }
			}
			keys[i] = out << (63 - emitted);
		} else if (size_every > 0) {   // position bits MSB first (x, y, z, x, ...), a size bit (MSB first; 0 = large) spliced in after every `size_every` of them
			float diag = 0; for (int d = 0; d < 3; d++) { float e = (tbox[i].hi[d] - tbox[i].lo[d]) / (scene.hi[d] - scene.lo[d]); diag = std::max(diag, e); }
			uint64_t sz = uint64_t(std::min(double((1u << bits) - 1), std::max(0.0, (1.0 - double(diag)) * double(1u << bits))));   // small triangles: large value
			uint64_t out = 0; int emitted = 0, size_bit = bits - 1, since = size_every - size_first;
			for (int pb = 3 * bits - 1; pb >= 0 && emitted < 63; pb--) {
				if (since == size_every && size_bit >= 0) { out = (out << 1) | ((sz >> size_bit) & 1ull); size_bit--; emitted++; since = 0; }
				out = (out << 1) | ((keys[i] >> pb) & 1ull); emitted++; since++;
			}
			keys[i] = out << (63 - emitted);
		} }
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return keys[a] < keys[b]; });
	std::vector<uint64_t> skeys(tri_count); std::vector<Box> sbox(tri_count); for (int i = 0; i < tri_count; i++) { skeys[i] = keys[order[i]]; sbox[i] = tbox[order[i]]; }
	{ long dup = 0; for (int i = 1; i < tri_count; i++) if (skeys[i] == skeys[i - 1]) dup++; printf("%d triangles, %d Morton bits per axis, %ld keys equal to their predecessor\n", tri_count, bits, dup); }

	std::vector<Ray> primary, bounce;
	V pos = { cam[0], cam[1], cam[2] }, blc = { cam[3], cam[4], cam[5] }, xa = { cam[6], cam[7], cam[8] }, ya = { cam[9], cam[10], cam[11] };
	for (int y = 0; y < H; y += 6) for (int x = 0; x < W; x += 3) primary.push_back({ pos, normalize(blc + xa * (x + 0.5f) + ya * (y + 0.5f)) });

	struct Variant { const char * name; Policy policy; bool area_first; int candidates; };
	const Variant variants[] = { { "linear BVH (highest differing bit, widest piece first)", LBVH, false, 0 }, { "linear BVH cuts, largest area x count first", LBVH, true, 0 },
		{ "linear BVH cuts, largest AREA first", LBVH, true, -1 }, { "linear BVH cuts, area first, pieces of <= 6 triangles cut first", LBVH, true, -4 }, { "linear BVH cuts, area first, pieces of <= 9 cut first", LBVH, true, -5 }, { "linear BVH cuts, area first, pieces of <= 12 cut first", LBVH, true, -6 }, { "linear BVH cuts, largest area x sqrt(count) first", LBVH, true, -2 }, { "linear BVH cuts, largest area x log2(count) first", LBVH, true, -3 },
		{ "SAH cut over ALL positions of the Morton order, widest first", SAH_EXACT, false, 0 }, { "SAH cut over all positions, largest area x count first", SAH_EXACT, true, 0 },
		{ "SAH cut over 63 candidates x 3 refinements, largest area x count first", SAH_CANDIDATES, true, 63 }, { "SAH cut over 15 candidates x 3 refinements, largest area x count first", SAH_CANDIDATES, true, 15 },
		{ "SAH over the cell boundaries of the next 2 Morton bits, area x count first", SAH_ALIGNED, true, 2 }, { "SAH over the cell boundaries of the next 3 Morton bits, area x count first", SAH_ALIGNED, true, 3 },
		{ "SAH over the cell boundaries of the next 6 Morton bits, area x count first", SAH_ALIGNED, true, 6 } };
	bool have_bounce = false;
	for (const Variant & v : variants) {
		Tree t; t.order = order; Build b { skeys, sbox, v.policy, v.area_first, v.candidates, {}, {} };
		build_node(t, b, 0, tri_count);
		if (!have_bounce) { std::mt19937 rng(1); std::normal_distribution<float> nd;
			for (auto & r : primary) { float best = 1e30f; int hit = -1; long a = 0, c = 0; trace(t, r, best, hit, a, c); if (hit >= 0) bounce.push_back({ r.o + r.d * (best * 0.999f), normalize(V{ nd(rng), nd(rng), nd(rng) }) }); }
			have_bounce = true; }
		double root_area = scene.area(), sah = 0; for (const Node & n : t.nodes) for (int c = 0; c < n.n; c++) sah += n.child_box[c].area() / root_area * (n.child[c] >= 0 ? 1.0 : 0.3 * n.count[c]);
		long pn = 0, pt = 0, bn = 0, bt = 0;
		#pragma omp parallel for reduction(+:pn, pt)
		for (size_t i = 0; i < primary.size(); i++) { float best = 1e30f; int hit = -1; long a = 0, c = 0; trace(t, primary[i], best, hit, a, c); pn += a; pt += c; }
		#pragma omp parallel for reduction(+:bn, bt)
		for (size_t i = 0; i < bounce.size(); i++) { float best = 1e30f; int hit = -1; long a = 0, c = 0; trace(t, bounce[i], best, hit, a, c); bn += a; bt += c; }
		printf("%-74s nodes %7zu  cost %7.1f | primary %5.2f nodes %5.2f tris | bounce %5.2f nodes %5.2f tris\n", v.name, t.nodes.size(), sah, double(pn) / primary.size(), double(pt) / primary.size(), double(bn) / bounce.size(), double(bt) / bounce.size());
	}
	for (int bins : { 8, 16, 32 }) for (int area_first = 0; area_first < 2; area_first++) {
		std::vector<int> ids(tri_count); for (int i = 0; i < tri_count; i++) ids[i] = i;
		cut_depth.assign(size_t(tri_count) + 1, 0x7fffffff);
		binned_sah(ids, tbox, 0, tri_count, 0, bins);
		std::vector<Box> ob(tri_count); for (int i = 0; i < tri_count; i++) ob[i] = tbox[ids[i]];
		Tree t; t.order = ids; Build b { skeys, ob, RECORDED, area_first != 0, -1, {}, {} };
		build_node(t, b, 0, tri_count);
		long pn = 0, pt = 0, bn = 0, bt = 0;
		#pragma omp parallel for reduction(+:pn, pt)
		for (size_t i = 0; i < primary.size(); i++) { float best = 1e30f; int hit = -1; long a = 0, c = 0; trace(t, primary[i], best, hit, a, c); pn += a; pt += c; }
		#pragma omp parallel for reduction(+:bn, bt)
		for (size_t i = 0; i < bounce.size(); i++) { float best = 1e30f; int hit = -1; long a = 0, c = 0; trace(t, bounce[i], best, hit, a, c); bn += a; bt += c; }
		printf("binned SAH, %2d bins, binary tree collapsed 8-wide, %-34s nodes %7zu               | primary %5.2f nodes %5.2f tris | bounce %5.2f nodes %5.2f tris\n", bins, area_first ? "largest area first" : "widest piece first", t.nodes.size(), double(pn) / primary.size(), double(pt) / primary.size(), double(bn) / bounce.size(), double(bt) / bounce.size());
	}
	const int morton_bits = bits;
	for (int extents = 0; extents < 2; extents++) for (int K : { 3, 64, 256, 512, 1024, 2048 }) for (int max_depth : { 24, 32 }) {
		if (max_depth == 32 && K > 8) continue;
		std::vector<int> ids(tri_count); for (int i = 0; i < tri_count; i++) ids[i] = i;
		std::vector<TopLeaf> leaves; hybrid_top(ids, tbox, 0u, 0, K, max_depth, getenv("HYBRID_BINS") ? atoi(getenv("HYBRID_BINS")) : 16, extents, leaves);
		std::vector<uint64_t> hk(tri_count); int deepest = 0;
		for (const TopLeaf & leaf : leaves) { deepest = std::max(deepest, leaf.depth);
			uint64_t high = leaf.depth == 0 ? 0ull : uint64_t(leaf.path) << (32 - leaf.depth);
			for (int id : leaf.ids) { uint64_t q[3]; for (int d = 0; d < 3; d++) { float c = 0.5f * (tbox[id].lo[d] + tbox[id].hi[d]); double g = double(1u << morton_bits), e = double(leaf.box.hi[d] - leaf.box.lo[d]); q[d] = e > 0 ? uint64_t(std::min(g - 1.0, std::max(0.0, double(c - leaf.box.lo[d]) / e * g))) : 0ull; }
				hk[id] = (high << 32) | (expand64(q[0]) << 2) | (expand64(q[1]) << 1) | expand64(q[2]); } }
		std::vector<int> ho(tri_count); for (int i = 0; i < tri_count; i++) ho[i] = i;
		std::stable_sort(ho.begin(), ho.end(), [&](int a, int b) { return hk[a] < hk[b]; });
		std::vector<uint64_t> sk(tri_count); std::vector<Box> sb(tri_count); for (int i = 0; i < tri_count; i++) { sk[i] = hk[ho[i]]; sb[i] = tbox[ho[i]]; }
		Tree t; t.order = ho; Build b { sk, sb, LBVH, true, -1, {}, {} };
		build_node(t, b, 0, tri_count);
		long pn = 0, pt = 0, bn = 0, bt = 0;
		#pragma omp parallel for reduction(+:pn, pt)
		for (size_t i = 0; i < primary.size(); i++) { float best = 1e30f; int hit = -1; long a = 0, c = 0; trace(t, primary[i], best, hit, a, c); pn += a; pt += c; }
		#pragma omp parallel for reduction(+:bn, bt)
		for (size_t i = 0; i < bounce.size(); i++) { float best = 1e30f; int hit = -1; long a = 0, c = 0; trace(t, bounce[i], best, hit, a, c); bn += a; bt += c; }
		printf("hybrid: binned SAH (16 bins over the %s box) down to <= %4d refs / %2d levels (%6zu pieces, deepest %2d), Morton below  nodes %7zu | primary %5.2f nodes %5.2f tris | bounce %5.2f nodes %5.2f tris\n",
		       extents ? "node's" : "centroid", K, max_depth, leaves.size(), deepest, t.nodes.size(), double(pn) / primary.size(), double(pt) / primary.size(), double(bn) / bounce.size(), double(bt) / bounce.size());
		fflush(stdout);
	}
	return 0;
}
