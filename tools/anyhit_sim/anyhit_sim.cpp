// anyhit_sim.cpp -- what a shadow ray's walk costs under different any-hit strategies (development tool, round 4).
//
// Input: the flattened tree the product stages for the benchmark scene (/tmp/oexp/flat.bin: node count, triangle count, 80-byte
// nodes, 9 floats per triangle) and the shadow rays of real frames (/tmp/oexp/shadow.bin: 13 words per ray -- origin, direction,
// max distance, pixel, bounce << 16 | sample, occluded, occluding triangle, node steps, triangle tests), both written by throw-away
// scripts around a locally patched copy of the CPU checker (never committed: the checker stays what it is). The traversal
// arithmetic is restated here, nothing under oracle/ is used. Results: profiles/r04_shadow_rays.txt.
//   g++ -O2 -std=c++17 -fopenmp -mfma -ffp-contract=off -o /tmp/anyhit_sim tools/anyhit_sim/anyhit_sim.cpp && /tmp/anyhit_sim
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
struct f3 { float x, y, z; };
static inline f3 mk3(float x, float y, float z) { return { x, y, z }; }
static inline f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline f3 operator*(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
static inline float dot_fma(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline f3 cross_fma(f3 a, f3 b) { return { fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)) }; }
static inline unsigned msb(unsigned x) { return 31u - unsigned(__builtin_clz(x)); }
static inline unsigned extract_byte(unsigned x, unsigned i) { return (x >> (i * 8)) & 0xffu; }
static inline unsigned sign_extend_s8x4(unsigned x) { return ((x >> 7) & 0x01010101u) * 0xffu; }
static inline float as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
struct Ray { f3 o, d; float tmax; int pixel, bounce, sample, occ, tri, nodes, tris; };
std::vector<uint8_t> nodes; std::vector<float> tris; int node_count, tri_count;
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
static bool tri_test(int id, const Ray & ray) {
	const float * tr = &tris[size_t(id) * 9];
	f3 p0 = mk3(tr[0], tr[1], tr[2]), e1 = mk3(tr[3], tr[4], tr[5]), e2 = mk3(tr[6], tr[7], tr[8]);
	f3 h = cross_fma(ray.d, e2); float a = dot_fma(e1, h), f = 1.0f / a; f3 sv = ray.o - p0; float u = f * dot_fma(sv, h);
	if (u >= 0 && u <= 1) { f3 q = cross_fma(sv, e1); float v = f * dot_fma(ray.d, q);
		if (v >= 0 && u + v <= 1) { float t = f * dot_fma(e2, q); if (t > 0 && t < ray.tmax) return true; } }
	return false;
}
// mode 0: ordered as the product (pop highest bit first = near first). mode 1: reversed (lowest inner bit first). mode 2: triangles of a node AFTER descending (children first)
struct Res { int nodes = 0, tris = 0, tri = -1; bool occ = false; };
static Res walk(const Ray & ray, int mode) {
	Res r; f3 inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z); unsigned oct = octant_inv4(ray.d);
	uint32_t stack[64][2]; int sp = 0; uint32_t cgx = 0, cgy = 0x80000000u, tgx = 0, tgy = 0;
	while (true) {
		if (cgy & 0xff000000u) {
			unsigned hits_imask = cgy, off = mode == 1 ? unsigned(__builtin_ctz(cgy & 0xff000000u)) : msb(hits_imask), base = cgx; cgy &= ~(1u << off);
			if (cgy & 0xff000000u) { stack[sp][0] = cgx; stack[sp][1] = cgy; sp++; }
			unsigned slot = (off - 24) ^ (oct & 0xffu), rel = __builtin_popcount(hits_imask & ~(0xffffffffu << slot));
			const uint8_t * node = &nodes[size_t(base + rel) * 80]; r.nodes++;
			unsigned hm = node_intersect(ray, inv, oct, ray.tmax, node);
			uint32_t w[8]; memcpy(w, node, 32);
			cgx = w[4]; tgx = w[5]; cgy = (hm & 0xff000000u) | (w[3] >> 24); tgy = hm & 0x00ffffffu;
		} else { tgx = cgx; tgy = cgy; cgx = cgy = 0; }
		while (tgy) { int ti = int(msb(tgy)); tgy &= ~(1u << ti); r.tris++; if (tri_test(int(tgx) + ti, ray)) { r.occ = true; r.tri = int(tgx) + ti; return r; } }
		if ((cgy & 0xff000000u) == 0) { if (sp == 0) return r; sp--; cgx = stack[sp][0]; cgy = stack[sp][1]; }
	}
}
int main(int argc, char ** argv) {
	FILE * f = fopen("/tmp/oexp/flat.bin", "rb"); int hdr[2]; fread(hdr, 4, 2, f); node_count = hdr[0]; tri_count = hdr[1];
	nodes.resize(size_t(node_count) * 80); tris.resize(size_t(tri_count) * 9); fread(nodes.data(), 1, nodes.size(), f); fread(tris.data(), 4, tris.size(), f); fclose(f);
	f = fopen("/tmp/oexp/shadow.bin", "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f) / 52; fseek(f, 0, SEEK_SET);
	std::vector<Ray> rays(n);
	for (size_t i = 0; i < n; i++) { float rec[13]; fread(rec, 4, 13, f); int ir[6]; memcpy(ir, rec + 7, 24);
		rays[i] = { mk3(rec[0], rec[1], rec[2]), mk3(rec[3], rec[4], rec[5]), rec[6], ir[0], ir[1] >> 16, ir[1] & 0xffff, ir[2], ir[3], ir[4], ir[5] }; }
	fclose(f);
	printf("%zu shadow rays\n", n);
	for (int mode = 0; mode < 2; mode++) {
		std::vector<Res> res(n);
		#pragma omp parallel for schedule(dynamic, 4096)
		for (size_t i = 0; i < n; i++) res[i] = walk(rays[i], mode);
		double nn[2] = {}, tt[2] = {}, cnt[2] = {}; size_t mism = 0;
		for (size_t i = 0; i < n; i++) { nn[res[i].occ] += res[i].nodes; tt[res[i].occ] += res[i].tris; cnt[res[i].occ]++; if (int(res[i].occ) != rays[i].occ) mism++; }
		printf("mode %d: occluded %.4f  nodes/tris occluded %.2f / %.2f  free %.2f / %.2f  all %.2f / %.2f  mismatches vs dump %zu\n", mode, cnt[1] / n, nn[1] / cnt[1], tt[1] / cnt[1], nn[0] / cnt[0], tt[0] / cnt[0], (nn[0] + nn[1]) / n, (tt[0] + tt[1]) / n, mism);
		if (mode == 0) {
			// caches: key = pixel (bounce-0 rays only | all rays), state carried across samples; within a sample: rays see the state left by previous samples (variant A) or by every earlier ray (variant B)
			for (int all_bounces = 0; all_bounces < 2; all_bounces++) for (int immediate = 0; immediate < 2; immediate++) for (int group = 1; group <= 4; group *= 4) {
				std::vector<int> cache(1920 * 1080, -1), pending;
				double tested = 0, hit = 0, saved_n = 0, saved_t = 0, elig = 0; int cur_sample = group == 1 ? 0 : 0;
				std::vector<std::pair<int,int>> updates;
				for (size_t i = 0; i < n; i++) {   // the dump is in sample-major, bounce-major order
					const Ray & r = rays[i];
					if (r.sample / group != cur_sample) { for (auto & u : updates) cache[u.first] = u.second; updates.clear(); cur_sample = r.sample / group; }
					if (!all_bounces && r.bounce != 0) continue;
					elig++;
					int c = cache[r.pixel];
					if (c >= 0) { tested++; if (tri_test(c, r)) { hit++; saved_n += res[i].nodes; saved_t += res[i].tris - 1; } else saved_t -= 1; }
					if (res[i].occ) { if (immediate) cache[r.pixel] = res[i].tri; else updates.push_back({ r.pixel, res[i].tri }); }
				}
				printf("  cache per pixel, %s, %s, launch = %d samples: eligible %.3f tested %.3f of rays, hits %.3f of rays (%.3f of tested); node steps saved %.3f, triangle tests saved %.3f of the shadow total\n",
					all_bounces ? "all bounces" : "bounce 0 only", immediate ? "updated at once" : "updated between launches", group, elig / n, tested / n, hit / n, hit / std::max(1.0, tested), saved_n / (nn[0] + nn[1]), saved_t / (tt[0] + tt[1]));
			}
		}
	}
	return 0;
}
