// pipeline_check.cpp -- CPU check of the lane state machine of bvh8_trace_engine_flat_pipelined (kernels_trace.hip) before GPU minutes
// are spent on it: one lane, the round written exactly as the kernel writes it (slab tests on the node fetched a round earlier, pop
// before the triangle loads, choice of the next node before the triangle tests), against the plain sequential walk. Compared per
// ray: the hit (t bits, triangle), the occlusion answer, and the SEQUENCE of nodes and triangles visited (which is what keeps ties
// and the counters where the oracle has them).
//   g++ -O2 -std=c++17 -fopenmp -o /tmp/pipeline_check tools/wave_sim/pipeline_check.cpp
//   python tools/wave_sim/export_merged.py sponza /tmp/flat.bin 2 && /tmp/pipeline_check /tmp/flat.bin
#define main wave_sim_main
#include "wave_sim.cpp"
#undef main

struct Visit { std::vector<int> nodes, tris; };

// the sequential walk of one ray over a one-tree scene (rounds of bvh8_trace_engine<FLAT> for one lane)
static bool walk_reference(const Scene & s, const Ray & ray, bool shadow, int batch, Hit & hit, Visit & v) {
	f3 inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z); unsigned oct = octant_inv4(ray.d);
	uint32_t cg_x = 0, cg_y = 0x80000000u, tg_x = 0, tg_y = 0; std::vector<std::pair<uint32_t, uint32_t>> stack;
	while (true) {
		if (tg_y == 0 && (cg_y & 0xff000000u)) {
			unsigned hits_imask = cg_y, off = msb(hits_imask), base = cg_x; cg_y &= ~(1u << off);
			if (cg_y & 0xff000000u) stack.push_back({ cg_x, cg_y });
			unsigned slot = (off - 24) ^ (oct & 0xffu), rel = __builtin_popcount(hits_imask & ~(0xffffffffu << slot));
			const uint8_t * node = &s.nodes[size_t(base + rel) * 80]; v.nodes.push_back(int(base + rel));
			unsigned hm = node_intersect(ray, inv, oct, shadow ? ray.tmax : hit.t, node);
			uint32_t w[8]; memcpy(w, node, 32);
			cg_x = w[4]; tg_x = w[5]; cg_y = (hm & 0xff000000u) | (w[3] >> 24); tg_y = hm & 0x00ffffffu;
		}
		bool occluded = false;
		for (int k = 0; k < batch && tg_y != 0; k++) { int ti = int(msb(tg_y)); tg_y &= ~(1u << ti);
			if (!occluded) { v.tris.push_back(int(tg_x) + ti); if (triangle_test(s, shadow, int(tg_x) + ti, ray, ray.tmax, hit)) occluded = true; } }
		if (shadow && occluded) return true;
		if (tg_y == 0 && (cg_y & 0xff000000u) == 0) {
			if (stack.empty()) return false;
			cg_x = stack.back().first; cg_y = stack.back().second; stack.pop_back();
		}
	}
}

// the pipelined lane
static bool walk_pipelined(const Scene & s, const Ray & ray, bool shadow, int batch, Hit & hit, Visit & v) {
	f3 inv = mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z); unsigned oct = octant_inv4(ray.d);
	uint32_t cg_x = 0, cg_y = 0, tg_x = 0, tg_y = 0; std::vector<std::pair<uint32_t, uint32_t>> stack;
	int pn = 0; bool have_next = true;   // the root's loads are issued when the ray is fetched
	while (true) {
		if (have_next) {
			const uint8_t * node = &s.nodes[size_t(pn) * 80]; v.nodes.push_back(pn);
			unsigned hm = node_intersect(ray, inv, oct, shadow ? ray.tmax : hit.t, node);
			uint32_t w[8]; memcpy(w, node, 32);
			cg_x = w[4]; cg_y = (hm & 0xff000000u) | (w[3] >> 24); tg_x = w[5]; tg_y = hm & 0x00ffffffu;
			have_next = false;
		}
		const bool leaf_done = __builtin_popcount(tg_y) <= batch;
		if (leaf_done && (cg_y & 0xff000000u) == 0 && !stack.empty()) { cg_x = stack.back().first; cg_y = stack.back().second; stack.pop_back(); }
		const bool need_node = leaf_done && (cg_y & 0xff000000u) != 0;
		int tri_id[8]; for (int k = 0; k < batch; k++) { tri_id[k] = -1; if (tg_y != 0) { int ti = int(msb(tg_y)); tg_y &= ~(1u << ti); tri_id[k] = int(tg_x) + ti; } }
		unsigned next = 0;
		if (need_node) {
			unsigned hits_imask = cg_y, off = msb(hits_imask); cg_y &= ~(1u << off);
			if (cg_y & 0xff000000u) stack.push_back({ cg_x, cg_y });
			unsigned slot = (off - 24) ^ (oct & 0xffu), rel = __builtin_popcount(hits_imask & ~(0xffffffffu << slot));
			next = cg_x + rel; cg_y = 0;
		}
		pn = int(next); have_next = need_node;
		bool occluded = false;
		for (int k = 0; k < batch; k++) if (tri_id[k] >= 0) { if (!occluded) v.tris.push_back(tri_id[k]); /* (the kernel runs a shadow ray's second test regardless; its answer is the same) */
			if (triangle_test(s, shadow, tri_id[k], ray, ray.tmax, hit)) occluded = true; }
		if (shadow && occluded) return true;
		if (!have_next && tg_y == 0) return false;
	}
}

int main(int argc, char ** argv) {
	if (argc < 2) { fprintf(stderr, "usage: pipeline_check flat_scene.bin\n"); return 1; }
	Scene s; FILE * f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 1; }
	int hdr[6]; if (fread(hdr, 4, 6, f) != 6) return 1; s.node_count = hdr[0]; s.tri_count = hdr[1]; s.mesh_count = hdr[2]; s.tlas_count = hdr[3]; s.width = hdr[4]; s.height = hdr[5];
	if (s.tlas_count != 0) { fprintf(stderr, "a one-tree scene (export_merged.py) is needed\n"); return 1; }
	s.nodes.resize(size_t(s.node_count) * 80); s.tris.resize(size_t(s.tri_count) * 9); s.roots.resize(s.mesh_count); s.xinv.resize(size_t(s.mesh_count) * 12);
	if (fread(s.nodes.data(), 1, s.nodes.size(), f) != s.nodes.size() || fread(s.tris.data(), 4, s.tris.size(), f) != s.tris.size() || fread(s.roots.data(), 4, s.roots.size(), f) != s.roots.size() || fread(s.xinv.data(), 4, s.xinv.size(), f) != s.xinv.size() || fread(s.cam, 4, 15, f) != 15) return 1;
	fclose(f);
	std::vector<Ray> rays;
	f3 pos = mk3(s.cam[0], s.cam[1], s.cam[2]), blc = mk3(s.cam[3], s.cam[4], s.cam[5]), xa = mk3(s.cam[6], s.cam[7], s.cam[8]), ya = mk3(s.cam[9], s.cam[10], s.cam[11]);
	for (int y = 0; y < s.height; y += 8) for (int x = 0; x < s.width; x += 2) rays.push_back({ pos, normalize(blc + xa * (x + 0.5f) + ya * (y + 0.5f)), INFINITY });
	size_t primary = rays.size();
	std::mt19937 rng(7); std::normal_distribution<float> nd; f3 light = mk3(0.0f, 12.0f, 0.0f);
	std::vector<Ray> shadow_rays;
	for (size_t i = 0; i < primary; i++) { Hit h; Visit v; walk_reference(s, rays[i], false, 2, h, v);
		if (h.tri >= 0) { f3 p = rays[i].o + rays[i].d * (h.t * 0.999f); rays.push_back({ p, normalize(mk3(nd(rng), nd(rng), nd(rng))), INFINITY });
			f3 to = light - p; float dist = sqrtf(dot_fma(to, to)); shadow_rays.push_back({ p, to * (1.0f / dist), dist }); } }
	long bad = 0, total = 0;
	for (int batch : { 1, 2, 3 }) for (int kind = 0; kind < 2; kind++) {
		const std::vector<Ray> & set = kind ? shadow_rays : rays; long mismatches = 0; double nodes = 0, tris = 0;
		#pragma omp parallel for reduction(+:mismatches, nodes, tris)
		for (size_t i = 0; i < set.size(); i++) {
			Hit ha, hb; Visit va, vb;
			bool oa = walk_reference(s, set[i], kind != 0, batch, ha, va), ob = walk_pipelined(s, set[i], kind != 0, batch, hb, vb);
			bool same = oa == ob && va.nodes == vb.nodes && va.tris == vb.tris && (kind || (ha.tri == hb.tri && as_uint(ha.t) == as_uint(hb.t)));
			if (!same) mismatches++;
			nodes += double(va.nodes.size()); tris += double(va.tris.size());
		}
		printf("batch %d  %-8s %8zu rays  %.2f nodes %.2f triangles per ray  mismatches %ld\n", batch, kind ? "shadow" : "closest", set.size(), nodes / set.size(), tris / set.size(), mismatches);
		bad += mismatches; total += long(set.size());
	}
	printf("%s (%ld of %ld differ)\n", bad ? "FAILED" : "pipelined lane == sequential walk", bad, total);
	return bad ? 1 : 0;
}
