// StaticBVHBuilder -- the binary tree behind the flattened static geometry (Integrator::init_geometry, cpu_config.merge_static).
//
// Not one of the reference's builders: those (BVH.cpp, SBVH.cpp here) restate Builders/SAHBuilder.cpp and SBVHBuilder.cpp step
// by step because their trees have to come out byte-identical, and they are sequential -- the spatial-split builder takes 10 s
// for the 262 k triangles of Sponza on one core. The flattened tree has no reference bytes to match; it only has to be a good
// tree, quickly. So: the same two kinds of split as Stich et al.'s SBVH -- object splits and spatial splits with reference
// unsplitting, chosen by SAH cost, spatial ones only tried where the object split's children overlap -- but found by BINNING
// (32 bins per axis) instead of full sweeps over three sorted lists, over references that carry their own clipped box, and
// built by all host threads: the top of the tree is split level by level (the nodes of a level in parallel) until the pieces are
// small, the pieces are independent subtrees built in parallel. One triangle reference per leaf, as BVH8Converter wants them; a triangle
// cut by spatial splits appears in several leaves (the device copies it once per reference).
#include "BVH.h"
#include "Config.h"

#ifndef STATIC_BVH_BINS
#define STATIC_BVH_BINS 32
#endif

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace {

constexpr int BINS = STATIC_BVH_BINS;

struct Ref { AABB box; int triangle; };

struct Node { AABB box; int left = -1, right = -1; int triangle = -1; int axis = 0; int subtree = -1; };

struct Tree { std::vector<Node> nodes; };

// Sutherland-Hodgman of a convex polygon against the half space x[axis] >= plane (keep_above) or <= plane
int clip_polygon(const Vector3 * in, int count, int axis, float plane, bool keep_above, Vector3 * out) {
	int n = 0;
	for (int i = 0; i < count; i++) {
		const Vector3 & a = in[i], & b = in[(i + 1) % count];
		float da = keep_above ? a[axis] - plane : plane - a[axis];
		float db = keep_above ? b[axis] - plane : plane - b[axis];
		if (da >= 0.0f) out[n++] = a;
		if ((da > 0.0f && db < 0.0f) || (da < 0.0f && db > 0.0f)) {
			float t = da / (da - db);
			Vector3 p = a + t * (b - a);
			p[axis] = plane;
			out[n++] = p;
		}
	}
	return n;
}

// Box of the part of the triangle inside the slab lo <= x[axis] <= hi, never larger than the reference's own box
AABB clipped_box(const Triangle & triangle, const AABB & ref_box, int axis, float lo, float hi) {
	Vector3 a[8] = { triangle.position_0, triangle.position_1, triangle.position_2 }, b[8];
	int n = clip_polygon(a, 3, axis, lo, true, b);
	n = clip_polygon(b, n, axis, hi, false, a);
	if (n == 0) return AABB::create_empty();
	AABB box = AABB::from_points(a, n); // (pads flat boxes like every box of a triangle)
	box.min = Vector3::max(box.min, ref_box.min);
	box.max = Vector3::min(box.max, ref_box.max);
	for (int d = 0; d < 3; d++) if (box.max[d] < box.min[d]) box.max[d] = box.min[d];
	return box;
}



struct Split { float cost = INFINITY; int axis = -1; int bin = -1; float plane = 0.0f; AABB left, right; int count_left = 0, count_right = 0; };

struct Builder {
	const std::vector<Triangle> & triangles;
	float inv_root_area;
	float alpha;

	// SAH over 32 centroid bins per axis
	Split find_object_split(const std::vector<Ref> & refs, const AABB & centroid_box) const {
		Split best;
		for (int axis = 0; axis < 3; axis++) {
			float lo = centroid_box.min[axis], extent = centroid_box.max[axis] - lo;
			if (!(extent > 0.0f)) continue;
			float scale = float(BINS) / extent;
			AABB box[BINS]; int count[BINS];
			for (int k = 0; k < BINS; k++) { box[k] = AABB::create_empty(); count[k] = 0; }
			for (const Ref & r : refs) {
				int k = std::min(BINS - 1, std::max(0, int((r.box.get_center()[axis] - lo) * scale)));
				box[k].expand(r.box); count[k]++;
			}
			AABB right_box[BINS]; int right_count[BINS];
			AABB grown = AABB::create_empty(); int n = 0;
			for (int k = BINS - 1; k > 0; k--) { grown.expand(box[k]); n += count[k]; right_box[k] = grown; right_count[k] = n; }
			grown = AABB::create_empty(); n = 0;
			for (int k = 1; k < BINS; k++) {
				grown.expand(box[k - 1]); n += count[k - 1];
				if (n == 0 || right_count[k] == 0) continue;
				float cost = grown.surface_area() * float(n) + right_box[k].surface_area() * float(right_count[k]);
				if (cost < best.cost) { best.cost = cost; best.axis = axis; best.bin = k; best.plane = lo + float(k) / scale; best.left = grown; best.right = right_box[k]; best.count_left = n; best.count_right = right_count[k]; }
			}
		}
		return best;
	}

	// SAH over 32 slabs of the node box per axis; a reference is chopped into the slabs it spans
	Split find_spatial_split(const std::vector<Ref> & refs, const AABB & node_box) const {
		Split best;
		for (int axis = 0; axis < 3; axis++) {
			float lo = node_box.min[axis], extent = node_box.max[axis] - lo;
			if (!(extent > 0.0f)) continue;
			float width = extent / float(BINS), scale = float(BINS) / extent;
			AABB box[BINS]; int entries[BINS], exits[BINS];
			for (int k = 0; k < BINS; k++) { box[k] = AABB::create_empty(); entries[k] = exits[k] = 0; }
			for (const Ref & r : refs) {
				int k0 = std::min(BINS - 1, std::max(0, int((r.box.min[axis] - lo) * scale)));
				int k1 = std::min(BINS - 1, std::max(k0, int((r.box.max[axis] - lo) * scale)));
				entries[k0]++; exits[k1]++;
				if (k0 == k1) { box[k0].expand(r.box); continue; }
				for (int k = k0; k <= k1; k++) {
					AABB part = clipped_box(triangles[r.triangle], r.box, axis, lo + float(k) * width, lo + float(k + 1) * width);
					if (!part.is_empty()) box[k].expand(part);
				}
			}
			AABB right_box[BINS]; int right_count[BINS];
			AABB grown = AABB::create_empty(); int n = 0;
			for (int k = BINS - 1; k > 0; k--) { grown.expand(box[k]); n += exits[k]; right_box[k] = grown; right_count[k] = n; }
			grown = AABB::create_empty(); n = 0;
			for (int k = 1; k < BINS; k++) {
				grown.expand(box[k - 1]); n += entries[k - 1];
				if (n == 0 || right_count[k] == 0 || grown.is_empty() || right_box[k].is_empty()) continue;
				float cost = grown.surface_area() * float(n) + right_box[k].surface_area() * float(right_count[k]);
				if (cost < best.cost) { best.cost = cost; best.axis = axis; best.bin = k; best.plane = lo + float(k) * width; best.left = grown; best.right = right_box[k]; best.count_left = n; best.count_right = right_count[k]; }
			}
		}
		return best;
	}

	// Splits `refs` into two non-empty sets; returns the axis
	// `budget`: how many references spatial splits may still ADD below this node (in: the node's share, out: what is left for its
	// children). Without one, triangles that overlap everywhere (a stack of large coplanar triangles) are cut again and again:
	// 500 of them became 71 000 references. The share is dealt down the tree in proportion to the children's reference counts,
	// so the tree does not depend on the order in which threads build its parts.
	int partition(std::vector<Ref> & refs, const AABB & node_box, std::vector<Ref> & left, std::vector<Ref> & right, AABB & left_box, AABB & right_box, long & budget) const {
		AABB centroid_box = AABB::create_empty();
		for (const Ref & r : refs) centroid_box.expand(r.box.get_center());
		Split object = find_object_split(refs, centroid_box);

		Split spatial;
		if (budget > 0) {
			if (object.axis >= 0) {
				AABB shared = AABB::overlap(object.left, object.right);
				if ((shared.is_valid() ? shared.surface_area() : 0.0f) * inv_root_area > alpha) spatial = find_spatial_split(refs, node_box);
			} else {
				spatial = find_spatial_split(refs, node_box); // all centres in one point: only cutting the triangles can separate them
			}
		}

		left.clear(); right.clear();
		left_box = right_box = AABB::create_empty();
		if (spatial.cost < object.cost) {
			int axis = spatial.axis;
			float plane = spatial.plane;
			float n_left = float(spatial.count_left), n_right = float(spatial.count_right);
			AABB cost_left_box = spatial.left, cost_right_box = spatial.right;   // what the search saw; references kept whole grow them
			left.reserve(refs.size()); right.reserve(refs.size());
			for (const Ref & r : refs) {
				if (r.box.max[axis] <= plane) { left.push_back(r); continue; }
				if (r.box.min[axis] >= plane) { right.push_back(r); continue; }
				// straddles the plane: cut it, unless keeping it whole on one side is cheaper (reference unsplitting, Stich et al. section 4.4)
				AABB whole_left = cost_left_box, whole_right = cost_right_box;
				whole_left.expand(r.box); whole_right.expand(r.box);
				float cost_both  = cost_left_box.surface_area() *  n_left         + cost_right_box.surface_area() *  n_right;
				float cost_left  = whole_left   .surface_area() *  n_left         + cost_right_box.surface_area() * (n_right - 1.0f);
				float cost_right = cost_left_box.surface_area() * (n_left - 1.0f) + whole_right   .surface_area() *  n_right;
				if (cost_left < cost_both && cost_left <= cost_right) { left.push_back(r); cost_left_box = whole_left; n_right -= 1.0f; continue; }
				if (cost_right < cost_both)                            { right.push_back(r); cost_right_box = whole_right; n_left -= 1.0f; continue; }
				AABB part_left  = clipped_box(triangles[r.triangle], r.box, axis, -INFINITY, plane);
				AABB part_right = clipped_box(triangles[r.triangle], r.box, axis, plane, INFINITY);
				if (part_left.is_empty() || part_right.is_empty()) { // numerically on one side after all
					if (part_right.is_empty()) left.push_back(r); else right.push_back(r);
					continue;
				}
				left.push_back({ part_left, r.triangle });
				right.push_back({ part_right, r.triangle });
			}
			long added = long(left.size()) + long(right.size()) - long(refs.size());
			if (!left.empty() && !right.empty() && left.size() < refs.size() && right.size() < refs.size() && added <= budget) { // both sides smaller: progress; and paid for
				budget -= added;
				for (const Ref & r : left)  left_box .expand(r.box);
				for (const Ref & r : right) right_box.expand(r.box);
				return axis;
			}
			left.clear(); right.clear(); // did not make progress: fall through to an object split
			left_box = right_box = AABB::create_empty();
		}
		if (object.axis >= 0) {
			int axis = object.axis;
			float lo = centroid_box.min[axis], scale = float(BINS) / (centroid_box.max[axis] - lo);
			for (const Ref & r : refs) {
				int k = std::min(BINS - 1, std::max(0, int((r.box.get_center()[axis] - lo) * scale)));
				if (k < object.bin) { left.push_back(r); left_box.expand(r.box); } else { right.push_back(r); right_box.expand(r.box); }
			}
			if (!left.empty() && !right.empty()) return axis;
			left.clear(); right.clear();
			left_box = right_box = AABB::create_empty();
		}
		// identical centres (copies of one triangle, a stack of coincident references): halve the list
		size_t half = refs.size() / 2;
		for (size_t i = 0; i < refs.size(); i++) { if (i < half) { left.push_back(refs[i]); left_box.expand(refs[i].box); } else { right.push_back(refs[i]); right_box.expand(refs[i].box); } }
		return 0;
	}

	static void share(long budget, size_t n_left, size_t n_right, long & left, long & right) {
		left = long(double(budget) * double(n_left) / double(n_left + n_right));
		right = budget - left;
	}

	void build(Tree & tree, int node, std::vector<Ref> & refs, long budget) const {
		if (refs.size() == 1) { tree.nodes[size_t(node)].triangle = refs[0].triangle; return; }
		std::vector<Ref> left, right;
		AABB left_box, right_box;
		int axis = partition(refs, tree.nodes[size_t(node)].box, left, right, left_box, right_box, budget);
		long budget_left, budget_right;
		share(budget, left.size(), right.size(), budget_left, budget_right);
		std::vector<Ref>().swap(refs);
		int l = int(tree.nodes.size()); tree.nodes.emplace_back(); tree.nodes.emplace_back();
		tree.nodes[size_t(l)].box = left_box; tree.nodes[size_t(l) + 1].box = right_box;
		tree.nodes[size_t(node)].left = l; tree.nodes[size_t(node)].right = l + 1; tree.nodes[size_t(node)].axis = axis;
		build(tree, l, left, budget_left);
		build(tree, l + 1, right, budget_right);
	}
};

} // namespace

void StaticBVHBuilder::build(BVH2 & bvh, const std::vector<Triangle> & triangles, int thread_count) {
	size_t n = triangles.size();
	bvh.indices.clear(); bvh.nodes.clear();
	bvh.nodes.resize(2); // root + the dummy that keeps sibling pairs 64-byte aligned
	memset((void *)bvh.nodes.data(), 0, 2 * sizeof(BVHNode2));
	if (n == 0) return;

	std::vector<Ref> refs(n);
	AABB root_box = AABB::create_empty();
	for (size_t i = 0; i < n; i++) { refs[i] = { triangles[i].get_aabb(), int(i) }; root_box.expand(refs[i].box); }
	if (cpu_config.static_presplit > 0.0f) {   // early split clipping in front of the SAH + spatial-split build as well (experiment of round 5, off: see Config.h)
		float longest = 0.0f;
		for (int d = 0; d < 3; d++) longest = std::max(longest, root_box.max[d] - root_box.min[d]);
		std::vector<int> source; std::vector<float> boxes;
		presplit(triangles, cpu_config.static_presplit * longest, source, boxes);
		refs.resize(source.size());
		for (size_t r = 0; r < source.size(); r++) {
			AABB box; box.min = Vector3(boxes[6 * r], boxes[6 * r + 1], boxes[6 * r + 2]); box.max = Vector3(boxes[6 * r + 3], boxes[6 * r + 4], boxes[6 * r + 5]);
			refs[r] = { box, source[r] };
			root_box.expand(box);   // (a piece's box is an ulp wider than the piece)
		}
		n = refs.size();
	}
	Builder builder { triangles, 1.0f / root_box.surface_area(), cpu_config.sbvh_alpha };

	// top of the tree on this thread, breadth-first, until the pieces are small enough to hand out
	if (thread_count <= 0) thread_count = int(std::max(1u, std::thread::hardware_concurrency()));
	size_t piece = std::max<size_t>(1024, n / (size_t(thread_count) * 8));
	Tree top; top.nodes.emplace_back(); top.nodes[0].box = root_box;
	struct Pending { int node; std::vector<Ref> refs; long budget; };
	std::vector<Pending> open, pieces;
	open.push_back({ 0, std::move(refs), long(n) });   // spatial splits may double the references at most (Sponza: +16 %)
	while (!open.empty()) {
		// one level: its nodes are independent of each other -- split them on as many threads as there are nodes (the first levels
		// have one, two, four nodes: those few passes over all references are what stays sequential), then number the children
		// in the order of their parents so that the tree does not depend on which thread was first
		struct Cut { bool is_piece = true; std::vector<Ref> left, right; AABB left_box, right_box; int axis = 0; long budget_left = 0, budget_right = 0; };
		std::vector<Cut> cuts(open.size());
		std::atomic<size_t> next_node { 0 };
		auto split_level = [&]() {
			for (size_t i = next_node++; i < open.size(); i = next_node++) {
				Pending & p = open[i];
				if (p.refs.size() <= piece) continue;
				Cut & c = cuts[i];
				c.is_piece = false;
				long budget = p.budget;
				c.axis = builder.partition(p.refs, top.nodes[size_t(p.node)].box, c.left, c.right, c.left_box, c.right_box, budget);
				Builder::share(budget, c.left.size(), c.right.size(), c.budget_left, c.budget_right);
				std::vector<Ref>().swap(p.refs);
			}
		};
		{
			std::vector<std::thread> helpers;
			for (int t = 1; t < thread_count && size_t(t) < open.size(); t++) helpers.emplace_back(split_level);
			split_level();
			for (std::thread & t : helpers) t.join();
		}
		std::vector<Pending> next;
		for (size_t i = 0; i < open.size(); i++) {
			Pending & p = open[i];
			Cut & c = cuts[i];
			if (c.is_piece) { pieces.push_back(std::move(p)); continue; }
			int l = int(top.nodes.size()); top.nodes.emplace_back(); top.nodes.emplace_back();
			top.nodes[size_t(l)].box = c.left_box; top.nodes[size_t(l) + 1].box = c.right_box;
			top.nodes[size_t(p.node)].left = l; top.nodes[size_t(p.node)].right = l + 1; top.nodes[size_t(p.node)].axis = c.axis;
			next.push_back({ l, std::move(c.left), c.budget_left }); next.push_back({ l + 1, std::move(c.right), c.budget_right });
		}
		open.swap(next);
	}
	// the pieces in parallel, largest first
	std::vector<Tree> subtrees(pieces.size());
	std::vector<size_t> order(pieces.size());
	for (size_t i = 0; i < order.size(); i++) order[i] = i;
	std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return pieces[a].refs.size() > pieces[b].refs.size(); });
	std::atomic<size_t> cursor { 0 };
	auto work = [&]() {
		for (size_t k = cursor++; k < order.size(); k = cursor++) {
			size_t i = order[k];
			Tree & tree = subtrees[i];
			tree.nodes.reserve(2 * pieces[i].refs.size());
			tree.nodes.emplace_back(); tree.nodes[0].box = top.nodes[size_t(pieces[i].node)].box;
			builder.build(tree, 0, pieces[i].refs, pieces[i].budget);
		}
	};
	std::vector<std::thread> threads;
	for (int t = 1; t < thread_count && size_t(t) < pieces.size(); t++) threads.emplace_back(work);
	work();
	for (std::thread & t : threads) t.join();
	for (size_t i = 0; i < pieces.size(); i++) top.nodes[size_t(pieces[i].node)].subtree = int(i);

	// flatten: children in adjacent slots, leaves numbered in the order they are met
	size_t total = 2;
	for (const Tree & t : subtrees) total += t.nodes.size();
	bvh.nodes.reserve(total + top.nodes.size());
	struct Visit { const Tree * tree; int node; int out; };
	std::vector<Visit> stack;
	stack.push_back({ &top, 0, 0 });
	while (!stack.empty()) {
		Visit v = stack.back(); stack.pop_back();
		const Node * node = &v.tree->nodes[size_t(v.node)];
		if (v.tree == &top && node->subtree >= 0) { v.tree = &subtrees[size_t(node->subtree)]; v.node = 0; node = &v.tree->nodes[0]; }
		BVHNode2 out; memset((void *)&out, 0, sizeof(out));
		out.aabb = node->box;
		if (node->triangle >= 0) {
			out.first = int(bvh.indices.size()); out.count = 1;
			bvh.indices.push_back(node->triangle);
		} else {
			int child = int(bvh.nodes.size());
			bvh.nodes.resize(bvh.nodes.size() + 2);
			out.left = child; out.count = 0; out.axis = unsigned(node->axis);
			stack.push_back({ v.tree, node->right, child + 1 });
			stack.push_back({ v.tree, node->left,  child });     // left first: leaf order follows the tree left to right
		}
		bvh.nodes[size_t(v.out)] = out;
	}
}


// Early split clipping in front of the DEVICE's Morton-order builder (kernels_blas.hip), which cuts its ranges at Morton bits and knows no
// spatial splits: a triangle whose box is longer than `limit` along some axis is cut in two at the middle of that axis, the two polygons are
// boxed tightly, and so on -- every piece becomes a reference (a copy of the triangle for the device, with the piece's box). Blind, i.e. no
// cost function, but it is the handful of floor / wall / curtain triangles spanning half the scene that ruin a Morton build, and those it
// finds: Sponza at limit = 5 % of the longest side: +5 % references, node steps per bounce ray 15.0 -> 9.5, triangle tests 10.4 -> 6.6 in
// the builder's CPU model (tools/blas_proto/morton_sah.cpp, profiles/r05_device_blas_presplit.txt). Boxes are the pieces' own (a piece is
// inside its triangle, so the pieces' boxes cover the triangle), widened by an ulp: a cut point is an interpolation, rounded.
void StaticBVHBuilder::presplit(const std::vector<Triangle> & triangles, float limit, std::vector<int> & source, std::vector<float> & boxes, int max_pieces) {
	source.clear(); boxes.clear();
	source.reserve(triangles.size() + triangles.size() / 8); boxes.reserve(6 * (triangles.size() + triangles.size() / 8));
	constexpr int MAX_VERTICES = 24;   // a cut adds at most one vertex to a convex polygon
	struct Piece { Vector3 v[MAX_VERTICES]; int n; bool clipped; };
	std::vector<Piece> work;
	for (size_t t = 0; t < triangles.size(); t++) {
		work.clear();
		Piece whole; whole.v[0] = triangles[t].position_0; whole.v[1] = triangles[t].position_1; whole.v[2] = triangles[t].position_2; whole.n = 3; whole.clipped = false;
		work.push_back(whole);
		int made = 0;
		auto emit = [&](const Piece & piece, AABB box) {
			box.fix_if_needed();   // (no flat boxes: AABB::fix_if_needed, as every box of a triangle)
			if (piece.clipped) for (int d = 0; d < 3; d++) { box.min[d] = std::nextafter(box.min[d], -INFINITY); box.max[d] = std::nextafter(box.max[d], INFINITY); }
			source.push_back(int(t));
			boxes.insert(boxes.end(), { box.min.x, box.min.y, box.min.z, box.max.x, box.max.y, box.max.z });
			made++;
		};
		while (!work.empty()) {
			Piece piece = work.back(); work.pop_back();
			AABB box = AABB::create_empty();
			for (int i = 0; i < piece.n; i++) box.expand(piece.v[i]);
			int axis = 0; float extent = 0.0f;
			for (int d = 0; d < 3; d++) if (box.max[d] - box.min[d] > extent) { extent = box.max[d] - box.min[d]; axis = d; }
			if (!(extent > limit) || made + int(work.size()) + 2 > max_pieces || piece.n + 1 > MAX_VERTICES) { emit(piece, box); continue; }
			const float plane = 0.5f * (box.min[axis] + box.max[axis]);
			Piece below, above;
			below.n = clip_polygon(piece.v, piece.n, axis, plane, false, below.v); below.clipped = true;
			above.n = clip_polygon(piece.v, piece.n, axis, plane, true,  above.v); above.clipped = true;
			if (below.n < 3 || above.n < 3) { emit(piece, box); continue; }   // a degenerate cut (the polygon lies in the plane): the piece stays whole
			work.push_back(above); work.push_back(below);
		}
	}
}
