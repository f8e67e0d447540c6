// Binary BVH variants besides the plain SAH tree of BVH.cpp:
//   * SBVHBuilder   -- SAH object splits + binned spatial splits with reference unsplitting
//                      (Stich et al. 2009), cpu_config.bvh_type = SBVH
//   * BVHCollapser  -- merges subtrees into multi-triangle leaves where the SAH says a leaf is
//                      cheaper; the reference applies it to every file-loaded mesh unless the
//                      BVH8 is used (Assets/AssetManager.cpp:85-87)
// Both must produce the reference builder's bytes (node order, boxes, index lists), so every
// comparison, tie-break and rounding step follows Builders/SBVHBuilder.cpp:13-366,
// Builders/BVHPartitions.cpp:78-282 and BVHCollapser.cpp:10-114; tests/test_bvh_build.py checks
// the output against the verbatim reference build in oracle/_ref.
#include "BVH.h"
#include "Config.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

constexpr int   BIN_COUNT   = 256;    // BVHPartitions.h:42
constexpr float BOUNDS_SLOP = 0.001f; // binning range is the node box widened by this much

struct Bin {
	AABB box = AABB::create_empty();
	int  entries = 0; // references whose first bin this is
	int  exits   = 0; // references whose last bin this is
};

struct SpatialSplit {
	int   bin   = -1;       // references starting left of this bin boundary go left
	int   axis  = -1;
	float cost  = INFINITY;
	float plane = NAN;
	AABB  left, right;
	int   count_left = 0, count_right = 0;
};

// The triangle's corners ordered along `axis` by the same three compare-exchanges as the
// reference (equal keys keep their order, which decides which corner the clipper sees first).
inline void sorted_corners(const Triangle & triangle, int axis, Vector3 corner[3]) {
	corner[0] = triangle.position_0;
	corner[1] = triangle.position_1;
	corner[2] = triangle.position_2;
	if (corner[0][axis] > corner[1][axis]) std::swap(corner[0], corner[1]);
	if (corner[1][axis] > corner[2][axis]) std::swap(corner[1], corner[2]);
	if (corner[0][axis] > corner[1][axis]) std::swap(corner[0], corner[1]);
}

// Appends the points where the triangle's edges meet the axis-aligned plane; an edge that lies
// in the plane contributes both of its ends (BVHPartitions.cpp:78-101).
inline void clip_edges(const Vector3 corner[3], int axis, float plane, Vector3 * points, int & point_count) {
	for (int i = 0; i < 3; i++) {
		float a = corner[i][axis];
		for (int j = i + 1; j < 3; j++) {
			float b = corner[j][axis];
			if (!(a <= plane && plane <= b)) continue;
			float span = b - a;
			if (span == 0) {
				points[point_count++] = corner[i];
				points[point_count++] = corner[j];
			} else {
				float t = (plane - a) / span;
				points[point_count++] = (1.0f - t) * corner[i] + t * corner[j];
			}
		}
	}
}

inline int bin_of(float x, float range_min, float inv_range) {
	return int(float(BIN_COUNT) * ((x - range_min) * inv_range));
}

void sort_by_centre(std::vector<SBVHBuilder::Ref> & refs, int axis, std::vector<std::pair<unsigned, int>> & keys, std::vector<SBVHBuilder::Ref> & tmp) {
	keys.resize(refs.size());
	for (size_t i = 0; i < refs.size(); i++) keys[i] = { bvh_float_sort_key(refs[i].box.get_center()[axis]), int(i) };
	std::stable_sort(keys.begin(), keys.end(), [](const std::pair<unsigned, int> & l, const std::pair<unsigned, int> & r) { return l.first < r.first; });
	tmp.resize(refs.size());
	for (size_t i = 0; i < refs.size(); i++) tmp[i] = refs[keys[i].second];
	refs.swap(tmp);
}

struct SpatialBuild {
	SBVHBuilder & b;
	const std::vector<Triangle> & triangles;
	float inv_root_area;

	std::vector<std::pair<unsigned, int>> sort_keys;
	std::vector<SBVHBuilder::Ref>         sort_tmp;

	// Binned spatial split search over all three axes (BVHPartitions.cpp:103-282).
	SpatialSplit find_spatial_split(int first, int count, const AABB & bounds) const {
		SpatialSplit best;
		std::vector<Bin> bins(BIN_COUNT);
		AABB  grown_left [BIN_COUNT], grown_right[BIN_COUNT + 1];
		int   count_left [BIN_COUNT], count_right[BIN_COUNT + 1];
		float bin_cost   [BIN_COUNT];

		for (int axis = 0; axis < 3; axis++) {
			float range_min = bounds.min[axis] - BOUNDS_SLOP;
			float range_max = bounds.max[axis] + BOUNDS_SLOP;
			float bin_width = (range_max - range_min) / BIN_COUNT;
			float inv_range = 1.0f / (range_max - range_min);

			std::fill(bins.begin(), bins.end(), Bin());

			for (int i = first; i < first + count; i++) {
				const SBVHBuilder::Ref & ref = b.refs[axis][i];
				Vector3 corner[3];
				sorted_corners(triangles[ref.triangle], axis, corner);

				float lo = ref.box.min[axis];
				float hi = ref.box.max[axis];
				int bin_lo = std::min(std::max(bin_of(lo, range_min, inv_range), 0), BIN_COUNT - 1);
				int bin_hi = std::min(std::max(bin_of(hi, range_min, inv_range), 0), BIN_COUNT - 1);
				bins[bin_lo].entries++;
				bins[bin_hi].exits++;

				for (int k = bin_lo; k <= bin_hi; k++) {
					float plane_l = range_min + float(k) * bin_width;
					float plane_r = plane_l + bin_width;
					if (lo >= plane_r || hi <= plane_l) continue; // touches the slab with zero extent only

					AABB clipped;
					if (lo >= plane_l && hi <= plane_r) {
						clipped = ref.box; // entirely inside this slab
					} else {
						Vector3 points[12];
						int     point_count = 0;
						if (lo <= plane_l && plane_l <= hi) clip_edges(corner, axis, plane_l, points, point_count);
						if (lo <= plane_r && plane_r <= hi) clip_edges(corner, axis, plane_r, points, point_count);
						if (point_count == 0) {
							clipped = ref.box;
						} else {
							clipped = AABB::from_points(points, point_count);
							if (corner[1][axis] >= plane_l && corner[1][axis] <  plane_r) clipped.expand(corner[1]);
							if (corner[2][axis] <= plane_r && corner[2][axis] <= hi)      clipped.expand(corner[2]);
							if (corner[0][axis] >= plane_l && corner[0][axis] >= lo)      clipped.expand(corner[0]);
							clipped = AABB::overlap(clipped, ref.box);
						}
					}
					Bin & bin = bins[k];
					bin.box.expand(clipped);
					bin.box = AABB::overlap(bin.box, bounds);
					bin.box.fix_if_needed();
				}
			}

			// Prefix (left) and suffix (right) sweeps over the bin boundaries. A boundary with an
			// empty side has the cost inf * 0 = NaN, which no comparison below selects.
			grown_left[0] = AABB::create_empty();
			count_left[0] = 0;
			for (int k = 1; k < BIN_COUNT; k++) {
				grown_left[k] = grown_left[k - 1];
				grown_left[k].expand(bins[k - 1].box);
				count_left[k] = count_left[k - 1] + bins[k - 1].entries;
				bin_cost[k] = count_left[k] < count ? grown_left[k].surface_area() * float(count_left[k]) : INFINITY;
			}
			grown_right[BIN_COUNT] = AABB::create_empty();
			count_right[BIN_COUNT] = 0;
			for (int k = BIN_COUNT - 1; k > 0; k--) {
				grown_right[k] = grown_right[k + 1];
				grown_right[k].expand(bins[k].box);
				count_right[k] = count_right[k + 1] + bins[k].exits;
				if (count_right[k] < count) bin_cost[k] += grown_right[k].surface_area() * float(count_right[k]);
				else                        bin_cost[k]  = INFINITY;
			}
			for (int k = 1; k < BIN_COUNT; k++) {
				if (bin_cost[k] < best.cost) { // strict: the first minimum over (axis, bin) wins
					best.cost  = bin_cost[k];
					best.bin   = k;
					best.axis  = axis;
					best.plane = range_min + bin_width * float(k);
					best.left  = grown_left [k];
					best.right = grown_right[k];
					best.count_left  = count_left [k];
					best.count_right = count_right[k];
				}
			}
		}
		return best;
	}

	// Distributes the node's references over the two sides of a spatial split; a straddling
	// reference is either clipped into both sides or, when the SAH prefers it, kept whole on one
	// side ("unsplitting", SBVHBuilder.cpp:173-300). Grows split.left / split.right as it goes.
	void apply_spatial_split(SpatialSplit & split, int first, int count, const AABB & node_box, std::vector<SBVHBuilder::Ref> side[2]) const {
		int   axis = split.axis;
		float n_left  = float(split.count_left);
		float n_right = float(split.count_right);
		float range_min = node_box.min[axis] - BOUNDS_SLOP;
		float range_max = node_box.max[axis] + BOUNDS_SLOP;
		float inv_range = 1.0f / (range_max - range_min);

		for (int i = first; i < first + count; i++) {
			const SBVHBuilder::Ref & ref = b.refs[axis][i];
			Vector3 corner[3];
			sorted_corners(triangles[ref.triangle], axis, corner);

			// unclamped here, unlike the search above
			bool goes_left  = bin_of(ref.box.min[axis], range_min, inv_range) <  split.bin;
			bool goes_right = bin_of(ref.box.max[axis], range_min, inv_range) >= split.bin;

			if (goes_left && goes_right) {
				AABB whole_left  = split.left;  whole_left .expand(ref.box);
				AABB whole_right = split.right; whole_right.expand(ref.box);
				float area_left  = split.left .surface_area();
				float area_right = split.right.surface_area();

				float cost_both  = area_left                  *  n_left         + area_right                  *  n_right;
				float cost_left  = whole_left.surface_area()  *  n_left         + area_right                  * (n_right - 1.0f);
				float cost_right = area_left                  * (n_left - 1.0f) + whole_right.surface_area()  *  n_right;

				bool only_right = cost_left < cost_both ? cost_right < cost_left : cost_right < cost_both;
				bool only_left  = cost_left < cost_both && !only_right;
				if (only_right) {
					goes_left = false;
					n_left -= 1.0f;
					split.right.expand(ref.box);
				} else if (only_left) {
					goes_right = false;
					n_right -= 1.0f;
					split.left.expand(ref.box);
				}
			}

			if (goes_left && goes_right) {
				Vector3 points[6];
				int     point_count = 0;
				clip_edges(corner, axis, split.plane, points, point_count);

				AABB part[2];
				part[0] = part[1] = AABB::from_points(points, point_count);
				for (int c = 0; c < 3; c++) part[corner[c][axis] < split.plane ? 0 : 1].expand(corner[c]);
				for (int s = 0; s < 2; s++) {
					part[s].min = Vector3::max(part[s].min, ref.box.min);
					part[s].max = Vector3::min(part[s].max, ref.box.max);
					part[s].fix_if_needed();
				}
				split.left .expand(part[0]);
				split.right.expand(part[1]);
				side[0].push_back({ ref.triangle, part[0] });
				side[1].push_back({ ref.triangle, part[1] });
			} else if (goes_left) {
				split.left.expand(ref.box);
				side[0].push_back(ref);
			} else {
				split.right.expand(ref.box);
				side[1].push_back(ref);
			}
		}
	}

	void store(int first, const std::vector<SBVHBuilder::Ref> lists[3]) {
		size_t n = lists[0].size();
		for (int axis = 0; axis < 3; axis++) {
			if (b.refs[axis].size() < first + n) b.refs[axis].resize(first + n);
			std::copy(lists[axis].begin(), lists[axis].end(), b.refs[axis].begin() + first);
		}
	}

	// Returns the number of leaves (= index entries) the subtree produced; the right subtree's
	// references are only written into the shared lists once that number is known for the left.
	int build_node(int node_index, int first, int count) {
		if (count == 1) {
			b.bvh.nodes[node_index].first = first;
			b.bvh.nodes[node_index].count = 1;
			return 1;
		}
		BVHObjectSplit object = bvh_find_object_split([&](int axis, int i) -> const AABB & { return b.refs[axis][i].box; }, first, count, b.sweep_cost.data());

		// Spatial splits are only tried when the object split's children overlap by more than
		// alpha of the root's area.
		AABB  shared = AABB::overlap(object.left, object.right);
		float ratio  = (shared.is_valid() ? shared.surface_area() : 0.0f) * inv_root_area;
		AABB  node_box = b.bvh.nodes[node_index].aabb;
		SpatialSplit spatial;
		if (ratio > cpu_config.sbvh_alpha) spatial = find_spatial_split(first, count, node_box);

		int child = int(b.bvh.nodes.size());
		b.bvh.nodes.resize(b.bvh.nodes.size() + 2);
		memset((void *)&b.bvh.nodes[child], 0, 2 * sizeof(BVHNode2));
		b.bvh.nodes[node_index].left  = child;
		b.bvh.nodes[node_index].count = 0;

		std::vector<SBVHBuilder::Ref> left[3], right[3];
		AABB box_left, box_right;

		if (object.cost <= spatial.cost) {
			b.bvh.nodes[node_index].axis = unsigned(object.axis);
			const std::vector<SBVHBuilder::Ref> & order = b.refs[object.axis];
			for (int i = first;        i < object.index;  i++) b.goes_left[order[i].triangle] = 1;
			for (int i = object.index; i < first + count; i++) b.goes_left[order[i].triangle] = 0;
			for (int axis = 0; axis < 3; axis++) {
				left [axis].reserve(object.index - first);
				right[axis].reserve(first + count - object.index);
				for (int i = first; i < first + count; i++) {
					const SBVHBuilder::Ref & ref = b.refs[axis][i];
					(b.goes_left[ref.triangle] ? left : right)[axis].push_back(ref);
				}
			}
			box_left  = object.left;
			box_right = object.right;
		} else {
			b.bvh.nodes[node_index].axis = unsigned(spatial.axis);
			std::vector<SBVHBuilder::Ref> side[2];
			side[0].reserve(count);
			side[1].reserve(count);
			apply_spatial_split(spatial, first, count, node_box, side);
			for (int axis = 0; axis < 3; axis++) {
				left [axis] = side[0];
				right[axis] = side[1];
				sort_by_centre(left [axis], axis, sort_keys, sort_tmp);
				sort_by_centre(right[axis], axis, sort_keys, sort_tmp);
			}
			box_left  = spatial.left;
			box_right = spatial.right;
		}
		b.bvh.nodes[child    ].aabb = box_left;
		b.bvh.nodes[child + 1].aabb = box_right;

		int n_left  = int(left [0].size());
		int n_right = int(right[0].size());
		if (n_left == 0 || n_right == 0 || n_left == count + n_right || n_right == count + n_left) {
			throw std::runtime_error("SBVH: a split did not separate the references");
		}

		store(first, left);
		for (int axis = 0; axis < 3; axis++) std::vector<SBVHBuilder::Ref>().swap(left[axis]);
		int leaves_left = build_node(child, first, n_left);

		store(first + leaves_left, right);
		for (int axis = 0; axis < 3; axis++) std::vector<SBVHBuilder::Ref>().swap(right[axis]);
		int leaves_right = build_node(child + 1, first + leaves_left, n_right);

		return leaves_left + leaves_right;
	}
};

} // namespace

void SBVHBuilder::build(const std::vector<Triangle> & triangles) {
	size_t n = triangles.size();
	sweep_cost.resize(n);
	goes_left .resize(n);

	AABB root = AABB::create_empty();
	refs[0].resize(n);
	for (size_t i = 0; i < n; i++) {
		refs[0][i] = { int(i), triangles[i].get_aabb() };
		root.expand(refs[0][i].box);
	}
	refs[1] = refs[0];
	refs[2] = refs[0];

	SpatialBuild build { *this, triangles, 1.0f / root.surface_area() };
	for (int axis = 0; axis < 3; axis++) sort_by_centre(refs[axis], axis, build.sort_keys, build.sort_tmp);

	bvh.indices.clear();
	bvh.nodes.clear();
	bvh.nodes.reserve(std::max<size_t>(2 * n, 2));
	bvh.nodes.resize(2); // root + the dummy that keeps sibling pairs 64-byte aligned
	memset((void *)bvh.nodes.data(), 0, 2 * sizeof(BVHNode2));
	bvh.nodes[0].aabb = root;

	int leaf_count = build.build_node(0, 0, int(n));

	bvh.indices.resize(leaf_count);
	for (int i = 0; i < leaf_count; i++) bvh.indices[i] = refs[0][i].triangle;
}

// ---------------------------------------------------------------------------------------------
// Leaf collapse
// ---------------------------------------------------------------------------------------------

namespace {
struct Collapse {
	const BVH2 & in;
	BVH2       & out;
	std::vector<char> merge; // per input node: turn the whole subtree into one leaf

	struct Cost { int primitives; float sah; };

	// Bottom-up: a subtree becomes one leaf when that is cheaper than keeping its (already
	// optimally collapsed) children (BVHCollapser.cpp:10-38).
	Cost decide(int node_index) {
		const BVHNode2 & node = in.nodes[node_index];
		if (node.is_leaf()) return { int(node.count), float(node.count) * cpu_config.sah_cost_leaf };

		Cost l = decide(node.left);
		Cost r = decide(node.left + 1);
		int primitives = l.primitives + r.primitives;

		float as_leaf = cpu_config.sah_cost_leaf * float(primitives);
		float as_node = cpu_config.sah_cost_node + (
			in.nodes[node.left    ].aabb.surface_area() * l.sah +
			in.nodes[node.left + 1].aabb.surface_area() * r.sah) / node.aabb.surface_area();
		if (as_leaf < as_node) {
			merge[node_index] = 1;
			return { primitives, as_leaf };
		}
		return { primitives, as_node };
	}

	int append_leaves(int node_index) {
		const BVHNode2 & node = in.nodes[node_index];
		if (node.is_leaf()) {
			out.indices.insert(out.indices.end(), in.indices.begin() + node.first, in.indices.begin() + node.first + node.count);
			return int(node.count);
		}
		return append_leaves(node.left) + append_leaves(node.left + 1);
	}

	void emit(int node_index, int out_index) {
		const BVHNode2 & node = in.nodes[node_index];
		out.nodes[out_index].aabb  = node.aabb;
		out.nodes[out_index].count = node.count;
		out.nodes[out_index].axis  = node.axis;

		if (node.is_leaf() || merge[node_index]) {
			int first = int(out.indices.size());
			int n = append_leaves(node_index);
			out.nodes[out_index].first = first;
			out.nodes[out_index].count = unsigned(n);
			return;
		}
		int child = int(out.nodes.size());
		out.nodes.resize(out.nodes.size() + 2);
		memset((void *)&out.nodes[child], 0, 2 * sizeof(BVHNode2));
		out.nodes[out_index].left = child;
		emit(node.left,     child);
		emit(node.left + 1, child + 1);
	}
};
}

void BVHCollapser::collapse(BVH2 & bvh) {
	BVH2 collapsed;
	collapsed.indices.reserve(bvh.indices.size());
	collapsed.nodes  .reserve(bvh.nodes  .size());
	collapsed.nodes.resize(2);
	memset((void *)collapsed.nodes.data(), 0, 2 * sizeof(BVHNode2));

	Collapse pass { bvh, collapsed, std::vector<char>(bvh.nodes.size(), 0) };
	pass.decide(0);
	pass.emit(0, 0);

	bvh = std::move(collapsed);
}
