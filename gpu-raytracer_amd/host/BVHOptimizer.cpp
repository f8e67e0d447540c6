// Insertion-based optimisation of a binary BVH (Bittner, Hapala, Havran: "Fast Insertion-Based Optimization of
// Bounding Volume Hierarchies", 2013), the post-processing step behind cpu_config.enable_bvh_optimization / -O
// (reference: Src/BVH/BVHOptimizer.cpp:14-417). In batches, every inner node that has a grandparent is taken out
// of the tree -- the worst ones first, by an area-based inefficiency measure -- and its two children are put back
// wherever a branch-and-bound search finds the smallest increase in surface area; after five batches without
// progress the selection turns random, after ten it stops.
//
// While the selection is measure-driven the result is a pure function of the input tree, and that part is held
// byte-identical to the reference's optimiser compiled into oracle/_ref (tests/test_bvh_build.py): the two binary
// heaps below sift exactly like the reference's MinHeap, because equal costs are common and the pop order decides
// which node moves first. The reference seeds its random phase from the wall clock; here the seed is fixed, so the
// whole optimisation is reproducible.
#include "BVH.h"
#include "Config.h"

#include <chrono>
#include <cmath>

namespace {

// Array-backed binary heap; `before(a, b)` = a has to come out before b. Same insert / pop mechanics as the
// reference's Core/MinHeap.h (the new item sifts up while its parent is not before it; pop moves the last item to
// the root and sifts it down towards the child that comes first, left child preferred on ties).
template<typename T, typename Before>
struct Heap {
	std::vector<T> items;
	Before before;

	explicit Heap(Before before) : before(before) { }

	void push(T item) {
		size_t i = items.size();
		items.push_back(item);
		while (i > 0) {
			size_t parent = (i - 1) / 2;
			if (before(items[parent], items[i])) break;
			std::swap(items[parent], items[i]);
			i = parent;
		}
	}
	T pop() {
		T top = items[0];
		items[0] = items.back();
		items.pop_back();
		size_t i = 0;
		while (true) {
			size_t l = 2 * i + 1, r = 2 * i + 2, first = i;
			if (l < items.size() && before(items[l], items[first])) first = l;
			if (r < items.size() && before(items[r], items[first])) first = r;
			if (first == i) break;
			std::swap(items[i], items[first]);
			i = first;
		}
		return top;
	}
	bool empty() const { return items.empty(); }
};

inline AABB unite(const AABB & a, const AABB & b) { AABB r; r.min = Vector3::min(a.min, b.min); r.max = Vector3::max(a.max, b.max); return r; }

// PCG, as in the reference's Core/Random.h (only its bounded draw is needed, for selection sampling)
struct Random {
	uint64_t state;
	explicit Random(uint64_t seed) { state = (seed + 2891336453u) * 747796405u + 2891336453u; }
	uint32_t next() {
		uint32_t x = uint32_t(((state >> 18u) ^ state) >> 27u), r = uint32_t(state >> 59u);
		state = state * 6364136223846793005ull + 1;
		return (x >> r) | (x << ((~r + 1) & 31));
	}
	uint32_t below(uint32_t max) { // unbiased, Lemire's method
		uint32_t x = next();
		uint64_t m = uint64_t(x) * max;
		uint32_t l = uint32_t(m);
		if (l < max) {
			uint32_t t = ~max + 1;
			if (t >= max) { t -= max; if (t >= max) t %= max; }
			while (l < t) { x = next(); m = uint64_t(x) * max; l = uint32_t(m); }
		}
		return uint32_t(m >> 32);
	}
};

struct Optimiser {
	BVH2 & bvh;
	std::vector<int> parent;       // parent[i] of node slot i; INVALID for the root
	std::vector<int> originated;   // which node (by its slot at the start of the batch) now sits in slot i
	std::vector<int> displacement; // where the node that started the batch in slot i is now; INVALID once removed
	std::vector<int> batch;
	std::vector<float> measure;

	explicit Optimiser(BVH2 & bvh) : bvh(bvh), parent(bvh.nodes.size(), INVALID), originated(bvh.nodes.size()), displacement(bvh.nodes.size()), batch(bvh.nodes.size()), measure(bvh.nodes.size()) {
		for (size_t i = 2; i < bvh.nodes.size(); i++) {
			const BVHNode2 & node = bvh.nodes[i];
			if (!node.is_leaf()) parent[node.left] = parent[node.left + 1] = int(i);
		}
		const BVHNode2 & root = bvh.nodes[0];
		if (!root.is_leaf()) parent[root.left] = parent[root.left + 1] = 0;
	}

	float sah_cost() const {
		float leaves = 0.0f, inner = 0.0f;
		for (size_t i = 0; i < bvh.nodes.size(); i++) {
			if (i == 1) continue;
			const BVHNode2 & node = bvh.nodes[i];
			if (node.is_leaf()) leaves += node.aabb.surface_area() * node.count;
			else                inner  += node.aabb.surface_area();
		}
		return (cpu_config.sah_cost_node * inner + cpu_config.sah_cost_leaf * leaves) / bvh.nodes[0].aabb.surface_area();
	}

	bool is_candidate(size_t i) const { return !bvh.nodes[i].is_leaf() && parent[i] != 0; }

	// The `count` worst candidates by  (2A / (Al + Ar)) * (A / min(Al, Ar)) * A,  worst first
	void select_by_measure(int count) {
		auto worse = [this](int a, int b) { return measure[a] > measure[b]; };
		Heap<int, decltype(worse)> heap(worse);
		for (size_t i = 2; i < bvh.nodes.size(); i++) {
			if (!is_candidate(i)) continue;
			const BVHNode2 & node = bvh.nodes[i];
			float area = node.aabb.surface_area(), left = bvh.nodes[node.left].aabb.surface_area(), right = bvh.nodes[node.left + 1].aabb.surface_area();
			float cost_sum = 2.0f * area / (left + right);
			float cost_min = area / (left < right ? left : right);
			measure[i] = cost_sum * cost_min * area;
			heap.push(int(i));
		}
		for (int i = 0; i < count; i++) batch[i] = heap.pop();
	}

	// `count` candidates by selection sampling (Knuth), in slot order
	void select_at_random(int count, Random & rng) {
		std::vector<int> candidates;
		for (size_t i = 2; i < bvh.nodes.size(); i++) if (is_candidate(i)) candidates.push_back(int(i));
		int chosen = 0;
		for (size_t i = 0; i < candidates.size() && chosen < count; i++) {
			if (rng.below(uint32_t(candidates.size() - i)) < uint32_t(count - chosen)) batch[chosen++] = candidates[i];
		}
	}

	// Best node to pair the box with: the one whose replacement by a new parent of (node, box) adds the least
	// area over the whole path to the root. Branch and bound over induced cost, cheapest subtree first.
	int find_insertion_point(const AABB & box) const {
		struct Entry { int node; float induced; };
		auto cheaper = [](const Entry & a, const Entry & b) { return a.induced < b.induced; };
		Heap<Entry, decltype(cheaper)> queue(cheaper);
		queue.push({ 0, 0.0f });

		float box_area = box.surface_area();
		float best_cost = INFINITY;
		int   best = INVALID;
		while (!queue.empty()) {
			Entry e = queue.pop();
			if (e.induced + box_area >= best_cost) break; // nothing left can win
			const BVHNode2 & node = bvh.nodes[e.node];
			float cost = e.induced + unite(node.aabb, box).surface_area();
			if (cost < best_cost) { best_cost = cost; best = e.node; }
			if (!node.is_leaf()) {
				float below = cost - node.aabb.surface_area();
				if (below + box_area < best_cost) { queue.push({ node.left, below }); queue.push({ node.left + 1, below }); }
			}
		}
		return best;
	}

	void refit_upwards(int i) {
		for (; i != INVALID; i = parent[i]) {
			BVHNode2 & node = bvh.nodes[i];
			if (!node.is_leaf()) node.aabb = unite(bvh.nodes[node.left].aabb, bvh.nodes[node.left + 1].aabb);
		}
	}

	void adopt_children(int slot) { // the node in `slot` moved there: its children must point back at it
		const BVHNode2 & node = bvh.nodes[slot];
		if (!node.is_leaf()) parent[node.left] = parent[node.left + 1] = slot;
	}

	// A reinsertion invalidates the split axis stored in the new parent: pick the axis along which the children's
	// boxes differ most (last such axis on ties) and order them along it, as the traversal kernels expect.
	void choose_axis(int i) {
		BVHNode2 & node = bvh.nodes[i];
		int l = node.left, r = node.left + 1;
		int   axis = INVALID;
		float widest = 0.0f;
		for (int d = 0; d < 3; d++) {
			float distance = fabsf(bvh.nodes[l].aabb.min[d] - bvh.nodes[r].aabb.min[d]) + fabsf(bvh.nodes[l].aabb.max[d] - bvh.nodes[r].aabb.max[d]);
			if (distance >= widest) { widest = distance; axis = d; }
		}
		if (axis == INVALID) axis = 0; // NaN boxes only
		if (bvh.nodes[l].aabb.get_center()[axis] > bvh.nodes[r].aabb.get_center()[axis]) {
			std::swap(bvh.nodes[l], bvh.nodes[r]);
			displacement[originated[l]] = r;
			displacement[originated[r]] = l;
			adopt_children(l);
			adopt_children(r);
		}
		node.count = 0;
		node.axis  = unsigned(axis);
	}

	// Takes the inner node in `slot` out of the tree (its sibling moves up into the parent's slot) and reinserts its
	// two children, larger one first, into the two freed pairs of slots.
	void reinsert_children_of(int slot) {
		const BVHNode2 node = bvh.nodes[slot];
		int up = parent[slot];
		if (node.is_leaf() || up == 0 || up == INVALID) return;
		int up_up   = parent[up];
		int sibling = (slot & 1) ? slot - 1 : slot + 1;

		int moving[2] = { node.left, node.left + 1 };
		if (!(bvh.nodes[moving[0]].aabb.surface_area() > bvh.nodes[moving[1]].aabb.surface_area())) std::swap(moving[0], moving[1]);
		BVHNode2 moved[2] = { bvh.nodes[moving[0]], bvh.nodes[moving[1]] };
		int free_pair[2] = { slot & ~1, node.left };

		bvh.nodes[up] = bvh.nodes[sibling];
		parent[sibling] = up_up;
		displacement[originated[sibling]] = up;
		originated[up] = originated[sibling];
		adopt_children(up);
		displacement[originated[up]]   = INVALID; // (as the reference does: the sibling's entry is cleared again right away)
		displacement[originated[slot]] = INVALID;
		refit_upwards(up_up);

		for (int j = 0; j < 2; j++) {
			int pair = free_pair[j];
			int target = find_insertion_point(moved[j].aabb);
			if (target == INVALID) target = 0; // areas overflowed to infinity (coordinates beyond ~1e19): pair it with the root

			bvh.nodes[pair]     = bvh.nodes[target];
			bvh.nodes[pair + 1] = moved[j];
			parent[pair] = parent[pair + 1] = target;
			displacement[originated[target]]    = pair;
			displacement[originated[moving[j]]] = pair + 1;
			originated[pair]     = originated[target];
			originated[pair + 1] = originated[moving[j]];
			adopt_children(pair);
			adopt_children(pair + 1);

			bvh.nodes[target].left  = pair;
			bvh.nodes[target].count = 0;
			refit_upwards(target);
			choose_axis(target);
		}
	}

	void run_batch(int count) {
		for (size_t i = 0; i < bvh.nodes.size(); i++) originated[i] = displacement[i] = int(i);
		for (int i = 0; i < count; i++) {
			int slot = displacement[batch[i]];
			if (slot != INVALID) reinsert_children_of(slot); // else: removed by an earlier reinsertion of this batch
		}
	}
};

} // namespace

void BVHOptimizer::optimize(BVH2 & bvh) {
	// Candidates are inner nodes with a grandparent: none in a 7-node tree, one more for every further pair of nodes
	int candidates = std::max(int((long(bvh.nodes.size()) - 7) / 2), 0);
	if (candidates < 8) return; // too small to be worth it

	const int BATCHES_UNTIL_RANDOM = 5, BATCHES_UNTIL_STOP = 10;
	Optimiser optimiser(bvh);
	int batch_size = std::max(int(bvh.nodes.size() / 100), candidates);

	float best = optimiser.sah_cost();
	bool  by_measure = true;
	int   stalled = 0;
	Random rng(0x5eed5eedull);
	auto start = std::chrono::steady_clock::now();

	for (int batch_count = 0;; batch_count++) {
		if (by_measure) optimiser.select_by_measure(batch_size); else optimiser.select_at_random(batch_size, rng);
		optimiser.run_batch(batch_size);

		float cost = optimiser.sah_cost();
		if (cost < best) {
			best = cost;
			stalled = 0;
			by_measure = true;
		} else {
			stalled++;
			if (stalled == BATCHES_UNTIL_RANDOM) by_measure = false;
			if (stalled == BATCHES_UNTIL_STOP) break;
		}
		long elapsed_ms = long(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count());
		if (elapsed_ms >= cpu_config.bvh_optimizer_max_time || batch_count >= cpu_config.bvh_optimizer_max_num_batches) break;
	}
}
