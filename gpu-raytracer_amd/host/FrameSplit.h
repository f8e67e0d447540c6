// FrameSplit -- one frame rendered by several GPUs of a node from ONE C++ process (the reference is
// single-GPU: Main.cpp:68 constructs one Pathtracer). Every GPU gets an Integrator of its own with a full scene
// replica and renders the row tiles rt_set_pixel_tiles deals it round-robin (8 rows each: Sponza's sky rows end
// early, contiguous blocks would leave the floor rows' owners as stragglers); paths of different pixels are
// independent and the RNG is keyed on the pixel index (Sampling.h:46,71-72), so the split image is bit-identical
// to a single GPU's. After every render() ONE grouped all-gather over RCCL (rt_all_gather_framebuffers:
// ncclGroupStart, an ncclAllGather per GPU on its context's stream, ncclGroupEnd) makes every rank's final image
// the whole frame -- no Python, no torch.distributed. SVGF frames: each rank path-traces its tiles unfiltered, the
// filter's inputs are gathered (80 B per pixel) and every rank filters the whole frame (the reprojection reads
// from anywhere in the previous frame).
// The same protocol as the class it wraps: update(delta); render(); ... save_image(). Ranks that share a device
// ordinal (tests on one GPU) exchange by peer copies, see rt_comm_init_all.
// One submitting thread per device: render() / render_samples() hand every rank's launches to that rank's own
// worker thread -- an iteration of the wavefront is ~6 kernel launches per rank, and one host thread enqueueing them for 8 GPUs
// in turn would make every GPU wait for the seven others' launch overhead. update() stays on the calling thread (the ranks
// share the host scene and its camera), the exchange is one grouped call after the workers have enqueued their frames.
#pragma once
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "Pathtracer.h"

struct FrameSplit {
	static constexpr int TILE_ROWS = 8;

	Scene & scene;
	std::vector<std::unique_ptr<Pathtracer>> ranks;   // ranks[r] renders tiles r, r + world, r + 2 world, ...
	int tile_pixels = 0;

	FrameSplit(int width, int height, Scene & scene, const std::vector<int> & device_ordinals);
	~FrameSplit();

	int world() const { return int(ranks.size()); }
	int sample_index() const { return ranks.front()->sample_index; }
	Pathtracer & front() { return *ranks.front(); }

	void update(float delta);
	void render();                   // one sample per pixel on every rank's tiles, then the exchange
	void render_samples(int count);  // `count` samples as one wavefront per rank (plain path tracing), then the exchange

	std::vector<float> read_framebuffer() { return ranks.front()->read_framebuffer(); }
	void save_image(const std::string & filename) { ranks.front()->save_image(filename); }

	int submitting_threads() const { return int(workers.size()); }   // 0 for a single rank (the caller's thread submits)

private:
	struct Worker {
		std::thread thread;
		std::mutex mutex; std::condition_variable wake, done;
		std::function<void()> job; bool busy = false, stop = false;
		std::exception_ptr error;
	};
	std::vector<std::unique_ptr<Worker>> workers;   // workers[r] submits for ranks[r]
	std::vector<rt_context *> contexts;
	void on_every_rank(const std::function<void(int)> & job);   // job(r) on rank r's thread, all ranks concurrently; rethrows the first failure
	void exchange();
	void check(rt_context * ctx, int status) const;
};
