#include "FrameSplit.h"

#include <stdexcept>
#include <string>

FrameSplit::FrameSplit(int width, int height, Scene & scene, const std::vector<int> & device_ordinals) : scene(scene) {
	if (device_ordinals.empty()) throw std::runtime_error("FrameSplit: no devices given");
	for (int device : device_ordinals) {
		ranks.push_back(std::make_unique<Pathtracer>(width, height, scene, device));
		if (!ranks.back()->ctx) throw std::runtime_error("FrameSplit: no device context for device " + std::to_string(device));
		contexts.push_back(ranks.back()->ctx);
	}
	tile_pixels = width * TILE_ROWS;
	if (world() > 1) {
		for (int r = 0; r < world(); r++) check(contexts[r], rt_set_pixel_tiles(contexts[r], tile_pixels, r, world()));
		check(contexts[0], rt_comm_init_all(contexts.data(), world()));
		for (int r = 0; r < world(); r++) {
			workers.push_back(std::make_unique<Worker>());
			Worker * w = workers.back().get();
			w->thread = std::thread([w] {
				std::unique_lock<std::mutex> lock(w->mutex);
				while (true) {
					w->wake.wait(lock, [w] { return w->busy || w->stop; });
					if (w->stop) return;
					try { w->job(); } catch (...) { w->error = std::current_exception(); }
					w->busy = false;
					w->done.notify_all();
				}
			});
		}
	}
}

FrameSplit::~FrameSplit() {
	for (auto & w : workers) {
		{ std::lock_guard<std::mutex> lock(w->mutex); w->stop = true; }
		w->wake.notify_all();
		if (w->thread.joinable()) w->thread.join();
	}
	for (rt_context * ctx : contexts) (void)rt_comm_destroy(ctx);
}

void FrameSplit::on_every_rank(const std::function<void(int)> & job) {
	if (workers.empty()) { for (int r = 0; r < world(); r++) job(r); return; }
	for (int r = 0; r < world(); r++) {
		Worker & w = *workers[r];
		{ std::lock_guard<std::mutex> lock(w.mutex); w.job = [&job, r] { job(r); }; w.error = nullptr; w.busy = true; }
		w.wake.notify_all();
	}
	std::exception_ptr first;
	for (int r = 0; r < world(); r++) {
		Worker & w = *workers[r];
		std::unique_lock<std::mutex> lock(w.mutex);
		w.done.wait(lock, [&w] { return !w.busy; });
		if (w.error && !first) first = w.error;
	}
	if (first) std::rethrow_exception(first);
}

void FrameSplit::check(rt_context * ctx, int status) const {
	if (status != RT_OK) throw std::runtime_error(std::string("FrameSplit: ") + rt_last_error(ctx));
}

void FrameSplit::update(float delta) {
	// the ranks share the host scene: rank 0's update() advances it (camera matrices, animated meshes: "current" becomes
	// "previous"), the others upload the state it left
	for (int r = 0; r < world(); r++) {
		ranks[r]->scene_advanced_by_another_integrator = r > 0;
		ranks[r]->update(delta);
	}
}

void FrameSplit::exchange() {
	if (world() == 1) return;
	check(contexts[0], rt_all_gather_framebuffers(contexts.data(), world()));
}

void FrameSplit::render() {
	if (world() > 1 && gpu_config.enable_svgf) {
		// path-trace the own tiles, gather what the filter reads, filter the whole frame everywhere
		on_every_rank([this](int r) { check(contexts[r], rt_render_sample_unfiltered(contexts[r], ranks[r]->sample_index)); });
		check(contexts[0], rt_all_gather_svgf_inputs(contexts.data(), world()));
		on_every_rank([this](int r) { check(contexts[r], rt_filter_frame(contexts[r], ranks[r]->sample_index)); });
		return;
	}
	on_every_rank([this](int r) { ranks[r]->render(); });
	exchange();
}

void FrameSplit::render_samples(int count) {
	if (world() > 1 && gpu_config.enable_svgf) { // SVGF frames feed each other's history: one at a time
		for (int i = 0; i < count; i++) { if (i) for (auto & rank : ranks) rank->sample_index++; render(); }
		return;
	}
	on_every_rank([this, count](int r) { ranks[r]->render_samples(count); });
	exchange();
}
