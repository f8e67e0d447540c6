// Wavefront path tracing integrator, host side (reference: Src/Renderer/Integrators/
// Pathtracer.h:146-267, Pathtracer.cpp). The launch loop itself lives behind
// rt_render_sample(); this class packs materials/media, builds the light CDFs and keeps
// the reference's update()/render() protocol.
#pragma once
#include "Integrator.h"

struct Pathtracer final : Integrator {
	// Light sampling tables (reference: Pathtracer.cpp:384-534)
	std::vector<int>   light_triangle_indices;
	std::vector<float> light_triangle_cumulative_probability;
	std::vector<float> light_mesh_cumulative_probability;
	std::vector<int>   light_mesh_triangle_span;      // {first, last} per light mesh
	std::vector<int>   light_mesh_transform_indices;  // TLAS-order mesh id per light mesh
	float              lights_total_weight = 0.0f;

	// The SVGFData pair last handed to the device (reference: Pathtracer.cpp:707-717)
	std::vector<float> svgf_matrices = std::vector<float>(32, 0.0f);

	Pathtracer(int width, int height, Scene & scene, int device_ordinal = 0) : Integrator(scene, device_ordinal) {
		gpu_init(width, height);
	}
	// Source compatibility with `Pathtracer(frame_buffer_handle, width, height, scene)` (Main.cpp:68)
	Pathtracer(unsigned /*frame_buffer_handle*/, int width, int height, Scene & scene) : Pathtracer(width, height, scene, 0) { }

	void gpu_init(int width, int height) override;
	void gpu_free() override;

	void resize_init(int width, int height) override;
	void resize_free() override;

	void update(float delta) override;
	void render() override;
	// `count` samples per pixel in one wavefront (rt_render_samples): the same image as calling
	// update(); render(); `count` times. sample_index ends on the last sample rendered, so the next
	// update() continues the progression where the reference's loop would be.
	void render_samples(int count);

	void calc_light_power();
	void geometry_was_rebuilt() override { if (scene.has_lights) calc_light_power(); }   // light_triangle_indices name device triangles
	void calc_light_mesh_weights();

};
