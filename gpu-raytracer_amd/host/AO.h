// Ambient-occlusion integrator, host side (reference: Src/Renderer/Integrators/AO.h:73-112,
// AO.cpp). The launch loop -- generate, trace, kernel_ambient_occlusion, shadow trace, accumulate
// (AO.cpp:148-200) -- lives behind rt_render_ao_sample(); this class keeps the reference's
// constructor / update() / render() protocol and its one parameter, ao_radius.
#pragma once
#include "Integrator.h"

struct AO final : Integrator {
	float ao_radius = 1.0f; // AO.h:102

	AO(int width, int height, Scene & scene, int device_ordinal = 0) : Integrator(scene, device_ordinal) {
		gpu_init(width, height);
	}
	// Source compatibility with `AO(frame_buffer_handle, width, height, scene)` (Main.cpp:69)
	AO(unsigned /*frame_buffer_handle*/, int width, int height, Scene & scene) : AO(width, height, scene, 0) { }

	void resize_init(int width, int height) override;
	void resize_free() override { }

	void update(float delta) override;
	void render() override;
};
