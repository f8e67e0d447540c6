#include "AO.h"

// reference: AO::resize_init (AO.cpp:94-126) -- only RADIANCE is enabled by default, NORMAL and
// POSITION are opt-in (render_gui check boxes, AO.cpp:207-210).
void AO::resize_init(int width, int height) {
	screen_width  = width;
	screen_height = height;
	screen_pitch  = Math::round_up(width, 32);
	pixel_count   = width * height;

	if (ctx) check(rt_resize(ctx, width, height));
	aov_enable(AOVType::RADIANCE);

	scene.camera.resize(width, height);
	invalidated_camera = true;
	sample_index = 0;
}

// reference: AO::update (AO.cpp:140-146)
void AO::update(float delta) {
	if (invalidated_scene) sample_index = 0;
	Integrator::update(delta);
}

void AO::render() {
	require_device();
	check(rt_render_ao_sample(ctx, sample_index, ao_radius));
	if (pixel_query_status == PixelQueryStatus::PENDING) pixel_query_status = PixelQueryStatus::OUTPUT_READY; // AO.cpp:196-198
}
