// Pinhole / thin-lens camera (reference: Src/Renderer/Camera.h, Camera.cpp:20-96).
// The interactive key handling of the reference is not part of the render path;
// `moved` is set by callers that change position/rotation.
#pragma once
#include "Math.h"

struct Camera {
	Vector3    position;
	Quaternion rotation;

	float fov;
	float pixel_spread_angle;

	float aperture_radius =  0.0f;
	float focal_distance  = 10.0f;

	float near_plane, far_plane;
	float screen_width = 900.0f, screen_height = 600.0f;

	Vector3 bottom_left_corner, bottom_left_corner_rotated;
	Vector3 x_axis, x_axis_rotated;
	Vector3 y_axis, y_axis_rotated;

	Matrix4 projection;
	Matrix4 view_projection;
	Matrix4 view_projection_prev;

	bool moved = false;

	Camera(float fov, float near_plane = 0.1f, float far_plane = 300.0f) : near_plane(near_plane), far_plane(far_plane) { set_fov(fov); }

	void resize(int width, int height) { screen_width = float(width); screen_height = float(height); recalibrate(); }
	void set_fov(float fov) { this->fov = fov; recalibrate(); }

	void update(float delta) {
		(void)delta;
		bottom_left_corner_rotated = rotation * bottom_left_corner;
		x_axis_rotated             = rotation * x_axis;
		y_axis_rotated             = rotation * y_axis;

		view_projection_prev = view_projection;
		view_projection = projection * Matrix4::create_rotation(Quaternion::conjugate(rotation)) * Matrix4::create_translation(-position);
	}

private:
	void recalibrate() {
		float half_width  = 0.5f * screen_width;
		float half_height = 0.5f * screen_height;
		float tan_half_fov = tanf(0.5f * fov);
		float d = half_width / tan_half_fov; // distance to the image plane in pixels

		bottom_left_corner = Vector3(-half_width, -half_height, -d);
		x_axis = Vector3(1.0f, 0.0f, 0.0f);
		y_axis = Vector3(0.0f, 1.0f, 0.0f);

		projection = Matrix4::perspective(fov, half_height / half_width, near_plane, far_plane);

		// Eq. 30 of "Texture Level of Detail Strategies for Real-Time Ray Tracing"
		pixel_spread_angle = atanf(2.0f * tan_half_fov * (1.0f / screen_width));
	}
};
