// oracle/oracle_texture_api.cpp -- TEST INFRASTRUCTURE ONLY.
// C entry points to the software texture unit of the oracle (oracle_shading.h), so that the
// reference's device code running on the CPU (oracle/ref/ref_cuda_harness.cpp) filters textures by the same rules
// as the oracle and the HIP kernels: the NVIDIA texture unit is the one part of the reference's device path that
// has no definition to restate.
#include "oracle_shading.h"

extern "C" {

void oracle_tex2d(const oracle_texture * tex, float s, float t, float out[4]) {
	float4 c = texture_get(*tex, s, t);
	out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
void oracle_tex2d_lod(const oracle_texture * tex, float s, float t, float lod, float out[4]) {
	float4 c = texture_get_lod(*tex, s, t, lod);
	out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
void oracle_tex2d_grad(const oracle_texture * tex, float s, float t, const float dx[2], const float dy[2], float out[4]) {
	float4 c = texture_get_grad(*tex, s, t, make_float2(dx[0], dx[1]), make_float2(dy[0], dy[1]));
	out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
float oracle_lut_1d(const float * lut, int nx, float s) { return lut_get_1d(lut, nx, s); }
float oracle_lut_2d(const float * lut, int nx, int ny, float s, float t) { return lut_get_2d(lut, nx, ny, s, t); }
float oracle_lut_3d(const float * lut, int nx, int ny, int nz, float s, float t, float r) { return lut_get_3d(lut, nx, ny, nz, s, t, r); }

// Clamp-addressed bilinear fetch of a float4 image at normalised coordinates (the sky: Sky.h:15)
void oracle_image_bilinear_clamp(const float * rgba, int width, int height, float u, float v, float out[4]) {
	int x0, x1, y0, y1; float fx, fy;
	clamp_taps(u, width, x0, x1, fx); clamp_taps(v, height, y0, y1, fy);
	auto texel = [&](int x, int y) { const float * p = rgba + (size_t(x) + size_t(y) * width) * 4; return make_float4(p[0], p[1], p[2], p[3]); };
	float4 c = lerp4(lerp4(texel(x0, y0), texel(x1, y0), fx), lerp4(texel(x0, y1), texel(x1, y1), fx), fy);
	out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}

} // extern "C"
