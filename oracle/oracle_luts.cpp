// oracle_luts.cpp -- CPU restatement of the Kulla-Conty LUT integration kernels
// (CUDA/KullaConty.h:83-240). TEST INFRASTRUCTURE ONLY.
#include "oracle.h"
#include "oracle_shading.h"

#include <omp.h>

static inline float online_average(float avg, float sample, int n) { // Util.h:197-203
	return n == 0 ? sample : avg + (sample - avg) / float(n);
}

extern "C" {

void oracle_integrate_dielectric_cells(const oracle_scene * scene, int entering, int num_samples, int first_cell, int cell_count, float * out, int threads) {
	if (threads <= 0) threads = oracle_default_threads();
	#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
	for (int k = 0; k < cell_count; k++) {
		int thread_index = first_cell + k;
		int i = thread_index % 16, r = (thread_index / 16) % 16, c = (thread_index / 256) % 16;
		float ior = remap((float(i) + 0.5f) / 16.0f, 0.0f, 1.0f, LUT_DIELECTRIC_MIN_IOR, LUT_DIELECTRIC_MAX_IOR);
		float eta = entering ? 1.0f / ior : ior;
		float linear_roughness = (float(r) + 0.5f) / 16.0f;
		float cos_theta = (float(c) + 0.5f) / 16.0f;
		float sin_theta = safe_sqrt(1.0f - square(cos_theta));
		float3 omega_i = make_float3(sin_theta, 0.0f, cos_theta);
		float ax = roughness_to_alpha(linear_roughness), ay = ax;

		float avg = 0.0f;
		for (int s = 0; s < num_samples; s++) {
			float  rand_fresnel = oracle_random_sample(*scene, DIM_BSDF_0, uint32_t(thread_index), 0, uint32_t(s)).y;
			float2 rand_brdf    = oracle_random_sample(*scene, DIM_BSDF_1, uint32_t(thread_index), 0, uint32_t(s));
			float3 omega_m = sample_visible_normals_ggx(omega_i, ax, ay, rand_brdf.x, rand_brdf.y);
			float F = fresnel_dielectric(abs_dot(omega_i, omega_m), eta);
			bool reflected = rand_fresnel < F;
			float3 omega_o = reflected ? reflect_direction(omega_i, omega_m) : refract_direction(omega_i, omega_m, eta);
			float weight = 0.0f;
			if (!(reflected ^ (omega_o.z >= 0.0f))) {
				float D  = ggx_D(omega_m, ax, ay);
				float G1 = ggx_G1(omega_i, ax, ay);
				float G2 = ggx_G2(omega_o, omega_i, omega_m, ax, ay);
				float i_dot_m = abs_dot(omega_i, omega_m), o_dot_m = abs_dot(omega_o, omega_m);
				float pdf = reflected ? F * G1 * D / (4.0f * omega_i.z)
				                      : (1.0f - F) * G1 * D * i_dot_m * o_dot_m / (omega_i.z * square(eta * i_dot_m + o_dot_m));
				weight = pdf_is_valid(pdf) ? G2 / G1 : 0.0f;
			}
			avg = online_average(avg, weight, s + 1);
		}
		out[k] = avg;
	}
}

void oracle_integrate_conductor_cells(const oracle_scene * scene, int num_samples, int first_cell, int cell_count, float * out, int threads) {
	if (threads <= 0) threads = oracle_default_threads();
	#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
	for (int k = 0; k < cell_count; k++) {
		int thread_index = first_cell + k;
		int r = thread_index % 32, c = (thread_index / 32) % 32;
		float linear_roughness = (float(r) + 0.5f) / 32.0f;
		float cos_theta = (float(c) + 0.5f) / 32.0f;
		float sin_theta = safe_sqrt(1.0f - square(cos_theta));
		float3 omega_i = make_float3(sin_theta, 0.0f, cos_theta);
		float ax = roughness_to_alpha(linear_roughness), ay = ax;
		float avg = 0.0f;
		for (int s = 0; s < num_samples; s++) {
			float2 rand_brdf = oracle_random_sample(*scene, DIM_BSDF_0, uint32_t(thread_index), 0, uint32_t(s));
			float3 omega_m = sample_visible_normals_ggx(omega_i, ax, ay, rand_brdf.x, rand_brdf.y);
			float3 omega_o = reflect_direction(omega_i, omega_m);
			float weight = 0.0f;
			if (!(dot(omega_o, omega_m) <= 0.0f || omega_o.z <= 0.0f)) {
				float D  = ggx_D(omega_m, ax, ay);
				float G1 = ggx_G1(omega_i, ax, ay);
				float G2 = ggx_G2(omega_o, omega_i, omega_m, ax, ay);
				float pdf = G1 * D / (4.0f * omega_i.z);
				weight = pdf_is_valid(pdf) ? G2 / G1 : 0.0f;
			}
			avg = online_average(avg, weight, s + 1);
		}
		out[k] = avg;
	}
}

void oracle_average_dielectric(const float * directional, float * out) {
	for (int t = 0; t < 256; t++) {
		int i = t % 16, r = (t / 16) % 16;
		float avg = 0.0f;
		for (int c = 0; c < 16; c++) {
			float cos_theta = (float(c) + 0.5f) / 16.0f;
			avg = online_average(avg, directional[i + r * 16 + c * 256] * cos_theta, c + 1);
		}
		out[t] = 2.0f * avg;
	}
}

void oracle_average_conductor(const float * directional, float * out) {
	for (int r = 0; r < 32; r++) {
		float avg = 0.0f;
		for (int c = 0; c < 32; c++) {
			float cos_theta = (float(c) + 0.5f) / 32.0f;
			avg = online_average(avg, directional[r + c * 32] * cos_theta, c + 1);
		}
		out[r] = 2.0f * avg;
	}
}

} // extern "C"
