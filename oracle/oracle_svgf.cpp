// oracle_svgf.cpp -- CPU restatement of the SVGF + TAA image pipeline. TEST INFRASTRUCTURE ONLY.
// Follows CUDA/SVGF/SVGF.h:86-609 (is_tap_consistent, edge_stopping_weights, kernel_svgf_
// reproject / variance / atrous / finalize), CUDA/SVGF/TAA.h:10-172 and the launch order of
// Pathtracer::render (Pathtracer.cpp:798-838).  Surfaces with cudaBoundaryModeClamp become
// clamped reads of pitch x height arrays; buffers the reference leaves uninitialised
// (cuMemAlloc) are taken as zero-initialised.
#include "oracle.h"
#include "oracle_shading.h"

#include <utility>
#include <vector>

namespace {

constexpr float SVGF_EPSILON = 1e-8f;
constexpr int feedback_iteration = 1;

struct Img {
	const oracle_scene & s;
	explicit Img(const oracle_scene & s) : s(s) { }
	int idx(int x, int y) const { return x + y * s.screen_pitch; }
	int clamp_x(int x) const { return x < 0 ? 0 : (x > s.screen_pitch - 1 ? s.screen_pitch - 1 : x); }
	int clamp_y(int y) const { return y < 0 ? 0 : (y > s.screen_height - 1 ? s.screen_height - 1 : y); }
};

inline float4 ld4(const float * p, int i) { return make_float4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]); }
inline void st4(float * p, int i, float4 v) { p[4 * i] = v.x; p[4 * i + 1] = v.y; p[4 * i + 2] = v.z; p[4 * i + 3] = v.w; }

inline float3 oct_decode_normal(float2 f) { // Util.h:250-260
	f = make_float2(f.x * 2.0f - 1.0f, f.y * 2.0f - 1.0f);
	float3 n = make_float3(f.x, f.y, 1.0f - fabsf(f.x) - fabsf(f.y));
	float t = saturate(-n.z);
	n.x += n.x >= 0.0f ? -t : t;
	n.y += n.y >= 0.0f ? -t : t;
	return normalize(n);
}

inline float3 rgb_to_ycocg(float3 c) { // Util.h:70-76
	return make_float3(0.25f * c.x + 0.5f * c.y + 0.25f * c.z, 0.5f * c.x - 0.5f * c.z, -0.25f * c.x + 0.5f * c.y - 0.25f * c.z);
}
inline float3 ycocg_to_rgb(float3 c) { // Util.h:78-84
	return make_float3(saturate(c.x + c.y - c.z), saturate(c.x + c.z), saturate(c.x - c.y - c.z));
}
inline float mitchell_netravali(float x) { // Util.h:262-277
	const float B = 1.0f / 3.0f, C = 1.0f / 3.0f;
	x = fabsf(x);
	float x2 = x * x, x3 = x2 * x;
	if (x < 1.0f) return (1.0f / 6.0f) * ((12.0f - 9.0f * B - 6.0f * C) * x3 + (-18.0f + 12.0f * B + 6.0f * C) * x2 + (6.0f - 2.0f * B));
	if (x < 2.0f) return (1.0f / 6.0f) * ((-B - 6.0f * C) * x3 + (6.0f * B + 30.0f * C) * x2 + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C));
	return 0.0f;
}

inline bool is_tap_consistent(const oracle_scene & s, const oracle_frame & f, int x, int y, float3 normal, float depth) { // SVGF.h:86-103
	if (x < 0 || x >= s.screen_width)  return false;
	if (y < 0 || y >= s.screen_height) return false;
	float4 prev = ld4(f.history_normal_and_depth, x + y * s.screen_pitch);
	float3 prev_normal = oct_decode_normal(make_float2(prev.x, prev.y));
	return dot(normal, prev_normal) > 0.95f && fabsf(depth - prev.z) < 2.0f;
}

inline float2 edge_stopping_weights(const rt_gpu_config & cfg, int delta_x, int delta_y, float2 center_depth_gradient, float center_depth, float depth,
		float3 center_normal, float3 normal, float cl_direct, float cl_indirect, float l_direct, float l_indirect, float denom_direct, float denom_indirect) { // SVGF.h:105-128
	float d = center_depth_gradient.x * float(delta_x) + center_depth_gradient.y * float(delta_y);
	float ln_w_z = fabsf(center_depth - depth) / (cfg.sigma_z * fabsf(d) + SVGF_EPSILON);
	float w_n = powf(fmaxf(0.0f, dot(center_normal, normal)), cfg.sigma_n);
	float w_l_direct   = w_n * expf(-fabsf(cl_direct   - l_direct)   * denom_direct   - ln_w_z);
	float w_l_indirect = w_n * expf(-fabsf(cl_indirect - l_indirect) * denom_indirect - ln_w_z);
	return make_float2(w_l_direct, w_l_indirect);
}

void svgf_reproject(const oracle_scene & s, oracle_frame & f) { // SVGF.h:130-282
	Img im(s);
	const rt_gpu_config & cfg = s.config;
	float * fb_direct = f.framebuffer[RT_AOV_RADIANCE_DIRECT], * fb_indirect = f.framebuffer[RT_AOV_RADIANCE_INDIRECT];
	#pragma omp parallel for schedule(static) // every pixel writes its own outputs only, from buffers no stage writes
	for (int y = 0; y < s.screen_height; y++) for (int x = 0; x < s.screen_width; x++) {
		int pixel_index = im.idx(x, y);
		float4 direct = ld4(fb_direct, pixel_index), indirect = ld4(fb_indirect, pixel_index);

		float4 moment;
		moment.x = luminance(direct.x, direct.y, direct.z);
		moment.y = luminance(indirect.x, indirect.y, indirect.z);
		moment.z = moment.x * moment.x;
		moment.w = moment.y * moment.y;

		float4 normal_and_depth = ld4(f.gbuffer_normal_and_depth, pixel_index);
		float2 screen_position_prev = make_float2(f.gbuffer_screen_position_prev[2 * pixel_index], f.gbuffer_screen_position_prev[2 * pixel_index + 1]);

		float3 normal = oct_decode_normal(make_float2(normal_and_depth.x, normal_and_depth.y));
		float depth = normal_and_depth.z, depth_prev = normal_and_depth.w;
		if (depth == 0.0f) continue; // sky

		float u_prev = 0.5f + 0.5f * screen_position_prev.x;
		float v_prev = 0.5f + 0.5f * screen_position_prev.y;
		float s_prev = u_prev * float(s.screen_width);
		float t_prev = v_prev * float(s.screen_height);
		int x_prev = int(s_prev - 0.5f);
		int y_prev = int(t_prev - 0.5f);

		float fs = s_prev - floorf(s_prev), ft = t_prev - floorf(t_prev);
		float w0 = (1.0f - fs) * (1.0f - ft), w1 = fs * (1.0f - ft), w2 = (1.0f - fs) * ft;
		float w3 = 1.0f - w0 - w1 - w2;
		float weights[4] = { w0, w1, w2, w3 };
		float consistent_weights_sum = 0.0f;

		for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) {
			int tap = i + j * 2;
			if (is_tap_consistent(s, f, x_prev + i, y_prev + j, normal, depth_prev)) consistent_weights_sum += weights[tap];
			else weights[tap] = 0.0f;
		}

		float4 prev_direct = make_float4(0.0f), prev_indirect = make_float4(0.0f), prev_moment = make_float4(0.0f);
		if (consistent_weights_sum > 0.0f) {
			for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) {
				int tap = i + j * 2;
				if (weights[tap] != 0.0f) {
					int tap_index = (x_prev + i) + (y_prev + j) * s.screen_pitch;
					prev_direct   += weights[tap] * ld4(f.history_direct,   tap_index);
					prev_indirect += weights[tap] * ld4(f.history_indirect, tap_index);
					prev_moment   += weights[tap] * ld4(f.history_moment,   tap_index);
				}
			}
		} else {
			for (int j = -1; j <= 1; j++) for (int i = -1; i <= 1; i++) {
				int tap_x = x_prev + i, tap_y = y_prev + j;
				if (is_tap_consistent(s, f, tap_x, tap_y, normal, depth_prev)) {
					int tap_index = tap_x + tap_y * s.screen_pitch;
					prev_direct   += ld4(f.history_direct,   tap_index);
					prev_indirect += ld4(f.history_indirect, tap_index);
					prev_moment   += ld4(f.history_moment,   tap_index);
					consistent_weights_sum += 1.0f;
				}
			}
		}

		if (consistent_weights_sum > 0.0f) {
			prev_direct   = prev_direct   / consistent_weights_sum;
			prev_indirect = prev_indirect / consistent_weights_sum;
			prev_moment   = prev_moment   / consistent_weights_sum;

			int history = ++f.history_length[pixel_index];
			float inv_history = 1.0f / float(history);
			float alpha_colour = fmaxf(cfg.alpha_colour, inv_history);
			float alpha_moment = fmaxf(cfg.alpha_moment, inv_history);

			direct   = lerp_ref(prev_direct,   direct,   alpha_colour);
			indirect = lerp_ref(prev_indirect, indirect, alpha_colour);
			moment   = lerp_ref(prev_moment,   moment,   alpha_moment);

			if (history >= 4 || !cfg.enable_spatial_variance) {
				direct.w   = fmaxf(0.0f, moment.z - moment.x * moment.x);
				indirect.w = fmaxf(0.0f, moment.w - moment.y * moment.y);
			}
		} else {
			f.history_length[pixel_index] = 0;
			direct.w = 1.0f;
			indirect.w = 1.0f;
		}
		st4(fb_direct, pixel_index, direct);
		st4(fb_indirect, pixel_index, indirect);
		st4(f.frame_buffer_moment, pixel_index, moment);
	}
}

void svgf_variance(const oracle_scene & s, oracle_frame & f, const float * d_in, const float * i_in, float * d_out, float * i_out) { // SVGF.h:284-410
	Img im(s);
	const rt_gpu_config & cfg = s.config;
	#pragma omp parallel for schedule(static) // every pixel writes its own outputs only, from buffers no stage writes
	for (int y = 0; y < s.screen_height; y++) for (int x = 0; x < s.screen_pitch; x++) { // note: pitch, not width
		int pixel_index = im.idx(x, y);
		int history = f.history_length[pixel_index];
		if (history >= 4) { st4(d_out, pixel_index, ld4(d_in, pixel_index)); st4(i_out, pixel_index, ld4(i_in, pixel_index)); continue; }

		float luminance_denom = 1.0f / cfg.sigma_l;
		float4 cd = ld4(d_in, pixel_index), ci = ld4(i_in, pixel_index);
		float cl_d = luminance(cd.x, cd.y, cd.z), cl_i = luminance(ci.x, ci.y, ci.z);

		float4 cnd = ld4(f.gbuffer_normal_and_depth, pixel_index);
		float3 center_normal = oct_decode_normal(make_float2(cnd.x, cnd.y));
		float center_depth = cnd.z;
		float2 grad = make_float2(
			ld4(f.gbuffer_normal_and_depth, im.idx(im.clamp_x(x + 1), y)).z - center_depth,
			ld4(f.gbuffer_normal_and_depth, im.idx(x, im.clamp_y(y + 1))).z - center_depth);

		if (center_depth == 0.0f) { st4(d_out, pixel_index, cd); st4(i_out, pixel_index, ci); continue; }

		float sw_d = 1.0f, sw_i = 1.0f;
		float4 sc_d = cd, sc_i = ci;
		float4 sum_moment = make_float4(0.0f);
		const int radius = 3;
		for (int j = -radius; j <= radius; j++) {
			int tap_y = y + j;
			if (tap_y < 0 || tap_y >= s.screen_height) continue;
			for (int i = -radius; i <= radius; i++) {
				int tap_x = x + i;
				if (tap_x < 0 || tap_x >= s.screen_width) continue;
				if (i == 0 && j == 0) continue;
				int tap_index = tap_x + tap_y * s.screen_pitch;
				float4 td = ld4(d_in, tap_index), ti = ld4(i_in, tap_index), moment = ld4(f.frame_buffer_moment, tap_index);
				float l_d = luminance(td.x, td.y, td.z), l_i = luminance(ti.x, ti.y, ti.z);
				float4 nd = ld4(f.gbuffer_normal_and_depth, tap_index);
				float3 normal = oct_decode_normal(make_float2(nd.x, nd.y));
				float2 w = edge_stopping_weights(cfg, i, j, grad, center_depth, nd.z, center_normal, normal, cl_d, cl_i, l_d, l_i, luminance_denom, luminance_denom);
				sw_d += w.x; sw_i += w.y;
				sc_d += w.x * td;
				sc_i += w.y * ti;
				sum_moment += moment * make_float4(w.x, w.y, w.x, w.y);
			}
		}
		sw_d = fmaxf(sw_d, 1e-6f); sw_i = fmaxf(sw_i, 1e-6f);
		sc_d = sc_d / sw_d; sc_i = sc_i / sw_i;
		sum_moment = make_float4(sum_moment.x / sw_d, sum_moment.y / sw_i, sum_moment.z / sw_d, sum_moment.w / sw_i);
		sc_d.w = fmaxf(0.0f, sum_moment.z - sum_moment.x * sum_moment.x);
		sc_i.w = fmaxf(0.0f, sum_moment.w - sum_moment.y * sum_moment.y);
		st4(d_out, pixel_index, sc_d);
		st4(i_out, pixel_index, sc_i);
	}
}

void svgf_atrous(const oracle_scene & s, oracle_frame & f, const float * d_in, const float * i_in, float * d_out, float * i_out, int step_size) { // SVGF.h:416-554
	Img im(s);
	const rt_gpu_config & cfg = s.config;
	#pragma omp parallel for schedule(static) // every pixel writes its own outputs only, from buffers no stage writes
	for (int y = 0; y < s.screen_height; y++) for (int x = 0; x < s.screen_width; x++) {
		int pixel_index = im.idx(x, y);

		float vb_d = 0.0f, vb_i = 0.0f;
		for (int j = -1; j <= 1; j++) {
			int tap_y = y + j < 0 ? 0 : (y + j > s.screen_height - 1 ? s.screen_height - 1 : y + j);
			for (int i = -1; i <= 1; i++) {
				int tap_x = x + i < 0 ? 0 : (x + i > s.screen_width - 1 ? s.screen_width - 1 : x + i);
				float kernel_weight = scalbnf(0.25f, -(abs(i) + abs(j)));
				vb_d += d_in[4 * (tap_x + tap_y * s.screen_pitch) + 3] * kernel_weight;
				vb_i += i_in[4 * (tap_x + tap_y * s.screen_pitch) + 3] * kernel_weight;
			}
		}
		float denom_d = 1.0f / sqrtf(cfg.sigma_l * cfg.sigma_l * fmaxf(0.0f, vb_d) + SVGF_EPSILON); // rsqrtf
		float denom_i = 1.0f / sqrtf(cfg.sigma_l * cfg.sigma_l * fmaxf(0.0f, vb_i) + SVGF_EPSILON);

		float4 cd = ld4(d_in, pixel_index), ci = ld4(i_in, pixel_index);
		float cl_d = luminance(cd.x, cd.y, cd.z), cl_i = luminance(ci.x, ci.y, ci.z);

		float4 cnd = ld4(f.gbuffer_normal_and_depth, pixel_index);
		float3 center_normal = oct_decode_normal(make_float2(cnd.x, cnd.y));
		float center_depth = cnd.z;
		if (center_depth == 0.0f) continue; // sky: out buffers are NOT written

		float2 grad = make_float2(
			ld4(f.gbuffer_normal_and_depth, im.idx(im.clamp_x(x + 1), y)).z - center_depth,
			ld4(f.gbuffer_normal_and_depth, im.idx(x, im.clamp_y(y + 1))).z - center_depth);

		float sw_d = 1.0f, sw_i = 1.0f;
		float4 sc_d = cd, sc_i = ci;
		for (int j = -1; j <= 1; j++) {
			int tap_y = y + j * step_size;
			if (tap_y < 0 || tap_y >= s.screen_height) continue;
			for (int i = -1; i <= 1; i++) {
				int tap_x = x + i * step_size;
				if (tap_x < 0 || tap_x >= s.screen_width) continue;
				if (i == 0 && j == 0) continue;
				int tap_index = tap_x + tap_y * s.screen_pitch;
				float4 td = ld4(d_in, tap_index), ti = ld4(i_in, tap_index);
				float l_d = luminance(td.x, td.y, td.z), l_i = luminance(ti.x, ti.y, ti.z);
				float4 nd = ld4(f.gbuffer_normal_and_depth, tap_index);
				float3 normal = oct_decode_normal(make_float2(nd.x, nd.y));
				float2 w = edge_stopping_weights(cfg, i * step_size, j * step_size, grad, center_depth, nd.z, center_normal, normal, cl_d, cl_i, l_d, l_i, denom_d, denom_i);
				sw_d += w.x; sw_i += w.y;
				sc_d += make_float4(w.x, w.x, w.x, w.x * w.x) * td;
				sc_i += make_float4(w.y, w.y, w.y, w.y * w.y) * ti;
			}
		}
		float inv_d = 1.0f / sw_d, inv_i = 1.0f / sw_i;
		sc_d *= inv_d; sc_i *= inv_i;
		sc_d.w *= inv_d; sc_i.w *= inv_i;
		st4(d_out, pixel_index, sc_d);
		st4(i_out, pixel_index, sc_i);
		if (step_size == (1 << feedback_iteration)) { st4(f.history_direct, pixel_index, sc_d); st4(f.history_indirect, pixel_index, sc_i); }
	}
}

void svgf_finalize(const oracle_scene & s, oracle_frame & f, const float * colour_direct, const float * colour_indirect) { // SVGF.h:559-609
	const rt_gpu_config & cfg = s.config;
	#pragma omp parallel for schedule(static) // every pixel writes its own outputs only, from buffers no stage writes
	for (int y = 0; y < s.screen_height; y++) for (int x = 0; x < s.screen_width; x++) {
		int pixel_index = x + y * s.screen_pitch;
		float4 direct = ld4(colour_direct, pixel_index), indirect = ld4(colour_indirect, pixel_index);
		float4 colour = (direct + indirect) * ld4(f.framebuffer[RT_AOV_ALBEDO], pixel_index);
		st4(f.final_image, pixel_index, colour);

		if (cfg.enable_taa) {
			colour = colour / (1.0f + luminance(colour.x, colour.y, colour.z)); // "pseudo" Reinhard
			colour.x = safe_sqrt(colour.x); colour.y = safe_sqrt(colour.y); colour.z = safe_sqrt(colour.z);
			st4(f.taa_frame_curr, pixel_index, colour);
		}
		float4 moment = ld4(f.frame_buffer_moment, pixel_index);
		float4 normal_and_depth = ld4(f.gbuffer_normal_and_depth, pixel_index);
		if (cfg.num_atrous_iterations <= feedback_iteration) { st4(f.history_direct, pixel_index, direct); st4(f.history_indirect, pixel_index, indirect); }
		st4(f.history_moment, pixel_index, moment);
		st4(f.history_normal_and_depth, pixel_index, normal_and_depth);

		st4(f.gbuffer_normal_and_depth, pixel_index, make_float4(0.0f));
		f.gbuffer_mesh_id_and_triangle_id[2 * pixel_index] = 0; f.gbuffer_mesh_id_and_triangle_id[2 * pixel_index + 1] = 0;
		if (!cfg.enable_taa) { f.gbuffer_screen_position_prev[2 * pixel_index] = 0.0f; f.gbuffer_screen_position_prev[2 * pixel_index + 1] = 0.0f; }
	}
}

inline float3 clamp3(float3 v, float3 lo, float3 hi) { return make_float3(clampf(v.x, lo.x, hi.x), clampf(v.y, lo.y, hi.y), clampf(v.z, lo.z, hi.z)); }

void taa(const oracle_scene & s, oracle_frame & f, int sample_index) { // TAA.h:10-151
	std::vector<float> out(size_t(s.screen_pitch) * s.screen_height * 4);
	memcpy(out.data(), f.final_image, out.size() * sizeof(float));
	#pragma omp parallel for schedule(static) // every pixel writes its own outputs only, from buffers no stage writes
	for (int y = 0; y < s.screen_height; y++) for (int x = 0; x < s.screen_width; x++) {
		int pixel_index = x + y * s.screen_pitch;
		float4 colour = ld4(f.taa_frame_curr, pixel_index);
		if (sample_index == 0) { st4(out.data(), pixel_index, colour); continue; }

		float2 sp = make_float2(f.gbuffer_screen_position_prev[2 * pixel_index], f.gbuffer_screen_position_prev[2 * pixel_index + 1]);
		float s_prev = (0.5f + 0.5f * sp.x) * float(s.screen_width);
		float t_prev = (0.5f + 0.5f * sp.y) * float(s.screen_height);
		int x_prev = int(s_prev + 0.5f), y_prev = int(t_prev + 0.5f);

		float sum_weight = 0.0f;
		float4 sum = make_float4(0.0f);
		for (int j = y_prev - 2; j < y_prev + 2; j++) {
			if (j < 0 || j >= s.screen_height) continue;
			for (int i = x_prev - 2; i < x_prev + 2; i++) {
				if (i < 0 || i >= s.screen_width) continue;
				float weight = mitchell_netravali(float(i) + 0.5f - s_prev) * mitchell_netravali(float(j) + 0.5f - t_prev);
				sum_weight += weight;
				sum += weight * ld4(f.taa_frame_prev, i + j * s.screen_pitch);
			}
		}
		if (sum_weight > 0.0f) {
			float3 colour_curr = rgb_to_ycocg(make_float3(colour));
			float3 colour_prev = rgb_to_ycocg(make_float3(sum / sum_weight));
			float3 avg = colour_curr, var = colour_curr * colour_curr;
			auto tap = [&](int offset) { float3 c = rgb_to_ycocg(make_float3(ld4(f.taa_frame_curr, pixel_index + offset))); avg += c; var += c * c; };
			int p = s.screen_pitch;
			if (x >= 1) {
				if (y >= 1) tap(-p - 1);
				tap(-1);
				if (y < s.screen_height - 1) tap(p - 1);
			}
			if (y >= 1) tap(-p);
			if (y < s.screen_height - 1) tap(p);
			if (x < s.screen_width - 1) {
				if (y >= 1) tap(1 - p);
				tap(1);
				if (y < s.screen_height - 1) tap(1 + p);
			}
			avg *= 1.0f / 9.0f; var *= 1.0f / 9.0f;
			float3 sigma2 = var - avg * avg;
			float3 sigma = make_float3(safe_sqrt(sigma2.x), safe_sqrt(sigma2.y), safe_sqrt(sigma2.z));
			colour_prev = clamp3(colour_prev, avg - 1.25f * sigma, avg + 1.25f * sigma);
			float3 integrated = ycocg_to_rgb(lerp_ref(colour_prev, colour_curr, 0.1f));
			colour.x = integrated.x; colour.y = integrated.y; colour.z = integrated.z;
		}
		st4(out.data(), pixel_index, colour);
	}
	memcpy(f.final_image, out.data(), out.size() * sizeof(float));
}

void taa_finalize(const oracle_scene & s, oracle_frame & f) { // TAA.h:153-172
	#pragma omp parallel for schedule(static) // every pixel writes its own outputs only, from buffers no stage writes
	for (int y = 0; y < s.screen_height; y++) for (int x = 0; x < s.screen_width; x++) {
		int pixel_index = x + y * s.screen_pitch;
		float4 colour = ld4(f.final_image, pixel_index);
		st4(f.taa_frame_prev, pixel_index, colour);
		colour = colour * colour;
		colour = colour / (1.0f - luminance(colour.x, colour.y, colour.z));
		st4(f.final_image, pixel_index, colour);
		f.gbuffer_screen_position_prev[2 * pixel_index] = 0.0f; f.gbuffer_screen_position_prev[2 * pixel_index + 1] = 0.0f;
	}
}

} // namespace

// Pathtracer.cpp:798-838
void oracle_svgf_taa(const oracle_scene & s, oracle_frame & f, int sample_index) {
	svgf_reproject(s, f);

	float * direct_in    = f.framebuffer[RT_AOV_RADIANCE_DIRECT];
	float * indirect_in  = f.framebuffer[RT_AOV_RADIANCE_INDIRECT];
	float * direct_out   = f.accumulator[RT_AOV_RADIANCE_DIRECT];
	float * indirect_out = f.accumulator[RT_AOV_RADIANCE_INDIRECT];

	if (s.config.enable_spatial_variance) {
		svgf_variance(s, f, direct_in, indirect_in, direct_out, indirect_out);
		std::swap(direct_in, direct_out);
		std::swap(indirect_in, indirect_out);
	}
	for (int i = 0; i < s.config.num_atrous_iterations; i++) {
		svgf_atrous(s, f, direct_in, indirect_in, direct_out, indirect_out, 1 << i);
		std::swap(direct_in, direct_out);
		std::swap(indirect_in, indirect_out);
	}
	svgf_finalize(s, f, direct_in, indirect_in);

	if (s.config.enable_taa) {
		taa(s, f, sample_index);
		taa_finalize(s, f);
	}
}
