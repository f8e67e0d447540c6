// Headless `pathtracer` executable: the reference's command line (Src/Args.cpp:51-184) and the
// part of its main loop that matters without a window (Src/Main.cpp:75-150): load the scene,
// run update() / render() until sample_index reaches -N, write the frame (-o file.ppm | file.exr)
// and exit. Window, GUI and the interactive camera are out of scope.
//
// Options without a counterpart in the reference: --device <ordinal>, --devices <a,b,...> (one frame split over several
// GPUs: row tiles dealt round-robin, one RCCL all-gather per render(), host/FrameSplit.h), --bvh-cache <bool>
// (the reference always uses its .bvh caches; here they are opt-in), --merge-static <0..3> (the flattened static geometry of
// DESIGN.md 4.6; 0 = the reference's one-tree-per-mesh layout), --batch <n> samples per
// submission (default 4; 1 = one render() per sample exactly like the reference loop).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../AO.h"
#include "../FrameSplit.h"
#include "../Pathtracer.h"

namespace {

struct Option {
	const char * short_name; // may be null
	const char * long_name;
	const char * help;
	int          argument_count;
	std::function<void(const char * value)> apply;
};

struct CommandLine {
	int  device = 0;
	std::vector<int> devices;   // --devices: more than one entry = tile split (FrameSplit)
	int  batch  = 4;
	bool help   = false;
	bool print_config = false;
};

[[noreturn]] void die(const std::string & message) {
	fprintf(stderr, "%s\n", message.c_str());
	exit(1);
}

int parse_int(const char * text, const char * what) {
	char * end = nullptr;
	long value = strtol(text, &end, 10);
	if (end == text || *end != '\0') die(std::string("invalid integer '") + text + "' for " + what);
	return int(value);
}

float parse_float(const char * text, const char * what) {
	char * end = nullptr;
	float value = strtof(text, &end);
	if (end == text || *end != '\0') die(std::string("invalid number '") + text + "' for " + what);
	return value;
}

bool parse_bool(const char * text) { // Args.cpp:19-37: anything unrecognised counts as true, with a message
	for (const char * t : { "true", "True", "TRUE", "1" })   if (strcmp(text, t) == 0) return true;
	for (const char * f : { "false", "False", "FALSE", "0" }) if (strcmp(text, f) == 0) return false;
	printf("Invalid boolean argument '%s'!\n", text);
	return true;
}

std::vector<Option> make_options(CommandLine & cl) {
	std::vector<Option> o;
	o.push_back({ "I", "integrator", "Choose the integrator type. Supported options: pathtracer, ao", 1, [](const char * v) {
		if      (strcmp(v, "pathtracer") == 0) cpu_config.integrator = IntegratorType::PATHTRACER;
		else if (strcmp(v, "ao") == 0)         cpu_config.integrator = IntegratorType::AO;
		else die(std::string("'") + v + "' is not a recognized integrator type! Supported options: pathtracer, ao");
	} });
	o.push_back({ "W", "width",   "Sets the width of the image",  1, [](const char * v) { cpu_config.initial_width  = parse_int(v, "--width"); } });
	o.push_back({ "H", "height",  "Sets the height of the image", 1, [](const char * v) { cpu_config.initial_height = parse_int(v, "--height"); } });
	o.push_back({ "b", "bounce",  "Sets the number of pathtracing bounces", 1, [](const char * v) {
		int b = parse_int(v, "--bounce");
		gpu_config.num_bounces = b < 0 ? 0 : (b > RT_MAX_BOUNCES - 1 ? RT_MAX_BOUNCES - 1 : b);
	} });
	o.push_back({ "N", "samples", "Sets a target number of samples to use", 1, [](const char * v) { cpu_config.output_sample_index = parse_int(v, "--samples"); } });
	o.push_back({ "o", "output",  "Sets path to output file. Supported formats: ppm, exr", 1, [](const char * v) { cpu_config.output_filename = v; } });
	o.push_back({ "s", "scene",   "Sets path to scene file. Supported formats: Mitsuba XML, OBJ, and PLY", 1, [](const char * v) { cpu_config.scene_filenames.push_back(v); } });
	o.push_back({ "S", "sky",     "Sets path to sky file. Supported formats: HDR", 1, [](const char * v) { cpu_config.sky_filename = v; } });
	// the reference gives --bvh the short name -b as well; --bounce is listed first and wins it
	o.push_back({ nullptr, "bvh", "Sets type of BLAS BVH used. Supported options: sah, sbvh, bvh4, bvh8", 1, [](const char * v) {
		if      (strcmp(v, "sah")  == 0) cpu_config.bvh_type = BVHType::BVH;
		else if (strcmp(v, "sbvh") == 0) cpu_config.bvh_type = BVHType::SBVH;
		else if (strcmp(v, "bvh4") == 0) cpu_config.bvh_type = BVHType::BVH4;
		else if (strcmp(v, "bvh8") == 0) cpu_config.bvh_type = BVHType::BVH8;
		else die(std::string("'") + v + "' is not a recognized BVH type! Supported options: sah, sbvh, bvh4, bvh8");
	} });
	o.push_back({ nullptr, "nee", "Enables or disables Next Event Estimation",        1, [](const char * v) { gpu_config.enable_next_event_estimation        = parse_bool(v); } });
	o.push_back({ nullptr, "mis", "Enables or disables Multiple Importance Sampling", 1, [](const char * v) { gpu_config.enable_multiple_importance_sampling = parse_bool(v); } });
	o.push_back({ nullptr, "force-rebuild", "BVH will not be loaded from disk but rebuilt from scratch", 0, [](const char *) { cpu_config.bvh_force_rebuild = true; } });
	o.push_back({ "O",  "optimize",    "Enables or disables BVH optimization post-processing step", 1, [](const char * v) { cpu_config.enable_bvh_optimization = parse_bool(v); } });
	o.push_back({ "Ot", "opt-time",    "Sets time limit for BVH optimization (milliseconds, as the reference stores it)", 1, [](const char * v) { cpu_config.bvh_optimizer_max_time = parse_int(v, "--opt-time"); } });
	o.push_back({ "Ob", "opt-batches", "Sets a limit on the maximum number of batches used in BVH optimization", 1, [](const char * v) { cpu_config.bvh_optimizer_max_num_batches = parse_int(v, "--opt-batches"); } });
	o.push_back({ nullptr, "sah-node",   "Sets the SAH cost of an internal BVH node", 1, [](const char * v) { cpu_config.sah_cost_node = parse_float(v, "--sah-node"); } });
	o.push_back({ nullptr, "sah-leaf",   "Sets the SAH cost of a leaf BVH node",      1, [](const char * v) { cpu_config.sah_cost_leaf = parse_float(v, "--sah-leaf"); } });
	o.push_back({ nullptr, "sbvh-alpha", "Sets the SBVH alpha constant. An alpha of 1 results in a regular BVH, alpha of 0 results in full SBVH", 1, [](const char * v) { cpu_config.sbvh_alpha = parse_float(v, "--sbvh-alpha"); } });
	o.push_back({ nullptr, "mipmap",     "Enables or disables texture mipmapping",    1, [](const char * v) { gpu_config.enable_mipmapping = parse_bool(v); } });
	o.push_back({ nullptr, "mip-filter", "Sets the downsampling filter for creating mipmaps. Supported options: box, lanczos, kaiser", 1, [](const char * v) {
		if      (strcmp(v, "box")     == 0) cpu_config.mipmap_filter = MipmapFilterType::BOX;
		else if (strcmp(v, "lanczos") == 0) cpu_config.mipmap_filter = MipmapFilterType::LANCZOS;
		else if (strcmp(v, "kaiser")  == 0) cpu_config.mipmap_filter = MipmapFilterType::KAISER;
		else die(std::string("'") + v + "' is not a recognized Mipmap Filter!");
	} });
	o.push_back({ "c", "compress",  "Enables or disables texture block compression (BC1, decoded in the shade kernels; default true, as in the reference)", 1, [](const char * v) { cpu_config.enable_block_compression = parse_bool(v); } });
	o.push_back({ nullptr, "device",    "HIP device ordinal to render on", 1, [&cl](const char * v) { cl.device = parse_int(v, "--device"); } });
	o.push_back({ nullptr, "devices",   "Comma-separated HIP device ordinals: the frame is split into row tiles over them (an ordinal may repeat: contexts sharing a GPU)", 1, [&cl](const char * v) {
		cl.devices.clear();
		std::string list(v);
		for (size_t at = 0; at <= list.size(); ) {
			size_t comma = list.find(',', at); if (comma == std::string::npos) comma = list.size();
			cl.devices.push_back(parse_int(list.substr(at, comma - at).c_str(), "--devices"));
			at = comma + 1;
		}
	} });
	o.push_back({ nullptr, "bvh-cache", "Enables or disables reading and writing <mesh>.bvh cache files", 1, [](const char * v) { cpu_config.enable_bvh_cache = parse_bool(v); } });
	o.push_back({ nullptr, "merge-static", "Static instances flattened into one tree: 1 (default), 2 (no spatial splits), 3 (identity instances only), 0 (one BLAS per mesh under the TLAS, as the reference)", 1, [](const char * v) {
		cpu_config.merge_static = parse_int(v, "--merge-static");
		if (cpu_config.merge_static < 0 || cpu_config.merge_static > 3) die("--merge-static must be 0, 1, 2 or 3");
	} });
	o.push_back({ nullptr, "batch",     "Samples per submission to the device (1..16)", 1, [&cl](const char * v) {
		cl.batch = parse_int(v, "--batch");
		if (cl.batch < 1 || cl.batch > 16) die("--batch must be between 1 and 16");
	} });
	o.push_back({ nullptr, "print-config", "Prints the configuration the command line amounts to and exits", 0, [&cl](const char *) { cl.print_config = true; } });
	o.push_back({ "h", "help", "Displays this message", 0, [&cl](const char *) { cl.help = true; } });
	return o;
}

void print_help(const std::vector<Option> & options) {
	for (const Option & o : options) {
		if (o.short_name) printf("-%s,\t--%-16s%s\n", o.short_name, o.long_name, o.help);
		else              printf("\t--%-16s%s\n", o.long_name, o.help);
	}
}

void parse_command_line(int argc, char ** argv, CommandLine & cl) {
	std::vector<Option> options = make_options(cl);
	for (int i = 1; i < argc; i++) {
		const char * arg = argv[i];
		if (arg[0] != '-') { // without an explicit option: a scene file
			cpu_config.scene_filenames.push_back(arg);
			continue;
		}
		bool long_form = arg[1] == '-';
		const char * name = arg + (long_form ? 2 : 1);
		const Option * match = nullptr;
		for (const Option & o : options) {
			const char * candidate = long_form ? o.long_name : o.short_name;
			if (candidate && strcmp(candidate, name) == 0) { match = &o; break; }
		}
		if (!match) {
			printf("Unrecognized command line option '%s'\nUse --help for a list of valid options\n", arg);
			continue;
		}
		if (i + match->argument_count >= argc) {
			printf("Not enough arguments provided to option '%s'!\n", match->long_name);
			break; // the reference stops parsing here
		}
		match->apply(match->argument_count ? argv[i + 1] : nullptr);
		i += match->argument_count;
	}
	if (cl.help) {
		print_help(options);
		exit(0);
	}
	if (cl.print_config) { // one line, floats as bit patterns; the form oracle/ref/ref_scene_harness.cpp writes for the reference's Args::parse
		auto bits = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
		std::string scenes;
		for (const std::string & s : cpu_config.scene_filenames) { if (!scenes.empty()) scenes += '|'; scenes += s; }
		printf("integrator=%d width=%d height=%d num_bounces=%d samples=%d output=\"%s\" scenes=\"%s\" sky=\"%s\" bvh_type=%d nee=%d mis=%d force_rebuild=%d "
		       "optimize=%d opt_time=%d opt_batches=%d sah_node=%08x sah_leaf=%08x sbvh_alpha=%08x mipmap=%d mip_filter=%d compress=%d\n",
		       int(cpu_config.integrator), cpu_config.initial_width, cpu_config.initial_height, gpu_config.num_bounces, cpu_config.output_sample_index,
		       cpu_config.output_filename.c_str(), scenes.c_str(), cpu_config.sky_filename.c_str(), int(cpu_config.bvh_type),
		       int(gpu_config.enable_next_event_estimation), int(gpu_config.enable_multiple_importance_sampling), int(cpu_config.bvh_force_rebuild),
		       int(cpu_config.enable_bvh_optimization), cpu_config.bvh_optimizer_max_time, cpu_config.bvh_optimizer_max_num_batches,
		       bits(cpu_config.sah_cost_node), bits(cpu_config.sah_cost_leaf), bits(cpu_config.sbvh_alpha), int(gpu_config.enable_mipmapping),
		       int(cpu_config.mipmap_filter), int(cpu_config.enable_block_compression));
		exit(0);
	}
}

double seconds_since(std::chrono::steady_clock::time_point t0) {
	return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace

int main(int argc, char ** argv) {
	setenv("GPU_MAX_HW_QUEUES", "24", 0);   // before the first HIP call: HIP multiplexes its streams onto 4 hardware queues otherwise (rt_api.hip)
	CommandLine cl;
	parse_command_line(argc, argv, cl);

	if (cpu_config.scene_filenames.empty()) die("no scene file given (use -s <scene.xml|.obj|.ply>); --help lists the options");
	if (cpu_config.output_sample_index == CPUConfig::INVALID_SAMPLE) cpu_config.output_sample_index = 1; // no window to keep open: one sample

	try {
		auto t0 = std::chrono::steady_clock::now();
		Scene scene;
		if (cl.devices.size() > 1) { // one frame over several GPUs
			if (cpu_config.integrator == IntegratorType::AO) die("--devices: the tile split exists for the path tracer");
			FrameSplit split(cpu_config.initial_width, cpu_config.initial_height, scene, cl.devices);
			printf("Initialization: %.0f ms (%d ranks)\n", seconds_since(t0) * 1e3, split.world());
			auto t1 = std::chrono::steady_clock::now();
			int target = cpu_config.output_sample_index;
			while (true) {
				split.update(0.0f);
				int remaining = target - split.sample_index() + 1;
				if (cl.batch > 1 && split.sample_index() > 0 && remaining > 1) split.render_samples(remaining < cl.batch ? remaining : cl.batch);
				else split.render();
				if (split.sample_index() >= target) break;
			}
			split.read_framebuffer(); // waits for the devices
			printf("Rendered sample %d at %dx%d in %.1f ms\n", split.sample_index(), split.front().screen_width, split.front().screen_height, seconds_since(t1) * 1e3);
			split.save_image(cpu_config.output_filename);
			printf("Wrote %s\n", cpu_config.output_filename.c_str());
			return 0;
		}
		if (cl.devices.size() == 1) cl.device = cl.devices[0];
		std::unique_ptr<Integrator> integrator;
		if (cpu_config.integrator == IntegratorType::AO) integrator = std::make_unique<AO>        (cpu_config.initial_width, cpu_config.initial_height, scene, cl.device);
		else                                             integrator = std::make_unique<Pathtracer>(cpu_config.initial_width, cpu_config.initial_height, scene, cl.device);
		printf("Initialization: %.0f ms\n", seconds_since(t0) * 1e3);

		Pathtracer * pathtracer = dynamic_cast<Pathtracer *>(integrator.get());
		auto t1 = std::chrono::steady_clock::now();
		int target = cpu_config.output_sample_index;
		// A headless render of `-N n` samples is a batch job: nothing is read before the last sample. Declared to the device library
		// (rt_set_stream_batch), up to eight of its submissions enter the merged wavefront together and walk their bounces side by side
		// instead of one after the other (the same image). Not for SVGF frames (each inherits its predecessor's g-buffers).
		if (pathtracer && integrator->ctx && cl.batch > 1 && target + 1 > cl.batch && !gpu_config.enable_svgf) {
			const long long submissions = (long long)(target + cl.batch) / cl.batch;
			const long long burst = submissions < 8 ? submissions : 8;
			if (rt_set_frame_pipelining(integrator->ctx, 1) != RT_OK || rt_set_stream_batch(integrator->ctx, burst * cl.batch * (long long)cpu_config.initial_width * cpu_config.initial_height) != RT_OK)
				die(std::string("ERROR: ") + rt_last_error(integrator->ctx));
		}
		while (true) {
			integrator->update(0.0f);
			int remaining = target - integrator->sample_index + 1;
			if (pathtracer && cl.batch > 1 && integrator->sample_index > 0 && remaining > 1) {
				pathtracer->render_samples(remaining < cl.batch ? remaining : cl.batch);
			} else {
				integrator->render();
			}
			if (integrator->sample_index >= target) break;
		}
		integrator->read_framebuffer(); // waits for the device
		double render_s = seconds_since(t1);
		printf("Rendered sample %d at %dx%d in %.1f ms\n", integrator->sample_index, integrator->screen_width, integrator->screen_height, render_s * 1e3);

		integrator->save_image(cpu_config.output_filename);
		printf("Wrote %s\n", cpu_config.output_filename.c_str());
	} catch (const std::exception & e) {
		die(std::string("ERROR: ") + e.what());
	}
	return 0;
}
