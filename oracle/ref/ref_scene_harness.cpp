// oracle/ref/ref_scene_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The reference's whole scene-loading side, compiled verbatim from /root/reference/Src (nothing is copied):
// Scene, AssetManager, MitsubaLoader + XMLParser, OBJ / PLY / serialized / hair loaders, TextureLoader (stb_image,
// stb_dxt, mip maps), BVHLoader (.bvh caches, miniz), Sky, the exporters and the argument parser. This file only
// drives them and flattens what they produce, so that tests/test_loaders.py can compare the product's host side
// (gpu-raytracer_amd/host) with it field by field.
//
// Built with clang's MSVC preprocessor rules (-fms-compatibility): the reference's ERROR / WARNING macros rely on
// MSVC dropping the comma in front of an empty __VA_ARGS__.
#include "Core/Format.h"
#include "Core/IO.h"
#include "Config.h"
#include "Args.h"
#include "Renderer/Scene.h"
#include "Assets/BVHLoader.h"
#include "Assets/OBJLoader.h"
#include "Assets/PLYLoader.h"
#include "Assets/TextureLoader.h"
#include "Assets/Mitsuba/SerializedLoader.h"
#include "Assets/Mitsuba/MitshairLoader.h"
#include "Exporters/EXRExporter.h"
#include "Exporters/PPMExporter.h"
#include "Util/ThreadPool.h"
#include "Input.h"

#include <string>
#include <unistd.h>
#include <fcntl.h>

#define EXPORT extern "C" __attribute__((visibility("default")))

bool Input::is_key_down   (SDL_Scancode) { return false; }
bool Input::is_key_pressed(SDL_Scancode) { return false; }

// Util/ThreadPool.cpp is the one file replaced: its sync() waits on a condition variable that the workers signal
// without holding the mutex, and with loads as short as these the wake-up is lost and the process hangs. Work runs
// on the submitting thread instead -- which also makes the order of loads deterministic.
void ThreadPool::init()          { }
void ThreadPool::init(int)       { }
void ThreadPool::free()          { }
void ThreadPool::submit(Work && work) { work(); }
void ThreadPool::sync()          { }

namespace {

// The reference reports progress -- and fatal errors, before it exits -- on stdout; REF_VERBOSE=1 lets it through
struct MuteStdout {
	int saved = -1;
	MuteStdout() { if (getenv("REF_VERBOSE")) return; fflush(stdout); saved = dup(1); int n = open("/dev/null", O_WRONLY); dup2(n, 1); close(n); }
	~MuteStdout() { if (saved < 0) return; fflush(stdout); dup2(saved, 1); close(saved); }
};

struct Loaded {
	Scene * scene;
	std::string description;
};


void put(std::string & s, const char * key, const float * v, int n) {
	char buf[16];
	s += ' '; s += key; s += '=';
	for (int i = 0; i < n; i++) { unsigned bits; memcpy(&bits, v + i, 4); snprintf(buf, sizeof buf, i ? ",%08x" : "%08x", bits); s += buf; }
}
void put(std::string & s, const char * key, int v) { s += ' '; s += key; s += '='; s += std::to_string(v); }
void put(std::string & s, const char * key, const String & v) { s += ' '; s += key; s += "=\""; s.append(v.data(), v.size()); s += '"'; }

// One line per object, floats as bit patterns; gpu-raytracer_amd/host/capi.cpp (grt_scene_describe) writes the same
void describe(Loaded & l) {
	Scene & scene = *l.scene;
	std::string & s = l.description;
	s += "config"; put(s, "width", cpu_config.initial_width); put(s, "height", cpu_config.initial_height); put(s, "num_bounces", gpu_config.num_bounces); s += '\n';
	s += "camera"; put(s, "position", &scene.camera.position.x, 3); put(s, "rotation", &scene.camera.rotation.x, 4); put(s, "fov", &scene.camera.fov, 1);
	put(s, "aperture_radius", &scene.camera.aperture_radius, 1); put(s, "focal_distance", &scene.camera.focal_distance, 1); s += '\n';
	for (size_t i = 0; i < scene.meshes.size(); i++) {
		const Mesh & m = scene.meshes[i];
		s += "mesh " + std::to_string(i); put(s, "name", m.name); put(s, "mesh_data", m.mesh_data_handle.handle); put(s, "material", m.material_handle.handle);
		put(s, "position", &m.position.x, 3); put(s, "rotation", &m.rotation.x, 4); put(s, "scale", &m.scale, 1); s += '\n';
	}
	for (size_t i = 0; i < scene.asset_manager.mesh_datas.size(); i++) {
		s += "mesh_data " + std::to_string(i); put(s, "triangles", int(scene.asset_manager.mesh_datas[i].triangles.size())); s += '\n';
	}
	for (size_t i = 0; i < scene.asset_manager.materials.size(); i++) {
		const Material & m = scene.asset_manager.materials[i];
		s += "material " + std::to_string(i); put(s, "name", m.name); put(s, "type", int(m.type)); put(s, "emission", &m.emission.x, 3); put(s, "diffuse", &m.diffuse.x, 3);
		put(s, "texture", m.texture_handle.handle); put(s, "medium", m.medium_handle.handle); put(s, "ior", &m.index_of_refraction, 1);
		put(s, "eta", &m.eta.x, 3); put(s, "k", &m.k.x, 3); put(s, "linear_roughness", &m.linear_roughness, 1); s += '\n';
	}
	for (size_t i = 0; i < scene.asset_manager.media.size(); i++) {
		const Medium & m = scene.asset_manager.media[i];
		s += "medium " + std::to_string(i); put(s, "name", m.name); put(s, "C", &m.C.x, 3); put(s, "mfp", &m.mfp.x, 3); put(s, "g", &m.g, 1); s += '\n';
	}
	for (size_t i = 0; i < scene.asset_manager.textures.size(); i++) {
		s += "texture " + std::to_string(i); put(s, "name", scene.asset_manager.textures[i].name); s += '\n';
	}
	s += "sky"; put(s, "width", scene.sky.width); put(s, "height", scene.sky.height); s += '\n';
}

} // namespace

// Scene::Scene (Renderer/Scene.cpp:17-46) for one scene file + AssetManager::wait_until_loaded. The .bvh caches the
// asset manager writes land next to the meshes, so callers work on a scratch copy of the scene directory.
EXPORT void * ref_scene_load(const char * filename, const char * sky_filename, int bvh_type, int enable_block_compression, int mipmap_filter, int enable_mipmapping) {
	MuteStdout mute;
	cpu_config = CPUConfig { };
	gpu_config = GPUConfig { };
	cpu_config.scene_filenames.clear();
	cpu_config.scene_filenames.push_back(String(filename));
	cpu_config.sky_filename = String(sky_filename);
	cpu_config.bvh_type = BVHType(bvh_type);
	cpu_config.enable_block_compression = enable_block_compression != 0;
	cpu_config.mipmap_filter = MipmapFilterType(mipmap_filter);
	gpu_config.enable_mipmapping = enable_mipmapping != 0;

	Loaded * l = new Loaded();
	l->scene = new Scene(nullptr);
	l->scene->asset_manager.wait_until_loaded();
	l->scene->update(0.0f);
	describe(*l);
	return l;
}
EXPORT void ref_scene_free(void * h) { Loaded * l = (Loaded *)h; delete l->scene; delete l; }
EXPORT const char * ref_scene_description(void * h) { return ((Loaded *)h)->description.c_str(); }

EXPORT int ref_scene_triangles(void * h, int mesh_data, float * out24) {
	const Array<Triangle> & t = ((Loaded *)h)->scene->asset_manager.mesh_datas[mesh_data].triangles;
	static_assert(sizeof(Triangle) == 96, "host Triangle is 24 floats");
	if (out24) memcpy(out24, t.data(), t.size() * sizeof(Triangle));
	return int(t.size());
}
// info: format (0 BC1, 1 BC2, 2 BC3, 3 RGBA), channels, width, height, mip levels, bytes
EXPORT void ref_scene_texture_info(void * h, int texture, int * info6) {
	const Texture & t = ((Loaded *)h)->scene->asset_manager.textures[texture];
	info6[0] = int(t.format); info6[1] = t.channels; info6[2] = t.width; info6[3] = t.height; info6[4] = t.mip_levels(); info6[5] = int(t.data.size());
}
EXPORT void ref_scene_texture_data(void * h, int texture, unsigned char * data, int * mip_offsets) {
	const Texture & t = ((Loaded *)h)->scene->asset_manager.textures[texture];
	memcpy(data, t.data.data(), t.data.size());
	for (int i = 0; i < t.mip_levels(); i++) mip_offsets[i] = t.mip_offsets[i];
}
EXPORT void ref_scene_mesh_transform(void * h, int mesh, float * out48) { // transform, transform_inv, transform_prev after one update
	const Mesh & m = ((Loaded *)h)->scene->meshes[mesh];
	memcpy(out48, m.transform.cells, 64); memcpy(out48 + 16, m.transform_inv.cells, 64); memcpy(out48 + 32, m.transform_prev.cells, 64);
}
EXPORT int ref_scene_sky(void * h, float * out) {
	const Sky & sky = ((Loaded *)h)->scene->sky;
	if (out) memcpy(out, sky.data.data(), sky.data.size() * sizeof(Vector4));
	return int(sky.data.size());
}

// PPMExporter::save (kind 0, values already in display space) / EXRExporter::save (kind 1) of an RGB float image
EXPORT void ref_export_image(int kind, const char * filename, int pitch, int width, int height, const float * rgb) {
	MuteStdout mute;
	Array<Vector3> data(size_t(pitch) * height);
	memcpy((void *)data.data(), rgb, data.size() * sizeof(Vector3));
	if (kind == 0) PPMExporter::save(String(filename), pitch, width, height, data);
	else           EXRExporter::save(String(filename), pitch, width, height, data);
}

// Args::parse (Args.cpp:51-184) on fresh configurations -> the one-line form `pathtracer --print-config` writes
EXPORT void ref_args_parse(int argc, char ** argv, char * out, int capacity) {
	MuteStdout mute;
	cpu_config = CPUConfig { };
	gpu_config = GPUConfig { };
	cpu_config.scene_filenames.clear();
	Args::parse(argc, argv);
	auto bits = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
	std::string scenes;
	for (size_t i = 0; i < cpu_config.scene_filenames.size(); i++) { if (i) scenes += '|'; scenes.append(cpu_config.scene_filenames[i].data(), cpu_config.scene_filenames[i].size()); }
	std::string output(cpu_config.output_filename.data(), cpu_config.output_filename.size()), sky(cpu_config.sky_filename.data(), cpu_config.sky_filename.size());
	snprintf(out, capacity, "integrator=%d width=%d height=%d num_bounces=%d samples=%d output=\"%s\" scenes=\"%s\" sky=\"%s\" bvh_type=%d nee=%d mis=%d force_rebuild=%d "
	         "optimize=%d opt_time=%d opt_batches=%d sah_node=%08x sah_leaf=%08x sbvh_alpha=%08x mipmap=%d mip_filter=%d compress=%d\n",
	         int(cpu_config.integrator), cpu_config.initial_width, cpu_config.initial_height, gpu_config.num_bounces, cpu_config.output_sample_index,
	         output.c_str(), scenes.c_str(), sky.c_str(), int(cpu_config.bvh_type),
	         int(gpu_config.enable_next_event_estimation), int(gpu_config.enable_multiple_importance_sampling), int(cpu_config.bvh_force_rebuild),
	         int(cpu_config.enable_bvh_optimization), cpu_config.bvh_optimizer_max_time, cpu_config.bvh_optimizer_max_num_batches,
	         bits(cpu_config.sah_cost_node), bits(cpu_config.sah_cost_leaf), bits(cpu_config.sbvh_alpha), int(gpu_config.enable_mipmapping),
	         int(cpu_config.mipmap_filter), int(cpu_config.enable_block_compression));
}

// The mesh loaders on their own: kind 0 OBJLoader::load, 1 PLYLoader::load, 2 SerializedLoader::load(shape_index = arg),
// 3 MitshairLoader::load(radius = arg). Returns the triangle count; call with out24 = NULL first.
EXPORT int ref_load_mesh_file(int kind, const char * filename, float arg, float * out24) {
	MuteStdout mute;
	static Array<Triangle> triangles;
	if (!out24) {
		switch (kind) {
			case 0: triangles = OBJLoader::load(String(filename), nullptr); break;
			case 1: triangles = PLYLoader::load(String(filename), nullptr); break;
			case 2: triangles = SerializedLoader::load(String(filename), nullptr, SourceLocation { }, int(arg)); break;
			case 3: triangles = MitshairLoader::load(String(filename), nullptr, SourceLocation { }, arg); break;
			default: return -1;
		}
	} else {
		memcpy(out24, triangles.data(), triangles.size() * sizeof(Triangle));
	}
	return int(triangles.size());
}
