// Flat C entry points over the C++ host classes, for ctypes (tests, bench.py) and for
// host applications written in C. Nothing here touches the GPU directly: the device is
// reached only through the Pathtracer, i.e. through include/gpu_raytracer_amd.h.
#include <stdexcept>
#include "FrameSplit.h"
#include "Pathtracer.h"
#include "Exporters.h"
#include "BlockCompression.h"
#include "AO.h"

#include <cstring>
#include <string>

static thread_local std::string g_host_error;

#define GRT_TRY   try {
#define GRT_CATCH(ret) } catch (const std::exception & e) { g_host_error = e.what(); return ret; } catch (...) { g_host_error = "unknown C++ exception"; return ret; }

extern "C" {

const char * grt_last_error() { return g_host_error.c_str(); }

// key/value access to the two global config structs (reference: Src/Config.h)
int grt_config_set(const char * key, double value) {
	std::string k(key);
	if      (k == "num_bounces")                         gpu_config.num_bounces = int(value);
	else if (k == "reconstruction_filter")               gpu_config.reconstruction_filter = ReconstructionFilter(int(value));
	else if (k == "enable_mipmapping")                   gpu_config.enable_mipmapping = value != 0;
	else if (k == "enable_next_event_estimation")        gpu_config.enable_next_event_estimation = value != 0;
	else if (k == "enable_multiple_importance_sampling") gpu_config.enable_multiple_importance_sampling = value != 0;
	else if (k == "enable_russian_roulette")             gpu_config.enable_russian_roulette = value != 0;
	else if (k == "enable_svgf")                         gpu_config.enable_svgf = value != 0;
	else if (k == "enable_spatial_variance")             gpu_config.enable_spatial_variance = value != 0;
	else if (k == "enable_taa")                          gpu_config.enable_taa = value != 0;
	else if (k == "alpha_colour")                        gpu_config.alpha_colour = float(value);
	else if (k == "alpha_moment")                        gpu_config.alpha_moment = float(value);
	else if (k == "num_atrous_iterations")               gpu_config.num_atrous_iterations = int(value);
	else if (k == "sigma_z")                             gpu_config.sigma_z = float(value);
	else if (k == "sigma_n")                             gpu_config.sigma_n = float(value);
	else if (k == "sigma_l")                             gpu_config.sigma_l = float(value);
	else if (k == "aov_mask")                            gpu_config.aov_mask = unsigned(value);
	else if (k == "bvh_type") { // 2: binary SAH, 1: binary with spatial splits (SBVH), 4, 8: the wide collapses
		switch (int(value)) {
			case 1:  cpu_config.bvh_type = BVHType::SBVH; break;
			case 2:  cpu_config.bvh_type = BVHType::BVH;  break;
			case 4:  cpu_config.bvh_type = BVHType::BVH4; break;
			case 8:  cpu_config.bvh_type = BVHType::BVH8; break;
			default: g_host_error = "bvh_type must be 1 (sbvh), 2 (sah), 4 (bvh4) or 8 (bvh8)"; return -1;
		}
	}
	else if (k == "mipmap_filter") { // 0 box, 1 lanczos, 2 kaiser (--mip-filter)
		if (int(value) < 0 || int(value) > 2) { g_host_error = "mipmap_filter must be 0 (box), 1 (lanczos) or 2 (kaiser)"; return -1; }
		cpu_config.mipmap_filter = MipmapFilterType(int(value));
	}
	else if (k == "enable_block_compression")            cpu_config.enable_block_compression = value != 0;
	else if (k == "expand_block_compressed_textures")    cpu_config.expand_block_compressed_textures = value != 0;
	else if (k == "svgf_lds_tiles")                      cpu_config.svgf_lds_tiles = value != 0;
	else if (k == "enable_bvh_optimization")             cpu_config.enable_bvh_optimization = value != 0;
	else if (k == "bvh_optimizer_max_time")              cpu_config.bvh_optimizer_max_time = int(value);
	else if (k == "bvh_optimizer_max_num_batches")       cpu_config.bvh_optimizer_max_num_batches = int(value);
	else if (k == "enable_bvh_cache")                    cpu_config.enable_bvh_cache = value != 0;
	else if (k == "bvh_force_rebuild")                   cpu_config.bvh_force_rebuild = value != 0;
	else if (k == "sah_cost_node")                       cpu_config.sah_cost_node = float(value);
	else if (k == "sah_cost_leaf")                       cpu_config.sah_cost_leaf = float(value);
	else if (k == "sbvh_alpha")                          cpu_config.sbvh_alpha = float(value);
	else if (k == "enable_scene_update")                 cpu_config.enable_scene_update = value != 0;
	else if (k == "device_tlas")                         cpu_config.device_tlas = int(value);
	else if (k == "device_blas")                         cpu_config.device_blas = int(value);
	else if (k == "device_presplit")                     cpu_config.device_presplit = float(value);
	else if (k == "static_presplit")                     cpu_config.static_presplit = float(value);
	else if (k == "merge_static")                        cpu_config.merge_static = int(value);
	else if (k == "static_primitive_cost")               cpu_config.static_primitive_cost = float(value);
	else if (k == "static_slot_assignment")              cpu_config.static_slot_assignment = int(value);
	else if (k == "static_slot_learning_rays")           cpu_config.static_slot_learning_rays = int(value);
	else if (k == "static_slot_learning_viewpoint")      cpu_config.static_slot_learning_viewpoint = int(value);
	else if (k == "skip_behind_hit")                     cpu_config.skip_behind_hit = value != 0;
	else if (k == "static_reseat_distance")              cpu_config.static_reseat_distance = float(value);
	else if (k == "static_mesh_copy_limit_mb")           cpu_config.static_mesh_copy_limit_mb = int(value);
	else if (k == "static_copy_budget_mb")               cpu_config.static_copy_budget_mb = int(value);
	else if (k == "initial_width")                       cpu_config.initial_width = int(value);
	else if (k == "initial_height")                      cpu_config.initial_height = int(value);
	else { g_host_error = "unknown config key '" + k + "'"; return -1; }
	return 0;
}

double grt_config_get(const char * key) {
	std::string k(key);
	if (k == "num_bounces")    return gpu_config.num_bounces;
	if (k == "initial_width")  return cpu_config.initial_width;
	if (k == "initial_height") return cpu_config.initial_height;
	if (k == "aov_mask")       return gpu_config.aov_mask;
	if (k == "enable_svgf")    return gpu_config.enable_svgf;
	if (k == "skip_behind_hit") return cpu_config.skip_behind_hit;
	return -1.0;
}

void grt_config_reset() { gpu_config = GPUConfig(); cpu_config = CPUConfig(); }

// Loads one scene file (.xml / .obj); sky_filename may be NULL/"" for the constant sky.
void * grt_scene_load(const char * filename, const char * sky_filename) {
	GRT_TRY
		cpu_config.scene_filenames = { std::string(filename) };
		cpu_config.sky_filename = sky_filename ? sky_filename : "";
		return new Scene();
	GRT_CATCH(nullptr)
}
void grt_scene_free(void * scene) { delete (Scene *)scene; }

int grt_scene_mesh_count(void * scene)     { return int(((Scene *)scene)->meshes.size()); }
int grt_scene_material_count(void * scene) { return int(((Scene *)scene)->asset_manager.materials.size()); }
int grt_scene_texture_count(void * scene)  { return int(((Scene *)scene)->asset_manager.textures.size()); }
double grt_scene_bvh_build_ms(void * scene) { return ((Scene *)scene)->asset_manager.bvh_build_ms; }

void grt_scene_set_sky_scale(void * scene, float scale) { ((Scene *)scene)->sky.scale = scale; }

// position xyz, rotation quaternion xyzw, fov in radians (<= 0 keeps the current one)
void grt_scene_set_camera(void * scene, const float * position, const float * rotation, float fov) {
	Camera & c = ((Scene *)scene)->camera;
	c.position = Vector3(position[0], position[1], position[2]);
	c.rotation = Quaternion(rotation[0], rotation[1], rotation[2], rotation[3]);
	if (fov > 0.0f) c.set_fov(fov);
	c.moved = true;
}
void grt_scene_get_camera(void * scene, float * position, float * rotation, float * fov) {
	const Camera & c = ((Scene *)scene)->camera;
	position[0] = c.position.x; position[1] = c.position.y; position[2] = c.position.z;
	rotation[0] = c.rotation.x; rotation[1] = c.rotation.y; rotation[2] = c.rotation.z; rotation[3] = c.rotation.w;
	*fov = c.fov;
}

// Overrides one material (used by the "odd materials -> roughplastic" Sponza variant)
// Mesh::position / rotation (quaternion x, y, z, w) / scale, as the reference's UI edits them; takes effect
// at the next Integrator::update() that rebuilds the TLAS (invalidated_scene or enable_scene_update).
int grt_scene_set_mesh_transform(void * scene, int index, const float * position, const float * rotation_xyzw, float scale) {
	Scene * s = (Scene *)scene;
	if (index < 0 || index >= int(s->meshes.size())) { g_host_error = "grt_scene_set_mesh_transform: mesh index out of range"; return -1; }
	Mesh & m = s->meshes[index];
	m.position = Vector3(position[0], position[1], position[2]);
	m.rotation = Quaternion(rotation_xyzw[0], rotation_xyzw[1], rotation_xyzw[2], rotation_xyzw[3]);
	m.scale = scale;
	return 0;
}
int grt_scene_get_mesh_transform(void * scene, int index, float * position, float * rotation_xyzw, float * scale) {
	Scene * s = (Scene *)scene;
	if (index < 0 || index >= int(s->meshes.size())) { g_host_error = "grt_scene_get_mesh_transform: mesh index out of range"; return -1; }
	const Mesh & m = s->meshes[index];
	position[0] = m.position.x; position[1] = m.position.y; position[2] = m.position.z;
	rotation_xyzw[0] = m.rotation.x; rotation_xyzw[1] = m.rotation.y; rotation_xyzw[2] = m.rotation.z; rotation_xyzw[3] = m.rotation.w;
	*scale = m.scale;
	return 0;
}

int grt_scene_set_material(void * scene, int index, int type, const float * diffuse, float linear_roughness) {
	Scene * s = (Scene *)scene;
	if (index < 0 || index >= int(s->asset_manager.materials.size())) { g_host_error = "material index out of range"; return -1; }
	Material & m = s->asset_manager.materials[index];
	m.type = Material::Type(type);
	if (diffuse) m.diffuse = Vector3(diffuse[0], diffuse[1], diffuse[2]);
	m.linear_roughness = linear_roughness;
	return 0;
}
int grt_scene_material_type(void * scene, int index) { return int(((Scene *)scene)->asset_manager.materials[index].type); }

// A line-per-object listing of what the loaders produced (camera, meshes, materials, media, texture names, sky size),
// floats as bit patterns: `pathtracer`-independent way to diff two loads of a scene. Returns the length needed.
static void describe_floats(std::string & s, const char * key, const float * v, int n) {
	char buf[16];
	s += ' '; s += key; s += '=';
	for (int i = 0; i < n; i++) { unsigned bits; memcpy(&bits, v + i, 4); snprintf(buf, sizeof buf, i ? ",%08x" : "%08x", bits); s += buf; }
}
static void describe_int(std::string & s, const char * key, int v) { s += ' '; s += key; s += '='; s += std::to_string(v); }
static void describe_str(std::string & s, const char * key, const std::string & v) { s += ' '; s += key; s += "=\""; s += v; s += '"'; }
size_t grt_scene_describe(void * scene_handle, char * out, size_t capacity) {
	Scene & scene = *(Scene *)scene_handle;
	std::string s;
	s += "config"; describe_int(s, "width", cpu_config.initial_width); describe_int(s, "height", cpu_config.initial_height); describe_int(s, "num_bounces", gpu_config.num_bounces); s += '\n';
	s += "camera"; describe_floats(s, "position", &scene.camera.position.x, 3); describe_floats(s, "rotation", &scene.camera.rotation.x, 4); describe_floats(s, "fov", &scene.camera.fov, 1);
	describe_floats(s, "aperture_radius", &scene.camera.aperture_radius, 1); describe_floats(s, "focal_distance", &scene.camera.focal_distance, 1); s += '\n';
	for (size_t i = 0; i < scene.meshes.size(); i++) {
		const Mesh & m = scene.meshes[i];
		s += "mesh " + std::to_string(i); describe_str(s, "name", m.name); describe_int(s, "mesh_data", m.mesh_data_handle.handle); describe_int(s, "material", m.material_handle.handle);
		describe_floats(s, "position", &m.position.x, 3); describe_floats(s, "rotation", &m.rotation.x, 4); describe_floats(s, "scale", &m.scale, 1); s += '\n';
	}
	for (size_t i = 0; i < scene.asset_manager.mesh_datas.size(); i++) {
		s += "mesh_data " + std::to_string(i); describe_int(s, "triangles", int(scene.asset_manager.mesh_datas[i].triangles.size())); s += '\n';
	}
	for (size_t i = 0; i < scene.asset_manager.materials.size(); i++) {
		const Material & m = scene.asset_manager.materials[i];
		s += "material " + std::to_string(i); describe_str(s, "name", m.name); describe_int(s, "type", int(m.type)); describe_floats(s, "emission", &m.emission.x, 3); describe_floats(s, "diffuse", &m.diffuse.x, 3);
		describe_int(s, "texture", m.texture_handle.handle); describe_int(s, "medium", m.medium_handle.handle); describe_floats(s, "ior", &m.index_of_refraction, 1);
		describe_floats(s, "eta", &m.eta.x, 3); describe_floats(s, "k", &m.k.x, 3); describe_floats(s, "linear_roughness", &m.linear_roughness, 1); s += '\n';
	}
	for (size_t i = 0; i < scene.asset_manager.media.size(); i++) {
		const Medium & m = scene.asset_manager.media[i];
		s += "medium " + std::to_string(i); describe_str(s, "name", m.name); describe_floats(s, "C", &m.C.x, 3); describe_floats(s, "mfp", &m.mfp.x, 3); describe_floats(s, "g", &m.g, 1); s += '\n';
	}
	for (size_t i = 0; i < scene.asset_manager.textures.size(); i++) {
		s += "texture " + std::to_string(i); describe_str(s, "name", scene.asset_manager.textures[i].name); s += '\n';
	}
	s += "sky"; describe_int(s, "width", scene.sky.width); describe_int(s, "height", scene.sky.height); s += '\n';
	if (out && capacity) { size_t n = std::min(capacity - 1, s.size()); memcpy(out, s.data(), n); out[n] = 0; }
	return s.size() + 1;
}
// One texture of the scene as the device will get it. info: width, height, mip levels, lod_width, lod_height, texel count
int grt_scene_texture_info(void * scene, int texture, int * info6) {
	GRT_TRY
		const Texture & t = ((Scene *)scene)->asset_manager.textures.at(texture);
		info6[0] = t.width; info6[1] = t.height; info6[2] = t.mip_levels(); info6[3] = t.lod_width; info6[4] = t.lod_height; info6[5] = int(t.texels.size() / 4);
		return 0;
	GRT_CATCH(-1)
}
int grt_scene_texture_data(void * scene, int texture, unsigned char * rgba8, int * mip_offsets_in_texels) {
	GRT_TRY
		const Texture & t = ((Scene *)scene)->asset_manager.textures.at(texture);
		memcpy(rgba8, t.texels.data(), t.texels.size());
		for (int i = 0; i < t.mip_levels(); i++) mip_offsets_in_texels[i] = int(t.mip_offsets[i]);
		return 0;
	GRT_CATCH(-1)
}
int grt_scene_sky(void * scene, float * out_rgba) {
	const Sky & sky = ((Scene *)scene)->sky;
	if (out_rgba) memcpy(out_rgba, sky.data.data(), sky.data.size() * sizeof(Vector4));
	return int(sky.data.size());
}

// Triangles / BLAS of one MeshData, for the builder parity tests
int grt_scene_mesh_data_count(void * scene) { return int(((Scene *)scene)->asset_manager.mesh_datas.size()); }
int grt_scene_wait_until_loaded(void * scene) {
	GRT_TRY
		((Scene *)scene)->asset_manager.wait_until_loaded();
		return 0;
	GRT_CATCH(-1)
}
const void * grt_mesh_data_array(void * scene, int mesh_data, const char * name, size_t * bytes) {
	Scene * s = (Scene *)scene;
	const MeshData & md = s->asset_manager.mesh_datas[mesh_data];
	std::string n(name);
#define RET(vec) { *bytes = (vec).size() * sizeof((vec)[0]); return (vec).data(); }
	if (n == "triangles")    RET(md.triangles)
	if (n == "bvh2_nodes")   RET(md.bvh2.nodes)
	if (n == "bvh2_indices") RET(md.bvh2.indices)
	if (n == "bvh4_nodes")   RET(md.bvh4.nodes)
	if (n == "bvh4_indices") RET(md.bvh4.indices)
	if (n == "bvh8_nodes")   RET(md.bvh8.nodes)
	if (n == "bvh8_indices") RET(md.bvh8.indices)
	if (n == "device_bvh2_nodes")   RET(md.device_bvh2.nodes)
	if (n == "device_bvh2_indices") RET(md.device_bvh2.indices)
	if (n == "device_bvh4_nodes")   RET(md.device_bvh4.nodes)
	*bytes = 0;
	return nullptr;
}

// device_ordinal < 0: host-only baking (no GPU needed)
// The handles below are Integrator pointers: the path tracer and the AO integrator share every
// accessor except the few that only the path tracer has (light tables, render_samples).
static Integrator * as_integrator(void * handle) { return (Integrator *)handle; }
static Pathtracer * as_pathtracer(void * handle) {
	Pathtracer * p = dynamic_cast<Pathtracer *>((Integrator *)handle);
	if (!p) throw std::runtime_error("this call needs a Pathtracer handle");
	return p;
}

void * grt_ao_create(void * scene, int width, int height, int device_ordinal) {
	GRT_TRY
		return static_cast<Integrator *>(new AO(width, height, *(Scene *)scene, device_ordinal));
	GRT_CATCH(nullptr)
}
int grt_ao_set_radius(void * ao, float radius) {
	GRT_TRY
		AO * a = dynamic_cast<AO *>(as_integrator(ao));
		if (!a) throw std::runtime_error("grt_ao_set_radius needs an AO handle");
		a->ao_radius = radius;
		return 0;
	GRT_CATCH(-1)
}

void * grt_pathtracer_create(void * scene, int width, int height, int device_ordinal) {
	GRT_TRY
		return static_cast<Integrator *>(new Pathtracer(width, height, *(Scene *)scene, device_ordinal));
	GRT_CATCH(nullptr)
}
void grt_pathtracer_free(void * pt) { delete as_integrator(pt); }

int grt_pathtracer_update(void * pt, float delta) {
	GRT_TRY
		as_integrator(pt)->update(delta);
		return 0;
	GRT_CATCH(-1)
}
int grt_pathtracer_render(void * pt) {
	GRT_TRY
		as_integrator(pt)->render();
		return 0;
	GRT_CATCH(-1)
}
int grt_pathtracer_render_samples(void * pt, int count) {
	GRT_TRY
		as_pathtracer(pt)->render_samples(count);
		return 0;
	GRT_CATCH(-1)
}
int grt_pathtracer_resize(void * pt, int width, int height) {
	GRT_TRY
		Integrator * p = as_integrator(pt);
		p->resize_free();
		p->resize_init(width, height);
		return 0;
	GRT_CATCH(-1)
}
int grt_pathtracer_set_pixel_range(void * pt, int offset, int count) {
	GRT_TRY
		as_integrator(pt)->set_pixel_range(offset, count);
		return 0;
	GRT_CATCH(-1)
}
int  grt_pathtracer_sample_index(void * pt) { return as_integrator(pt)->sample_index; }
int  grt_pathtracer_screen_pitch(void * pt) { return as_integrator(pt)->screen_pitch; }
void grt_pathtracer_invalidate(void * pt, const char * what) {
	Integrator * p = as_integrator(pt);
	std::string w(what);
	if (w == "scene")      p->invalidated_scene = true;
	if (w == "sky")        p->invalidated_sky = true;
	if (w == "materials")  p->invalidated_materials = true;
	if (w == "mediums")    p->invalidated_mediums = true;
	if (w == "camera")     p->invalidated_camera = true;
	if (w == "gpu_config") p->invalidated_gpu_config = true;
	if (w == "aovs")       p->invalidated_aovs = true;
}
void grt_pathtracer_aov_enable(void * pt, int aov, int enable) {
	if (enable) as_integrator(pt)->aov_enable(AOVType(aov)); else as_integrator(pt)->aov_disable(AOVType(aov));
}
void grt_pathtracer_set_pixel_query(void * pt, int x, int y) { as_integrator(pt)->set_pixel_query(x, y); }
void grt_pathtracer_get_pixel_query(void * pt, int * pixel_index, int * mesh_id, int * triangle_id, int * status) {
	Integrator * p = as_integrator(pt);
	*pixel_index = p->pixel_query.pixel_index; *mesh_id = p->pixel_query.mesh_id; *triangle_id = p->pixel_query.triangle_id;
	*status = int(p->pixel_query_status);
}
void * grt_pathtracer_context(void * pt) { return as_integrator(pt)->ctx; }
float  grt_pathtracer_device_blas_build_ms(void * pt) { return as_integrator(pt)->device_blas_build_ms; }
// Flattened static geometry: instances in it (0: none, or dissolved because one of them moved), and what its tree took to build on the host
int    grt_pathtracer_static_geometry_members(void * pt) { Integrator * p = as_integrator(pt); return p->static_geometry.active ? int(p->static_geometry.members.size()) : 0; }
// 1: everything is in the flattened tree, rays start inside it (rt_set_static_geometry); 0: there is a TLAS
// 1: closest-hit rays take the skipping walk (the rule of rt_skip_walk, csrc/rt_types.h, evaluated on the host's own arrays so that a host-only integrator answers too)
int    grt_pathtracer_skip_behind_hit(void * pt) {
	Integrator * p = as_integrator(pt);
	bool whole_scene = p->static_geometry.active && p->static_geometry.movers.empty();
	bool below_4gib = rt_geometry_fits_flat_engine(p->aggregated_bvh_nodes_8.size(), p->aggregated_triangles.size()) != 0;
	return cpu_config.skip_behind_hit && cpu_config.bvh_type == BVHType::BVH8 && whole_scene && below_4gib ? 1 : 0;
}
int    grt_pathtracer_static_geometry_whole_scene(void * pt) { Integrator * p = as_integrator(pt); return p->static_geometry.active && p->static_geometry.movers.empty() ? 1 : 0; }
// the flattened tree's root node and how many nodes from it (breadth-first order) are its top levels
int    grt_pathtracer_static_geometry_root(void * pt) { return as_integrator(pt)->static_geometry.root; }
int    grt_pathtracer_static_geometry_top_nodes(void * pt) { Integrator * p = as_integrator(pt); return p->static_geometry.active ? p->static_geometry.top_nodes : 0; }
// re-flattening when a member starts to move: 1 (default) builds the new tree on a worker thread while the frame loop renders in the
// reference's layout, 0 rebuilds inside update(); how many such background builds have been installed; whether one is in progress
// the seating of the flattened tree beside the frame loop (Integrator.h: PendingReseat)
int    grt_pathtracer_reseats_completed(void * pt) { return as_integrator(pt)->reseats_completed; }
int    grt_pathtracer_reseat_pending(void * pt) { return as_integrator(pt)->pending_reseat ? 1 : 0; }
double grt_pathtracer_last_reseat_seconds(void * pt) { return as_integrator(pt)->last_reseat_seconds; }
void   grt_pathtracer_set_reseat_asynchronously(void * pt, int enable) { as_integrator(pt)->reseat_asynchronously = enable != 0; }
void grt_pathtracer_set_flatten_asynchronously(void * pt, int enable) { as_integrator(pt)->flatten_asynchronously = enable != 0; }
int  grt_pathtracer_reflattens_completed(void * pt) { return as_integrator(pt)->reflattens_completed; }
int  grt_pathtracer_reflatten_in_progress(void * pt) { Integrator * p = as_integrator(pt); return p->pending_flatten ? (p->pending_flatten->ready.load() ? 2 : 1) : 0; }
// bytes the flattened tree adds to the device's geometry: its triangle copies (shading + traversal records) and its nodes
double grt_pathtracer_static_geometry_bytes(void * pt) { Integrator * p = as_integrator(pt); return p->static_geometry.active ? double(p->static_geometry.copy_bytes) : 0.0; }
double grt_pathtracer_static_geometry_build_seconds(void * pt) { return as_integrator(pt)->static_geometry.build_seconds; }
float  grt_pathtracer_lights_total_weight(void * pt) { { Pathtracer * p = dynamic_cast<Pathtracer *>(as_integrator(pt)); return p ? p->lights_total_weight : 0.0f; } }

int grt_pathtracer_read_aov(void * pt, int aov, int accumulated, float * dst) {
	GRT_TRY
		std::vector<float> image = as_integrator(pt)->read_aov(AOVType(aov), accumulated != 0);
		memcpy(dst, image.data(), image.size() * sizeof(float));
		return 0;
	GRT_CATCH(-1)
}
int grt_pathtracer_save_image(void * pt, const char * filename) {
	GRT_TRY
		as_integrator(pt)->save_image(filename);
		return 0;
	GRT_CATCH(-1)
}
// Writes an RGB float image (x + y * pitch, row 0 at the bottom) through the exporters, without an integrator
int grt_export_image(const char * filename, int pitch, int width, int height, const float * rgb) {
	GRT_TRY
		std::vector<Vector3> data(size_t(pitch) * height);
		memcpy((void *)data.data(), rgb, data.size() * sizeof(Vector3));
		std::string error;
		if (!Exporters::save(filename, pitch, width, height, data, &error)) throw std::runtime_error(error);
		return 0;
	GRT_CATCH(-1)
}
// PPMExporter::save on its own: `rgb` is already in display space (no tone mapping)
int grt_export_ppm_display(const char * filename, int pitch, int width, int height, const float * rgb) {
	GRT_TRY
		std::vector<Vector3> data(size_t(pitch) * height);
		memcpy((void *)data.data(), rgb, data.size() * sizeof(Vector3));
		if (!PPMExporter::save(filename, pitch, width, height, data)) throw std::runtime_error(std::string("failed to write '") + filename + "'");
		return 0;
	GRT_CATCH(-1)
}
int grt_pathtracer_read_framebuffer(void * pt, float * dst) {
	GRT_TRY
		std::vector<float> image = as_integrator(pt)->read_framebuffer();
		memcpy(dst, image.data(), image.size() * sizeof(float));
		return 0;
	GRT_CATCH(-1)
}

// ---- FrameSplit: several GPUs (or, for tests, several contexts of one GPU) render one frame ------------------------
void * grt_frame_split_create(void * scene, int width, int height, const int * device_ordinals, int count) {
	GRT_TRY
		return new FrameSplit(width, height, *(Scene *)scene, std::vector<int>(device_ordinals, device_ordinals + count));
	GRT_CATCH(nullptr)
}
void grt_frame_split_free(void * split) { delete (FrameSplit *)split; }
int grt_frame_split_update(void * split, float delta) {
	GRT_TRY
		((FrameSplit *)split)->update(delta);
		return 0;
	GRT_CATCH(-1)
}
int grt_frame_split_render(void * split) {
	GRT_TRY
		((FrameSplit *)split)->render();
		return 0;
	GRT_CATCH(-1)
}
int grt_frame_split_render_samples(void * split, int count) {
	GRT_TRY
		((FrameSplit *)split)->render_samples(count);
		return 0;
	GRT_CATCH(-1)
}
int grt_frame_split_submitting_threads(void * split) { return ((FrameSplit *)split)->submitting_threads(); }
// the integrator of one rank (a grt_pathtracer_* handle owned by the split): its framebuffer is the whole frame after render()
void * grt_frame_split_rank(void * split, int rank) {
	FrameSplit * s = (FrameSplit *)split;
	return rank >= 0 && rank < s->world() ? static_cast<Integrator *>(s->ranks[rank].get()) : nullptr;
}

// Host staging arrays by name (what the device was / would be given), read-only views.
const void * grt_pathtracer_array(void * pt, const char * name, size_t * bytes) {
	Integrator * p = as_integrator(pt);
	std::string n(name);
	if (p->tlas_on_device) { // the views below show what the device built
		p->sync_host_view_of_device_tlas();
		if (n == "light_mesh_transform_indices") if (Pathtracer * pt_only = dynamic_cast<Pathtracer *>(p)) {
			// the device gets scene indices and maps them itself; a reader of the TLAS-ordered tables wants positions
			static thread_local std::vector<int> positions;
			std::vector<int> position_of(p->tlas.indices.size());
			for (size_t i = 0; i < p->tlas.indices.size(); i++) position_of[size_t(p->tlas.indices[i])] = int(i);
			positions.clear();
			for (int scene_index : pt_only->light_mesh_transform_indices) positions.push_back(position_of[size_t(scene_index)]);
			*bytes = positions.size() * sizeof(int);
			return positions.data();
		}
	}
	if (n == "triangles")             RET(p->aggregated_triangles)
	if (n == "bvh8_nodes")            RET(p->aggregated_bvh_nodes_8)
	if (n == "bvh2_nodes")            RET(p->aggregated_bvh_nodes_2)
	if (n == "bvh4_nodes")            RET(p->aggregated_bvh_nodes_4)
	if (n == "reverse_indices")       RET(p->reverse_indices)
	if (n == "alias_mesh_ids")        RET(p->alias_mesh_ids)
	if (n == "alias_triangle_ids")    RET(p->alias_triangle_ids)
	if (n == "mesh_bvh_root_indices") RET(p->mesh_bvh_root_indices)
	if (n == "mesh_material_ids")     RET(p->mesh_material_ids)
	if (n == "mesh_transforms")       RET(p->mesh_transforms)
	if (n == "mesh_transforms_inv")   RET(p->mesh_transforms_inv)
	if (n == "mesh_transforms_prev")  RET(p->mesh_transforms_prev)
	if (n == "material_types")        RET(p->material_types)
	if (n == "materials")             RET(p->materials)
	if (n == "media")                 RET(p->media)
	if (n.rfind("scene_order_", 0) == 0) { // the inputs of rt_build_tlas, filled on demand (the parity tests build from them on the CPU)
		if (!p->tlas_on_device) p->fill_scene_order_tables();
		if (n == "scene_order_roots")           RET(p->scene_order_roots)
		if (n == "scene_order_materials")       RET(p->scene_order_materials)
		if (n == "scene_order_transforms")      RET(p->scene_order_transforms)
		if (n == "scene_order_transforms_inv")  RET(p->scene_order_transforms_inv)
		if (n == "scene_order_transforms_prev") RET(p->scene_order_transforms_prev)
		if (n == "scene_order_boxes")           RET(p->scene_order_boxes)
	}
	if (n == "tlas_indices")          RET(p->tlas.indices)
	if (n == "tlas_nodes")            RET(p->tlas.nodes)
	if (n == "tlas_raw_nodes")        RET(p->tlas_raw.nodes)
	if (n == "pmj_samples")           RET(p->pmj_samples)
	if (n == "blue_noise")            RET(p->blue_noise)
	if (Pathtracer * pt_only = dynamic_cast<Pathtracer *>(p)) {
		if (n == "light_triangle_indices")                RET(pt_only->light_triangle_indices)
		if (n == "light_triangle_cumulative_probability") RET(pt_only->light_triangle_cumulative_probability)
		if (n == "light_mesh_cumulative_probability")     RET(pt_only->light_mesh_cumulative_probability)
		if (n == "light_mesh_triangle_span")              RET(pt_only->light_mesh_triangle_span)
		if (n == "light_mesh_transform_indices")          RET(pt_only->light_mesh_transform_indices)
		if (n == "svgf_matrices")         RET(pt_only->svgf_matrices)
	}
	if (n == "sky")                   RET(p->scene.sky.data)
	if (n == "camera") { *bytes = sizeof(rt_camera); return &p->device_camera; }
	*bytes = 0;
	return nullptr;
#undef RET
}
void grt_pathtracer_sky_size(void * pt, int * w, int * h, float * scale) {
	Integrator * p = as_integrator(pt);
	*w = p->scene.sky.width; *h = p->scene.sky.height; *scale = p->scene.sky.scale;
}
// Texture table views
int grt_pathtracer_texture_lod_size(void * pt, int index, int * lod_width, int * lod_height) {
	const std::vector<Texture> & t = as_integrator(pt)->scene.asset_manager.textures;
	if (index < 0 || index >= int(t.size())) return -1;
	*lod_width = t[index].lod_width; *lod_height = t[index].lod_height;
	return 0;
}
int grt_pathtracer_texture(void * pt, int index, const unsigned char ** texels, int * width, int * height, int * mip_levels) {
	Integrator * p = as_integrator(pt);
	const std::vector<Texture> & t = p->scene.asset_manager.textures;
	if (index < 0 || index >= int(t.size())) return -1;
	*texels = t[index].texels.data(); *width = t[index].width; *height = t[index].height; *mip_levels = t[index].mip_levels();
	return 0;
}
void grt_pathtracer_device_config(void * pt, rt_gpu_config * out) { *out = as_integrator(pt)->make_device_config(); }
int  grt_pathtracer_counters(void * pt, rt_counters * out) {
	GRT_TRY
		*out = as_integrator(pt)->counters();
		return 0;
	GRT_CATCH(-1)
}

// One mip-filter step on float4 texels (filter: 0 box, 1 lanczos, 2 kaiser), for the parity test against oracle/_ref
int grt_mipmap_downsample(int filter, int w_src, int h_src, int w_dst, int h_dst, const float * src, float * dst) {
	GRT_TRY
		std::vector<Vector4> temp;
		TextureLoader::downsample(MipmapFilterType(filter), w_src, h_src, w_dst, h_dst, (const Vector4 *)src, (Vector4 *)dst, temp);
		return 0;
	GRT_CATCH(-1)
}

// Tessellated primitive shapes (shape: 0 rectangle, 1 cube, 2 disk, 3 cylinder, 4 sphere) and the sky loader,
// for the parity tests against the reference's Geometry.cpp / stbi_loadf in oracle/_ref
int grt_geometry_shape(int shape, const float * transform16, const float * p0, const float * p1, float radius, int detail, float * dst, int dst_triangles) {
	GRT_TRY
		Matrix4 transform;
		memcpy(transform.cells, transform16, 16 * sizeof(float));
		std::vector<Triangle> triangles;
		switch (shape) {
			case 0: triangles = Geometry::rectangle(transform); break;
			case 1: triangles = Geometry::cube(transform); break;
			case 2: triangles = detail > 0 ? Geometry::disk(transform, detail) : Geometry::disk(transform); break;
			case 3: triangles = detail > 0 ? Geometry::cylinder(transform, Vector3(p0[0], p0[1], p0[2]), Vector3(p1[0], p1[1], p1[2]), radius, detail)
			                               : Geometry::cylinder(transform, Vector3(p0[0], p0[1], p0[2]), Vector3(p1[0], p1[1], p1[2]), radius); break;
			default: triangles = detail >= 0 ? Geometry::sphere(transform, detail) : Geometry::sphere(transform); break;
		}
		int count = int(triangles.size());
		if (dst && dst_triangles >= count) memcpy((void *)dst, triangles.data(), size_t(count) * sizeof(Triangle));
		return count;
	GRT_CATCH(-1)
}
// Camera state after resize(width, height) and `updates` calls of update(0), laid out as oracle/ref's ref_camera_state
int grt_camera_state(float fov, int width, int height, const float * position3, const float * rotation4, int updates, float * out58) {
	GRT_TRY
		Camera camera(fov);
		camera.resize(width, height);
		camera.position = Vector3(position3[0], position3[1], position3[2]);
		camera.rotation = Quaternion(rotation4[0], rotation4[1], rotation4[2], rotation4[3]);
		memset(camera.view_projection.cells, 0, sizeof(camera.view_projection.cells));
		for (int i = 0; i < updates; i++) camera.update(0.0f);
		float * o = out58;
		for (const Vector3 & v : { camera.bottom_left_corner_rotated, camera.x_axis_rotated, camera.y_axis_rotated }) { *o++ = v.x; *o++ = v.y; *o++ = v.z; }
		*o++ = camera.pixel_spread_angle;
		for (const Matrix4 * m : { &camera.projection, &camera.view_projection, &camera.view_projection_prev }) { memcpy(o, m->cells, 64); o += 16; }
		return 0;
	GRT_CATCH(-1)
}
// Mesh::update and the Medium parameterisation, laid out as oracle/ref's ref_mesh_transform / ref_medium_round_trip
int grt_mesh_transform(const float * position3, const float * rotation4, float scale, const float * aabb6, float * out38) {
	GRT_TRY
		Mesh mesh("", Handle<MeshData> { 0 }, Handle<Material> { 0 });
		mesh.position = Vector3(position3[0], position3[1], position3[2]);
		mesh.rotation = Quaternion(rotation4[0], rotation4[1], rotation4[2], rotation4[3]);
		mesh.scale    = scale;
		mesh.aabb_untransformed.min = Vector3(aabb6[0], aabb6[1], aabb6[2]);
		mesh.aabb_untransformed.max = Vector3(aabb6[3], aabb6[4], aabb6[5]);
		mesh.update();
		memcpy(out38, mesh.transform.cells, 64); memcpy(out38 + 16, mesh.transform_inv.cells, 64);
		out38[32] = mesh.aabb.min.x; out38[33] = mesh.aabb.min.y; out38[34] = mesh.aabb.min.z;
		out38[35] = mesh.aabb.max.x; out38[36] = mesh.aabb.max.y; out38[37] = mesh.aabb.max.z;
		return 0;
	GRT_CATCH(-1)
}
int grt_medium_round_trip(const float * sigma_a3, const float * sigma_s3, float g, float * out12) {
	GRT_TRY
		Medium medium;
		medium.g = g;
		medium.from_sigmas(Vector3(sigma_a3[0], sigma_a3[1], sigma_a3[2]), Vector3(sigma_s3[0], sigma_s3[1], sigma_s3[2]));
		Vector3 a, s;
		medium.to_sigmas(a, s);
		const Vector3 * v[4] = { &medium.C, &medium.mfp, &a, &s };
		for (int i = 0; i < 4; i++) { out12[3 * i] = v[i]->x; out12[3 * i + 1] = v[i]->y; out12[3 * i + 2] = v[i]->z; }
		return 0;
	GRT_CATCH(-1)
}
int grt_sky_load(const char * filename, int * width, int * height, float * dst_rgba, size_t dst_floats) {
	GRT_TRY
		Sky sky;
		sky.load(filename);
		*width = sky.width; *height = sky.height;
		size_t count = sky.data.size() * 4;
		if (dst_rgba && dst_floats >= count) memcpy((void *)dst_rgba, sky.data.data(), count * sizeof(float));
		return int(count);
	GRT_CATCH(-1)
}

// One BC1 block through the product's encoder, for the parity test against stb_dxt in oracle/_ref
void grt_compress_bc1_block(const unsigned char * rgba64, unsigned char * dst8) { BlockCompression::compress_bc1_block(rgba64, dst8); }

// Stand-alone texture decode (file -> linear RGBA8 mip chain), for the decoder tests
void * grt_texture_load(const char * filename) {
	GRT_TRY
		Texture * texture = new Texture();
		if (!TextureLoader::load(filename, texture)) {
			delete texture;
			throw std::runtime_error(std::string("cannot decode texture '") + filename + "'");
		}
		return texture;
	GRT_CATCH(nullptr)
}
const unsigned char * grt_texture_data(void * texture, int * width, int * height, int * mip_levels, size_t * bytes) {
	Texture * t = (Texture *)texture;
	*width = t->width; *height = t->height; *mip_levels = t->mip_levels(); *bytes = t->texels.size();
	return t->texels.data();
}
void grt_texture_free(void * texture) { delete (Texture *)texture; }

// Stand-alone builders for the parity tests against oracle/_ref
// tris24: n x 24 floats (Triangle layout). Returns a MeshData* to query with grt_built_*.
void * grt_build_blas(const float * tris24, int n) {
	GRT_TRY
		MeshData * md = new MeshData();
		md->triangles.resize(n);
		memcpy((void *)md->triangles.data(), tris24, size_t(n) * sizeof(Triangle));
		md->bvh2 = BVH::create_sah_from_triangles(md->triangles);
		BVH8Converter(md->bvh8, md->bvh2).convert();
		BVH4Converter(md->bvh4, md->bvh2).convert();
		return md;
	GRT_CATCH(nullptr)
}
// The builder of the flattened static geometry on its own (tests): binary tree in bvh2_*, its 8-wide collapse in bvh8_*.
// threads <= 0: all hardware threads.
void * grt_build_static_bvh(const float * tris24, int n, int threads) {
	GRT_TRY
		MeshData * md = new MeshData();
		md->triangles.resize(n);
		memcpy((void *)md->triangles.data(), tris24, size_t(n) * sizeof(Triangle));
		StaticBVHBuilder::build(md->bvh2, md->triangles, threads);
		if (n > 0) BVH8Converter(md->bvh8, md->bvh2).convert();
		return md;
	GRT_CATCH(nullptr)
}
// The seating learner on a tree built by one of the two calls around it (tests): `rays` sample rays, `threads` host threads (<= 0: all); the tree's bvh8_nodes change in place.
int grt_built_learn_slot_order(void * handle, int rays, int threads) {
	GRT_TRY
		MeshData * md = (MeshData *)handle;
		if (!md) return -1;
		bvh8_learn_slot_order(md->bvh8, md->triangles, rays, threads);
		return 0;
	GRT_CATCH(-1)
}
// Experiments with the builder behind a CWBVH: spatial splits (SBVH: a triangle may sit in several leaves, `bvh8_indices` then
// repeats it) and / or the insertion optimiser, then the same 8-wide collapse.
void * grt_build_blas_variant(const float * tris24, int n, int spatial_splits, int optimize) {
	GRT_TRY
		MeshData * md = new MeshData();
		md->triangles.resize(n);
		memcpy((void *)md->triangles.data(), tris24, size_t(n) * sizeof(Triangle));
		if (spatial_splits == 2) StaticBVHBuilder::build(md->bvh2, md->triangles);
		else if (spatial_splits) SBVHBuilder(md->bvh2, md->triangles.size()).build(md->triangles);
		else                SAHBuilder (md->bvh2, md->triangles.size()).build(md->triangles);
		if (optimize) BVHOptimizer::optimize(md->bvh2);
		{ BVH8Converter converter(md->bvh8, md->bvh2); if (const char * c = getenv("GRT_PRIMITIVE_COST")) converter.primitive_cost = float(atof(c)); if (const char * c = getenv("GRT_SLOT_ASSIGNMENT")) converter.slot_assignment = atoi(c); converter.convert(); }
		if (const char * c = getenv("GRT_SLOT_LEARNING_RAYS")) bvh8_learn_slot_order(md->bvh8, md->triangles, atoi(c));
		return md;
	GRT_CATCH(nullptr)
}
// StaticBVHBuilder::presplit (early split clipping in front of the device's BLAS build) on `n` host triangles: writes up to `capacity` references
// (source triangle, 6 floats of box each) and returns how many there are (may exceed capacity: call again with room).
int grt_static_presplit(const float * tris24, int n, float limit, int * out_source, float * out_boxes, int capacity) {
	GRT_TRY
		std::vector<Triangle> triangles(static_cast<size_t>(n));
		memcpy((void *)triangles.data(), tris24, size_t(n) * sizeof(Triangle));
		std::vector<int> source; std::vector<float> boxes;
		StaticBVHBuilder::presplit(triangles, limit, source, boxes);
		int count = int(source.size());
		if (count <= capacity) { memcpy(out_source, source.data(), size_t(count) * sizeof(int)); memcpy(out_boxes, boxes.data(), size_t(count) * 6 * sizeof(float)); }
		return count;
	GRT_CATCH(-1)
}
// The binary tree of cpu_config.bvh_type (SAH or SBVH), optionally leaf-collapsed as for a
// file-loaded mesh, and its 4-wide form: query with "device_bvh2_nodes" / "device_bvh2_indices" /
// "device_bvh4_nodes".
void * grt_build_device_bvh(const float * tris24, int n, int collapse) {
	GRT_TRY
		MeshData * md = new MeshData();
		md->triangles.resize(n);
		memcpy((void *)md->triangles.data(), tris24, size_t(n) * sizeof(Triangle));
		md->from_file = collapse != 0;
		if (cpu_config.bvh_type != BVHType::SBVH) md->bvh2 = BVH::create_sah_from_triangles(md->triangles);
		BVHType type = cpu_config.bvh_type == BVHType::BVH8 ? BVHType::BVH : cpu_config.bvh_type;
		md->prepare_device_bvh(type);
		if (type != BVHType::BVH4) BVH4Converter(md->device_bvh4, md->device_bvh2).convert();
		return md;
	GRT_CATCH(nullptr)
}
const void * grt_built_array(void * mesh_data, const char * name, size_t * bytes) {
	const MeshData & md = *(MeshData *)mesh_data;
	std::string n(name);
#define RET(vec) { *bytes = (vec).size() * sizeof((vec)[0]); return (vec).data(); }
	if (n == "bvh2_nodes")   RET(md.bvh2.nodes)
	if (n == "bvh2_indices") RET(md.bvh2.indices)
	if (n == "bvh4_nodes")   RET(md.bvh4.nodes)
	if (n == "bvh4_indices") RET(md.bvh4.indices)
	if (n == "bvh8_nodes")   RET(md.bvh8.nodes)
	if (n == "bvh8_indices") RET(md.bvh8.indices)
	if (n == "device_bvh2_nodes")   RET(md.device_bvh2.nodes)
	if (n == "device_bvh2_indices") RET(md.device_bvh2.indices)
	if (n == "device_bvh4_nodes")   RET(md.device_bvh4.nodes)
#undef RET
	*bytes = 0;
	return nullptr;
}
void grt_built_free(void * mesh_data) { delete (MeshData *)mesh_data; }

} // extern "C"
