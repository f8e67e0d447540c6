#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include "Integrator.h"
#include "Exporters.h"

#include <chrono>
#include <cstring>
#include <stdexcept>

Integrator::Integrator(Scene & scene, int device_ordinal) : scene(scene) {
	if (device_ordinal >= 0) {
		int status = rt_create(device_ordinal, &ctx);
		if (status != RT_OK) {
			std::string message = rt_last_error(nullptr);
			throw std::runtime_error("rt_create(" + std::to_string(device_ordinal) + ") failed: " + message);
		}
	}
}

Integrator::~Integrator() {
	if (pending_reseat && pending_reseat->worker.joinable()) pending_reseat->worker.join();
	for (auto & retired : retired_reseats) if (retired->worker.joinable()) retired->worker.join();
	if (pending_flatten && pending_flatten->worker.joinable()) pending_flatten->worker.join();
	for (auto & retired : retired_flattens) if (retired->worker.joinable()) retired->worker.join();
	if (ctx) rt_destroy(ctx);
}

void Integrator::check(int status) const {
	if (status != RT_OK) throw std::runtime_error(std::string("device layer error: ") + rt_last_error(ctx));
}

void Integrator::require_device() const {
	if (!ctx) throw std::runtime_error("this Integrator was created without a device (host-only baking); rendering needs the HIP device layer");
}

void Integrator::gpu_init(int width, int height) {
	resize_init(width, height);
	init_materials();
	init_geometry();
	init_sky();
	init_rng();

	// reference: Integrator::cuda_init (Integrator.h:203-223)
	scene.camera.update(0.0f);
	scene.update(0.0f);
	scene.has_diffuse = scene.has_plastic = scene.has_dielectric = scene.has_conductor = scene.has_lights = false;
	invalidated_scene = invalidated_sky = invalidated_materials = invalidated_mediums = invalidated_gpu_config = invalidated_aovs = true;
}

void Integrator::gpu_free() {
	resize_free();
}

void Integrator::init_materials() {
	scene.asset_manager.wait_until_loaded();

	if (ctx) {
		const std::vector<Texture> & textures = scene.asset_manager.textures;
		std::vector<rt_texture_desc> descs(textures.size());
		for (size_t i = 0; i < textures.size(); i++) {
			descs[i].texels     = textures[i].texels.data();
			descs[i].width      = textures[i].width;
			descs[i].height     = textures[i].height;
			descs[i].mip_levels = textures[i].mip_levels();
			descs[i].lod_width  = textures[i].lod_width;
			descs[i].lod_height = textures[i].lod_height;
			descs[i].format     = RT_TEXTURE_RGBA8;
			if (!textures[i].bc1_blocks.empty()) { // block-compressed: the device keeps the 8-byte blocks and decodes per texel
				descs[i].texels = textures[i].bc1_blocks.data();
				descs[i].format = RT_TEXTURE_BC1;
			}
		}
		check(rt_set_texture_expansion(ctx, cpu_config.expand_block_compressed_textures ? 1 : 0));
		check(rt_upload_textures(ctx, descs.data(), descs.size()));
	}
}

// ---- flattened static geometry: who joins, their triangles in world space, the tree ------------------------------------------------

// Every instance that has not been seen moving -- merge_static 3: only those with the identity transform, whose copies are the
// original triangles bit for bit (a transformed instance's copies are its triangles taken to world space) -- as far as the copies
// fit the budget (Config.h static_mesh_copy_limit_mb / static_copy_budget_mb). Scene order.
std::vector<int> Integrator::flatten_candidates() const {
	const std::vector<MeshData> & mesh_datas = scene.asset_manager.mesh_datas;
	size_t mesh_count = scene.meshes.size();
	std::vector<int> members;
	const double bytes_per_copy = 176.0;
	auto moved = [&](size_t i) { return i < instance_has_moved.size() && instance_has_moved[i]; };
	std::vector<int> uses(mesh_datas.size(), 0);
	for (size_t i = 0; i < mesh_count; i++) if (!moved(i)) uses[size_t(scene.meshes[i].mesh_data_handle.handle)]++;
	double budget = double(cpu_config.static_copy_budget_mb) * 1048576.0;
	for (size_t i = 0; i < mesh_count; i++) {
		size_t handle = size_t(scene.meshes[i].mesh_data_handle.handle);
		double one_copy = double(mesh_datas[handle].triangles.size()) * bytes_per_copy;
		bool joins = !moved(i) && (cpu_config.merge_static != 3 || scene.meshes[i].has_identity_transform())
		          && double(uses[handle] - 1) * one_copy <= double(cpu_config.static_mesh_copy_limit_mb) * 1048576.0   // the copies beyond the first: what instancing saves
		          && one_copy <= budget;
		if (joins) { budget -= one_copy; members.push_back(int(i)); }
	}
	return members;
}

std::vector<Triangle> Integrator::world_triangles_of(const std::vector<int> & members, std::vector<int> * source_member, std::vector<int> * source_triangle) const {
	const std::vector<MeshData> & mesh_datas = scene.asset_manager.mesh_datas;
	std::vector<Triangle> world;
	for (size_t j = 0; j < members.size(); j++) {
		const Mesh & mesh = scene.meshes[size_t(members[j])];
		int handle = mesh.mesh_data_handle.handle;
		bool identity = mesh.has_identity_transform();
		Matrix4 to_world = Matrix4::create_translation(mesh.position) * Matrix4::create_rotation(mesh.rotation) * Matrix4::create_scale(mesh.scale);   // as Mesh::update
		for (size_t t = 0; t < mesh_datas[size_t(handle)].triangles.size(); t++) {
			if (source_member) { source_member->push_back(int(j)); source_triangle->push_back(mesh_data_triangle_offsets[size_t(handle)] + int(t)); }
			Triangle triangle = mesh_datas[size_t(handle)].triangles[t];
			if (!identity) {
				triangle.position_0 = Matrix4::transform_position(to_world, triangle.position_0);
				triangle.position_1 = Matrix4::transform_position(to_world, triangle.position_1);
				triangle.position_2 = Matrix4::transform_position(to_world, triangle.position_2);
			}
			world.push_back(triangle);
		}
	}
	return world;
}

static int processors_this_process_may_use();
// What a worker beside the frame loop may use: a quarter of the processors this process can keep busy, eight at most, at the lowest scheduling priority (on Linux a
// thread's nice value is its own and is inherited by the threads it starts).
static int background_threads() { setpriority(PRIO_PROCESS, id_t(syscall(SYS_gettid)), 19); return std::max(1, std::min(8, processors_this_process_may_use() / 4)); }

// SAH object + spatial splits on all host threads, the reference's 8-wide collapse, breadth-first node order. A pure function of
// its input and the configuration (it runs on a worker thread when a flattened instance has started to move).
// What the seating of the flattened tree's children is trained on (SlotOrder.cpp): the camera's rays as camera_generate_ray forms them, as the camera stands now.
SlotLearningView Integrator::slot_learning_view() const {
	SlotLearningView view;
	if (!cpu_config.static_slot_learning_viewpoint) return view;   // (width 0: no view)
	const Camera & c = scene.camera;
	// (rotated here, as Camera::update does: the first flatten of an integrator runs before its first camera update)
	view.position = c.position; view.bottom_left_corner = c.rotation * c.bottom_left_corner; view.x_axis = c.rotation * c.x_axis; view.y_axis = c.rotation * c.y_axis;
	view.width = int(c.screen_width); view.height = int(c.screen_height);
	return view;
}

// `threads`: 0 = all host threads (a build inside update()); a worker beside the frame loop passes its budget (background_threads)
static void build_flattened_tree(const std::vector<Triangle> & world, BVH8 & wide, int & top_nodes, const SlotLearningView & view, int threads = 0) {
	BVH2 binary;
	if (cpu_config.merge_static == 2) binary = BVH::create_sah_from_triangles(world);   // the per-mesh builder, for comparison
	else                              StaticBVHBuilder::build(binary, world, threads);  // spatial splits
	BVH8Converter converter(wide, binary);
	converter.primitive_cost = cpu_config.static_primitive_cost;
	converter.slot_assignment = cpu_config.static_slot_assignment;
	converter.convert();
	if (cpu_config.static_slot_learning_rays > 0) bvh8_learn_slot_order(wide, world, cpu_config.static_slot_learning_rays, threads, &view);
	top_nodes = std::min(bvh8_order_breadth_first(wide, 2), 64);   // levels 0..2 (at most 1 + 8 + 64 nodes) come first: the part of the tree every ray walks is one contiguous run
}

// Lets go of the current background build without waiting for it: a build that is still running moves to retired_flattens and is
// joined once it has finished (or when the integrator goes).
void Integrator::drop_flatten_worker() {
	for (size_t i = 0; i < retired_flattens.size();) {
		if (retired_flattens[i]->ready.load()) { if (retired_flattens[i]->worker.joinable()) retired_flattens[i]->worker.join(); retired_flattens.erase(retired_flattens.begin() + long(i)); } else i++;
	}
	if (!pending_flatten) return;
	if (pending_flatten->ready.load()) { if (pending_flatten->worker.joinable()) pending_flatten->worker.join(); pending_flatten.reset(); }
	else retired_flattens.push_back(std::move(pending_flatten));
}

// Do the members the background build was started for still stand where they stood? (The interim layout does not flatten, so nobody
// else watches them.) One that moved is a mover for good, like a flattened instance that starts to move.
bool Integrator::pending_flatten_is_current() {
	bool current = true;
	for (size_t j = 0; j < pending_flatten->members.size(); j++) {
		const Mesh & mesh = scene.meshes[size_t(pending_flatten->members[j])];
		const StaticGeometry::Pose & pose = pending_flatten->poses[j];
		if (memcmp(&mesh.position, &pose.position, sizeof(Vector3)) || memcmp(&mesh.rotation, &pose.rotation, sizeof(Quaternion)) || mesh.scale != pose.scale) {
			instance_has_moved[size_t(pending_flatten->members[j])] = 1;
			current = false;
		}
	}
	return current;
}

// The tree of the instances that still stand still, built beside the frame loop.
void Integrator::start_flatten_worker() {
	drop_flatten_worker();   // (a build for a member set that is out of date by now: it has to finish before its input can go)
	std::vector<int> members = flatten_candidates();
	size_t triangles = 0;
	for (int member : members) triangles += scene.asset_manager.mesh_datas[size_t(scene.meshes[size_t(member)].mesh_data_handle.handle)].triangles.size();
	if (members.size() < 2 || triangles == 0) return;   // nothing left to flatten
	pending_flatten = std::make_unique<PendingFlatten>();
	pending_flatten->members = members;
	for (int member : members) { const Mesh & mesh = scene.meshes[size_t(member)]; pending_flatten->poses.push_back({ mesh.position, mesh.rotation, mesh.scale }); }
	pending_flatten->world = world_triangles_of(members, &pending_flatten->source_member, &pending_flatten->source_triangle);
	PendingFlatten * job = pending_flatten.get();
	// (the worker reads the reference part of the staged arrays -- the originals of the triangles it copies -- which nothing rewrites
	// while it runs: init_geometry waits for every worker before it restages that part)
	const std::vector<DeviceTriangle> * originals = &aggregated_triangles; const std::vector<int> * device_index = &reverse_indices;
	const SlotLearningView view = slot_learning_view();   // (as things stand now: the worker must not read the scene)
	job->worker = std::thread([job, originals, device_index, view] {
		auto started = std::chrono::steady_clock::now();
		try {
			build_flattened_tree(job->world, job->wide, job->top_nodes, view, background_threads());
			size_t copies = job->wide.indices.size();
			job->copy_triangles.resize(copies); job->copy_member.resize(copies); job->copy_original.resize(copies);
			for (size_t c = 0; c < copies; c++) {
				size_t source = size_t(job->wide.indices[c]);
				int original = (*device_index)[size_t(job->source_triangle[source])];
				const Triangle & placed = job->world[source];
				DeviceTriangle & copy = job->copy_triangles[c];
				copy = (*originals)[size_t(original)];
				copy.position_0      = placed.position_0;
				copy.position_edge_1 = placed.position_1 - placed.position_0;
				copy.position_edge_2 = placed.position_2 - placed.position_0;
				job->copy_member[c] = job->source_member[source]; job->copy_original[c] = original;
			}
		} catch (...) { job->failed = true; }
		job->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count();
		job->ready.store(true);
	});
}

// ---- seating the flattened tree again, beside the frame loop (Integrator.h: PendingReseat) -----------------------------------------------------------
void Integrator::drop_reseat_worker() {
	for (size_t i = 0; i < retired_reseats.size();) {
		if (retired_reseats[i]->ready.load()) { if (retired_reseats[i]->worker.joinable()) retired_reseats[i]->worker.join(); retired_reseats.erase(retired_reseats.begin() + long(i)); } else i++;
	}
	if (!pending_reseat) return;
	if (pending_reseat->ready.load()) { if (pending_reseat->worker.joinable()) pending_reseat->worker.join(); pending_reseat.reset(); }
	else retired_reseats.push_back(std::move(pending_reseat));
}

// A copy of the flattened tree as it stands in the staged arrays (host- or device-built: all it takes is the nodes and the triangles of its leaves) goes to a
// worker that seats its children for the camera as it stands now. The copy is numbered breadth-first from the root; a node's inner children stay one
// contiguous run in their order, so a record the learner moves within its run has ONE place to go back to.
// Processors this process can really keep busy: the affinity mask, and a control group's CPU bandwidth limit on top of it (a container that shows 256 logical
// processors may be allowed 16). The limit matters beside a frame loop: when a group's threads use up its quota of a 100 ms period, EVERY thread of the group is
// stopped until the next period -- the thread that submits frames included, however low the priority of the threads that spent the quota (round 6: one of the
// reference's nine points of view at 3.7-5.5 ms per step for 1.25 while the seating worker ran on 64 threads; tools/gpu_jobs/r06_run31.sh).
static int processors_this_process_may_use() {
	int count = int(std::max(1u, std::thread::hardware_concurrency()));
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) count = std::min(count, std::max(1, CPU_COUNT(&set)));
	auto limit = [&](double quota, double period) { if (quota > 0.0 && period > 0.0) count = std::min(count, std::max(1, int(quota / period + 0.5))); };
	if (FILE * f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // control groups v2: "<quota | max> <period>"
		char quota[32] = ""; double period = 0.0;
		if (fscanf(f, "%31s %lf", quota, &period) == 2 && strcmp(quota, "max") != 0) limit(atof(quota), period);
		fclose(f);
	} else {   // v1
		double quota = -1.0, period = 0.0;
		if (FILE * q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(q, "%lf", &quota) != 1) quota = -1.0; fclose(q); }
		if (FILE * q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(q, "%lf", &period) != 1) period = 0.0; fclose(q); }
		limit(quota, period);
	}
	return count;
}

void Integrator::start_reseat_worker(bool beside_frame_loop) {
	const StaticGeometry & flat = static_geometry;
	if (!flat.active || cpu_config.bvh_type != BVHType::BVH8 || cpu_config.static_slot_learning_rays <= 0) return;
	if (size_t(flat.root) >= aggregated_bvh_nodes_8.size()) return;
	drop_reseat_worker();
	auto job = std::make_unique<PendingReseat>();
	job->generation = geometry_generation; job->camera_position = scene.camera.position;
	job->absolute.push_back(unsigned(flat.root));
	// (the leaf positions the tree names: the host's builders give it the tail of the triangle array; the device's builder emits the leaves of all trees of a
	// launch level by level, so the run from its first to its last position also holds other trees' triangles -- never named by this tree, carried along)
	size_t first_position = aggregated_triangles.size(), end_position = 0;
	for (size_t k = 0; k < job->absolute.size(); k++) {
		if (job->absolute[k] >= aggregated_bvh_nodes_8.size()) return;
		BVHNode8 node = aggregated_bvh_nodes_8[job->absolute[k]];
		const unsigned children = unsigned(__builtin_popcount(unsigned(node.imask)));
		const unsigned local_base = unsigned(job->absolute.size());
		for (unsigned c = 0; c < children; c++) job->absolute.push_back(node.base_index_child + c);
		if (children) node.base_index_child = local_base;
		for (int s = 0; s < 8; s++) if (!((node.imask >> s) & 1) && node.meta[s]) {
			const size_t first = size_t(node.base_index_triangle) + (node.meta[s] & 31u), count = size_t(__builtin_popcount(unsigned(node.meta[s]) >> 5));
			first_position = std::min(first_position, first); end_position = std::max(end_position, first + count);
		}
		job->tree.nodes.push_back(node);
	}
	if (first_position >= end_position || end_position > aggregated_triangles.size()) return;
	job->triangle_base = first_position;
	for (BVHNode8 & node : job->tree.nodes) node.base_index_triangle -= unsigned(job->triangle_base);   // (modulo 2^32 for a node without leaves, whose base means nothing: install adds it back)
	const size_t copies = end_position - first_position;
	job->tree.indices.resize(copies); job->triangles.resize(copies);
	for (size_t i = 0; i < copies; i++) {
		const DeviceTriangle & t = aggregated_triangles[job->triangle_base + i];
		job->tree.indices[i] = int(i);
		job->triangles[i].position_0 = t.position_0; job->triangles[i].position_1 = t.position_0 + t.position_edge_1; job->triangles[i].position_2 = t.position_0 + t.position_edge_2;
	}
	const SlotLearningView view = slot_learning_view();   // (as things stand now: the worker must not read the scene)
	const int rays = cpu_config.static_slot_learning_rays;
	PendingReseat * raw = job.get();
	raw->worker = std::thread([raw, view, rays, beside_frame_loop] {
		auto started = std::chrono::steady_clock::now();
		// (beside the frame loop the learner must not take the processor -- or the control group's quota -- from the thread that submits frames: background_threads)
		const int threads = beside_frame_loop ? background_threads() : 0;
		try { bvh8_learn_slot_order(raw->tree, raw->triangles, rays, threads, &view); } catch (...) { raw->failed = true; }
		raw->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count();
		raw->ready.store(true);
	});
	pending_reseat = std::move(job);
}

// The worker is done: its nodes go back to the places they were copied from (children runs and triangle bases in the staged arrays' own numbering again) and
// to the device (rt_update_nodes: between two frames). False: nothing went in (not ready, failed, or the geometry has been staged again meanwhile).
bool Integrator::install_reseat() {
	if (!pending_reseat || !pending_reseat->ready.load()) return false;
	if (pending_reseat->worker.joinable()) pending_reseat->worker.join();
	std::unique_ptr<PendingReseat> job = std::move(pending_reseat);
	const StaticGeometry & flat = static_geometry;
	if (job->failed || job->generation != geometry_generation || !flat.active || job->absolute.empty() || job->absolute[0] != unsigned(flat.root)) return false;
	unsigned lowest = ~0u, highest = 0u;
	for (size_t i = 0; i < job->tree.nodes.size(); i++) {
		BVHNode8 node = job->tree.nodes[i];
		if (node.imask) node.base_index_child = job->absolute[node.base_index_child];
		node.base_index_triangle += unsigned(job->triangle_base);
		aggregated_bvh_nodes_8[job->absolute[i]] = node;
		lowest = std::min(lowest, job->absolute[i]); highest = std::max(highest, job->absolute[i]);
	}
	// (a host-built tree is one run of nodes; the device's builder interleaves the trees of a launch level by level: the run in between is re-sent as it is)
	if (ctx) check(rt_update_nodes(ctx, &aggregated_bvh_nodes_8[lowest], lowest, size_t(highest - lowest) + 1));
	seated_for_position = job->camera_position; seated_for_a_viewpoint = cpu_config.static_slot_learning_viewpoint != 0;
	reseats_completed++; last_reseat_seconds = job->seconds;
	return true;
}

// Concatenates all BLASes into one node array / one triangle array
// (reference: Integrator::init_geometry, Integrator.cpp:101-283):
//   * triangles are stored permuted by the BLAS `indices`, as pre-subtracted edges
//   * node slots [0, 2*mesh_count) are reserved for the per-frame TLAS
//   * each BLAS' child / triangle base offsets are rebased into the shared arrays
void Integrator::init_geometry() {
	// A tree that the worker thread built for the instances that stand still (build_tlas) is being installed: the reference part
	// of the staged arrays -- every mesh's triangles and nodes, 40 of the 60 ms this function takes for Sponza -- is what it was,
	// only the copies and the flattened tree's nodes behind it are replaced
	const bool only_the_flattened_part = pending_flatten && pending_flatten->ready.load() && !pending_flatten->failed && staged_index_total > 0
	                                  && cpu_config.bvh_type == BVHType::BVH8 && cpu_config.device_blas <= 0 && aggregated_triangles.size() >= staged_index_total
	                                  && aggregated_bvh_nodes_8.size() >= staged_node_total;
	if (!only_the_flattened_part) {
	if (pending_flatten && pending_flatten->worker.joinable() && !pending_flatten->ready.load()) pending_flatten->worker.join();   // (they read what is restaged below)
	for (auto & retired : retired_flattens) if (retired->worker.joinable()) retired->worker.join();
	scene.asset_manager.wait_until_loaded();
	scene.asset_manager.prepare_device_bvhs(cpu_config.bvh_type);
	for (Mesh & mesh : scene.meshes) mesh.calc_aabb(scene);
	}

	const std::vector<MeshData> & mesh_datas = scene.asset_manager.mesh_datas;
	size_t mesh_data_count = mesh_datas.size();
	size_t mesh_count = scene.meshes.size();

	mesh_data_bvh_offsets     .resize(mesh_data_count);
	mesh_data_triangle_offsets.resize(mesh_data_count);
	mesh_data_index_offsets   .resize(mesh_data_count);

	bool use_bvh8 = cpu_config.bvh_type == BVHType::BVH8;

	size_t node_total     = 2 * mesh_count;
	size_t triangle_total = 0;
	size_t index_total    = 0;
	for (size_t m = 0; m < mesh_data_count; m++) {
		mesh_data_bvh_offsets     [m] = int(node_total);
		mesh_data_triangle_offsets[m] = int(triangle_total);
		mesh_data_index_offsets   [m] = int(index_total);
		node_total     += use_bvh8 ? mesh_datas[m].bvh8.nodes.size()   : mesh_datas[m].device_bvh2.nodes.size();
		triangle_total += mesh_datas[m].triangles.size();
		index_total    += use_bvh8 ? mesh_datas[m].bvh8.indices.size() : mesh_datas[m].device_bvh2.indices.size();
	}

	if (only_the_flattened_part) aggregated_triangles.resize(index_total);
	else {
	aggregated_triangles.assign(index_total, DeviceTriangle());
	reverse_indices.assign(triangle_total, 0);
	}
	for (size_t m = 0; m < mesh_data_count && !only_the_flattened_part; m++) {
		const MeshData & md = mesh_datas[m];
		const std::vector<int> & order = use_bvh8 ? md.bvh8.indices : md.device_bvh2.indices; // a spatial-split tree lists a triangle once per leaf that holds a part of it
		for (size_t i = 0; i < order.size(); i++) {
			const Triangle & t = md.triangles[order[i]];
			DeviceTriangle & d = aggregated_triangles[mesh_data_index_offsets[m] + i];
			d.position_0       = t.position_0;
			d.position_edge_1  = t.position_1 - t.position_0;
			d.position_edge_2  = t.position_2 - t.position_0;
			d.normal_0         = t.normal_0;
			d.normal_edge_1    = t.normal_1 - t.normal_0;
			d.normal_edge_2    = t.normal_2 - t.normal_0;
			d.tex_coord_0      = t.tex_coord_0;
			d.tex_coord_edge_1 = t.tex_coord_1 - t.tex_coord_0;
			d.tex_coord_edge_2 = t.tex_coord_2 - t.tex_coord_0;
			reverse_indices[mesh_data_triangle_offsets[m] + order[i]] = mesh_data_index_offsets[m] + int(i);
		}
	}

	mesh_bvh_root_indices.assign(mesh_count, 0);
	mesh_material_ids    .assign(mesh_count, 0);
	mesh_transforms      .assign(mesh_count, Matrix3x4());
	mesh_transforms_inv  .assign(mesh_count, Matrix3x4());
	mesh_transforms_prev .assign(mesh_count, Matrix3x4());

	tlas_raw.indices.resize(mesh_count);
	tlas_raw.nodes  .resize(mesh_count * 2);
	tlas_builder = std::make_unique<SAHBuilder>(tlas_raw, mesh_count);
	tlas_converter = std::make_unique<BVH8Converter>(tlas, tlas_raw);

	if (cpu_config.bvh_type == BVHType::BVH4) { // reference: Integrator.cpp:216-251
		aggregated_bvh_nodes_4.assign(node_total, BVHNode4());
		memset((void *)aggregated_bvh_nodes_4.data(), 0, node_total * sizeof(BVHNode4));
		for (size_t m = 0; m < mesh_data_count; m++) {
			const std::vector<BVHNode4> & nodes = mesh_datas[m].device_bvh4.nodes;
			BVHNode4 * dst = aggregated_bvh_nodes_4.data() + mesh_data_bvh_offsets[m];
			for (size_t n = 0; n < nodes.size(); n++) {
				dst[n] = nodes[n];
				int child_count = dst[n].get_child_count();
				for (int c = 0; c < child_count; c++) {
					if (dst[n].is_leaf(c)) dst[n].get_index(c) += mesh_data_index_offsets[m];
					else                   dst[n].get_index(c) += mesh_data_bvh_offsets[m];
				}
			}
		}
		if (ctx) check(rt_upload_geometry_bvh4(ctx, aggregated_triangles.data(), aggregated_triangles.size(), aggregated_bvh_nodes_4.data(), aggregated_bvh_nodes_4.size()));
		if (ctx) check(rt_set_bvh_type(ctx, 4));
		return;
	}
	if (use_bvh8) {
		staged_index_total = index_total; staged_node_total = node_total;
		geometry_generation++;   // (a seating in the making was copied from what is being replaced: install_reseat lets it go)
		if (only_the_flattened_part) aggregated_bvh_nodes_8.resize(node_total);
		else {
		aggregated_bvh_nodes_8.assign(node_total, BVHNode8());
		memset(aggregated_bvh_nodes_8.data(), 0, node_total * sizeof(BVHNode8));
		}
		for (size_t m = 0; m < mesh_data_count && !only_the_flattened_part; m++) {
			const std::vector<BVHNode8> & nodes = mesh_datas[m].bvh8.nodes;
			BVHNode8 * dst = aggregated_bvh_nodes_8.data() + mesh_data_bvh_offsets[m];
			for (size_t n = 0; n < nodes.size(); n++) {
				dst[n] = nodes[n];
				dst[n].base_index_triangle += unsigned(mesh_data_index_offsets[m]);
				dst[n].base_index_child    += unsigned(mesh_data_bvh_offsets[m]);
			}
		}
		// Flattened static geometry: copies of the identity instances' triangles behind everything else, one more tree over them
		StaticGeometry & flat = static_geometry;
		flat = StaticGeometry(); alias_mesh_ids.clear(); alias_triangle_ids.clear();
		bool build_on_device = ctx && cpu_config.device_blas > 0;
		std::vector<int> source_member, source_triangle;   // per triangle of the members, in member order: its member, its index among all original triangles
		std::vector<int> copy_source;                      // per copy, in device order: which of those it copies
		std::vector<float> copy_boxes;                     // device build with early split clipping: the box of the piece each copy stands for (6 floats)
		instance_has_moved.resize(mesh_count, 0);
		if (cpu_config.merge_static > 0 && !wants_device_tlas()) {
			flat.members = flatten_candidates();
			std::vector<char> is_member(mesh_count, 0);
			for (int member : flat.members) is_member[size_t(member)] = 1;
			for (size_t i = 0; i < mesh_count; i++) if (!is_member[i]) flat.movers.push_back(int(i));
			size_t triangles_to_copy = 0;
			for (int member : flat.members) triangles_to_copy += mesh_datas[size_t(scene.meshes[size_t(member)].mesh_data_handle.handle)].triangles.size();
			if (flat.members.size() < 2 || triangles_to_copy == 0) { flat.members.clear(); flat.movers.clear(); }   // nothing to gain / nothing to build a tree over
		}
		std::vector<Triangle> world;   // the members' triangles in world space, in member order
		if (!flat.members.empty()) {
			flat.member_poses.resize(flat.members.size());
			for (size_t j = 0; j < flat.members.size(); j++) {
				const Mesh & mesh = scene.meshes[flat.members[j]];
				flat.member_poses[j] = { mesh.position, mesh.rotation, mesh.scale };
			}
			// (the worker's input and what it derived from it are taken over as they are: build_tlas has checked that its members stand
			// where they stood when it started, pending_flatten_is_current)
			PendingFlatten * prebuilt = !build_on_device && pending_flatten && pending_flatten->ready.load() && !pending_flatten->failed && pending_flatten->members == flat.members
			                         && pending_flatten_is_current() ? pending_flatten.get() : nullptr;
			if (prebuilt) { world.swap(prebuilt->world); source_member.swap(prebuilt->source_member); source_triangle.swap(prebuilt->source_triangle); }
			else world = world_triangles_of(flat.members, &source_member, &source_triangle);
			if (build_on_device) {
				// the device's builder cuts at Morton bits and has no spatial splits: the few triangles that span a good part of the scene are cut here, into
				// references with boxes of their own (one copy per piece; the copies name their original as every copy does)
				float longest = 0.0f;
				if (cpu_config.device_presplit > 0.0f) {
					AABB all = AABB::create_empty();
					for (const Triangle & t : world) { all.expand(t.position_0); all.expand(t.position_1); all.expand(t.position_2); }
					for (int d = 0; d < 3; d++) longest = std::max(longest, all.max[d] - all.min[d]);
				}
				if (longest > 0.0f) {
					StaticBVHBuilder::presplit(world, cpu_config.device_presplit * longest, copy_source, copy_boxes);
					// The memory policy was decided on the triangles (flatten_candidates: 176 bytes per copy against static_copy_budget_mb); a cut triangle is up to
					// 64 copies. Pieces that would take the copies past the budget are not worth it: the references go in uncut (advisor finding, round 5).
					if (double(copy_source.size()) * 176.0 > double(cpu_config.static_copy_budget_mb) * 1048576.0) { copy_source.clear(); copy_boxes.clear(); longest = 0.0f; }
				}
				if (!(longest > 0.0f)) {
					copy_source.resize(source_member.size());
					for (size_t c = 0; c < copy_source.size(); c++) copy_source[c] = int(c);
				}
			} else {
				BVH8 wide;
				if (prebuilt) {   // the worker thread built exactly this tree while the frame loop went on (build_tlas, "a member moved")
					wide.nodes.swap(prebuilt->wide.nodes); wide.indices.swap(prebuilt->wide.indices);
					flat.top_nodes = prebuilt->top_nodes; flat.build_seconds = prebuilt->build_seconds;
					reflattens_completed++;
				} else {
					auto started = std::chrono::steady_clock::now();
					build_flattened_tree(world, wide, flat.top_nodes, slot_learning_view());
					flat.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count();
				}
				copy_source = wide.indices;
				flat.root = int(node_total);
				aggregated_bvh_nodes_8.resize(node_total + wide.nodes.size());
				for (size_t n = 0; n < wide.nodes.size(); n++) {
					BVHNode8 & dst = aggregated_bvh_nodes_8[node_total + n];
					dst = wide.nodes[n];
					dst.base_index_triangle += unsigned(index_total);
					dst.base_index_child    += unsigned(node_total);
				}
			}
			size_t copies = copy_source.size();   // more than the members have triangles where spatial splits cut some of them
			// A retired worker (a build whose members moved while it ran, start_flatten_worker) may still be reading the reference part of
			// this array through a raw pointer. Growing within the capacity leaves that part where it is; a reallocation would pull it
			// from under the worker, so such a worker is waited for first (only then: the install frame otherwise never waits).
			if (index_total + copies > aggregated_triangles.capacity())
				for (auto & retired : retired_flattens) if (retired->worker.joinable()) retired->worker.join();
			aggregated_triangles.resize(index_total + copies);
			alias_mesh_ids.assign(index_total + copies, -1); alias_triangle_ids.assign(index_total + copies, -1);
			if (prebuilt && prebuilt->copy_triangles.size() == copies) {   // ... the copies too (the worker filled them in: 15 ms for Sponza)
				memcpy((void *)&aggregated_triangles[index_total], prebuilt->copy_triangles.data(), copies * sizeof(DeviceTriangle));
				memcpy(&alias_triangle_ids[index_total], prebuilt->copy_original.data(), copies * sizeof(int));
				for (size_t c = 0; c < copies; c++) alias_mesh_ids[index_total + c] = flat.leaves() + prebuilt->copy_member[c];
			} else
			for (size_t c = 0; c < copies; c++) {
				int original = reverse_indices[source_triangle[copy_source[c]]];
				const Triangle & placed = world[size_t(copy_source[c])];
				DeviceTriangle & copy = aggregated_triangles[index_total + c];
				copy = aggregated_triangles[original];               // only the positions of a copy are ever read (traversal); the rest rides along
				copy.position_0      = placed.position_0;            // (identity instances: the original's bits)
				copy.position_edge_1 = placed.position_1 - placed.position_0;
				copy.position_edge_2 = placed.position_2 - placed.position_0;
				alias_mesh_ids     [index_total + c] = flat.leaves() + source_member[copy_source[c]];   // its member's row behind the TLAS leaves' rows
				alias_triangle_ids [index_total + c] = original;
			}
			flat.built = flat.active = true;
			// what the tree's children are seated for: the host's builder has just seated them for the camera as it stands (or as it stood when the worker
			// started: near enough); the device's collapse knows no seating -- update() sends such a tree to the reseat worker
			seated_for_position = scene.camera.position;
			seated_for_a_viewpoint = !build_on_device && cpu_config.static_slot_learning_viewpoint != 0 && cpu_config.static_slot_learning_rays > 0;
			flattened_tree_needs_seating = build_on_device && cpu_config.static_slot_learning_rays > 0;
			{   // how far the flattened geometry reaches (finite vertices only): the yardstick of static_reseat_distance
				Vector3 lo(+INFINITY), hi(-INFINITY);
				for (size_t c = 0; c < copies; c++) {
					const DeviceTriangle & t = aggregated_triangles[index_total + c];
					for (const Vector3 & p : { t.position_0, t.position_0 + t.position_edge_1, t.position_0 + t.position_edge_2 })
						if (std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z)) { lo = Vector3::min(lo, p); hi = Vector3::max(hi, p); }
				}
				flat.diagonal = lo.x <= hi.x ? Vector3::length(hi - lo) : 0.0f;
			}
			flat.copy_bytes = copies * (sizeof(DeviceTriangle) + 48 + 8) + (aggregated_bvh_nodes_8.size() - node_total) * sizeof(BVHNode8);
		}
		if (build_on_device) {
			// the trees are built on the device from the triangles alone (their order -- the host trees' leaf order -- is as good as
			// any); what names triangles or nodes by index on the host follows: BLAS roots, the original-triangle -> device-triangle
			// table behind the light tables, and the host's views of the device arrays (pixel queries, exporters, the checker)
			size_t tree_count = mesh_data_count + (flat.built ? 1 : 0), triangle_count = aggregated_triangles.size();
			std::vector<int> first(tree_count + 1), roots(tree_count), position(triangle_count);
			for (size_t m = 0; m < mesh_data_count; m++) first[m] = mesh_data_index_offsets[m];
			first[mesh_data_count] = int(index_total);
			first[tree_count]      = int(triangle_count);
			size_t built_nodes = 0;
			if (flat.built && copy_boxes.size() == 6 * (triangle_count - index_total)) check(rt_set_build_boxes(ctx, copy_boxes.data(), index_total, triangle_count - index_total));
			check(rt_build_geometry(ctx, aggregated_triangles.data(), triangle_count, first.data(), tree_count, 2 * mesh_count, roots.data(), position.data(), &built_nodes, &device_blas_build_ms));
			for (size_t m = 0; m < mesh_data_count; m++) mesh_data_bvh_offsets[m] = roots[m];
			for (int & device_index : reverse_indices) device_index = position[device_index];
			if (flat.built) { // the build moved every triangle, copies and originals alike
				flat.root = roots[mesh_data_count];
				std::vector<int> moved_mesh(triangle_count, -1), moved_triangle(triangle_count, -1);
				for (size_t i = 0; i < triangle_count; i++) if (alias_mesh_ids[i] >= 0) { moved_mesh[size_t(position[i])] = alias_mesh_ids[i]; moved_triangle[size_t(position[i])] = position[size_t(alias_triangle_ids[i])]; }
				alias_mesh_ids.swap(moved_mesh); alias_triangle_ids.swap(moved_triangle);
			}
			aggregated_bvh_nodes_8.assign(built_nodes, BVHNode8());
			check(rt_read_geometry(ctx, aggregated_triangles.data(), aggregated_bvh_nodes_8.data()));
		} else
		if (ctx) check(rt_upload_geometry(ctx, aggregated_triangles.data(), aggregated_triangles.size(), aggregated_bvh_nodes_8.data(), aggregated_bvh_nodes_8.size()));
		if (flat.built) {
			if (ctx) check(rt_upload_triangle_aliases(ctx, alias_mesh_ids.data(), alias_triangle_ids.data()));
			tlas_builder = std::make_unique<SAHBuilder>(tlas_raw, size_t(flat.leaves()));
		}
	} else {
		aggregated_bvh_nodes_2.assign(node_total, BVHNode2());
		memset(aggregated_bvh_nodes_2.data(), 0, node_total * sizeof(BVHNode2));
		for (size_t m = 0; m < mesh_data_count; m++) {
			const std::vector<BVHNode2> & nodes = mesh_datas[m].device_bvh2.nodes;
			BVHNode2 * dst = aggregated_bvh_nodes_2.data() + mesh_data_bvh_offsets[m];
			for (size_t n = 0; n < nodes.size(); n++) {
				dst[n] = nodes[n];
				if (dst[n].is_leaf()) dst[n].first += mesh_data_index_offsets[m];
				else                  dst[n].left  += mesh_data_bvh_offsets[m];
			}
		}
		if (ctx) check(rt_upload_geometry_bvh2(ctx, aggregated_triangles.data(), aggregated_triangles.size(), aggregated_bvh_nodes_2.data(), aggregated_bvh_nodes_2.size()));
	}
	if (ctx) check(rt_set_bvh_type(ctx, use_bvh8 ? 8 : 2));
}

void Integrator::init_sky() {
	if (ctx) check(rt_set_sky(ctx, &scene.sky.data[0].x, scene.sky.width, scene.sky.height, scene.sky.scale));
}

void Integrator::init_rng() {
	pmj_samples = PMJ::generate();
	blue_noise  = BlueNoise::load();
	if (ctx) check(rt_upload_rng(ctx, pmj_samples.data(), blue_noise.data()));
}

// Per-frame TLAS over the mesh AABBs; every per-instance table is written in TLAS
// leaf order, which is what `mesh_id` means on the device (reference: Integrator.cpp:399-430).
bool Integrator::wants_device_tlas() const {
	if (!ctx || cpu_config.bvh_type != BVHType::BVH8 || scene.meshes.empty() || scene.meshes.size() > 4096) return false;
	// auto: scenes that rebuild every frame AND are large enough for the host build to matter -- at 441 instances the host's
	// SAH + CWBVH build is 0.05 ms and overlaps the GPU while the one-workgroup kernel has to find a free CU beside the
	// persistent traversal launches (frames 3.40 -> 3.72 ms); at 4 000 instances the host's share of a frame drops from
	// 3.5 to 1.1 ms at equal frame time (profiles/r02_animated_scene*.txt)
	return cpu_config.device_tlas > 0 || (cpu_config.device_tlas < 0 && cpu_config.enable_scene_update && scene.meshes.size() >= 1024);
}

// The device-built TLAS as host arrays (TLAS nodes in the aggregated node array, the instance tables in TLAS order,
// tlas.indices): what pixel queries and the parity checker read. One synchronous read-back, only when asked for.
void Integrator::sync_host_view_of_device_tlas() {
	if (!tlas_on_device || !tlas_host_view_stale) return;
	size_t mesh_count = scene.meshes.size();
	tlas.indices.resize(mesh_count);
	tlas.nodes.assign(2 * mesh_count, BVHNode8());
	int node_count = 0;
	check(rt_read_tlas(ctx, tlas.indices.data(), tlas.nodes.data(), tlas.nodes.size(), &node_count));
	tlas.nodes.resize(size_t(node_count));
	memcpy(aggregated_bvh_nodes_8.data(), tlas.nodes.data(), tlas.nodes.size() * sizeof(BVHNode8));
	for (size_t i = 0; i < mesh_count; i++) {
		int src = tlas.indices[i];
		mesh_bvh_root_indices[i] = scene_order_roots[src];
		mesh_material_ids[i]     = scene_order_materials[src];
		mesh_transforms[i] = scene_order_transforms[src]; mesh_transforms_inv[i] = scene_order_transforms_inv[src]; mesh_transforms_prev[i] = scene_order_transforms_prev[src];
	}
	tlas_host_view_stale = false;
}

// What rt_build_tlas takes: one entry per instance in scene order
void Integrator::fill_scene_order_tables() {
	size_t mesh_count = scene.meshes.size();
	scene_order_roots.resize(mesh_count); scene_order_materials.resize(mesh_count); scene_order_boxes.resize(6 * mesh_count);
	scene_order_transforms.resize(mesh_count); scene_order_transforms_inv.resize(mesh_count); scene_order_transforms_prev.resize(mesh_count);
	for (size_t i = 0; i < mesh_count; i++) {
		const Mesh & mesh = scene.meshes[i];
		scene_order_roots[i]     = mesh_data_bvh_offsets[mesh.mesh_data_handle.handle] | (int(mesh.has_identity_transform()) << 31);
		scene_order_materials[i] = mesh.material_handle.handle;
		memcpy(scene_order_transforms     [i].cells, mesh.transform     .cells, sizeof(Matrix3x4));
		memcpy(scene_order_transforms_inv [i].cells, mesh.transform_inv .cells, sizeof(Matrix3x4));
		memcpy(scene_order_transforms_prev[i].cells, mesh.transform_prev.cells, sizeof(Matrix3x4));
		memcpy(&scene_order_boxes[6 * i],     &mesh.aabb_untransformed.min.x, 12);
		memcpy(&scene_order_boxes[6 * i + 3], &mesh.aabb_untransformed.max.x, 12);
	}
}

void Integrator::build_tlas() {
	size_t mesh_count = scene.meshes.size();
	if (wants_device_tlas()) {
		// Everything in scene order; the device sorts, builds and re-orders (replaces the SAH build, the CWBVH conversion and
		// the table shuffle below: Integrator.cpp:399-430 of the reference)
		if (pending_flatten) drop_flatten_worker();   // a tree built beside the frame loop has no place under a device-built TLAS: let go of it (update() would ask for a rebuild every frame while it is "ready")
		if (static_geometry.active) {
			// the device TLAS was switched on (device_tlas, enable_scene_update) after the scene had been flattened: a device-built TLAS
			// has a leaf per scene instance and knows nothing of the flattened tree, its aliases or a ray entry inside node 0 --
			// stage the geometry again the reference's way (init_geometry does not flatten while wants_device_tlas() holds)
			init_geometry();
			if (cpu_config.device_blas > 0) geometry_was_rebuilt();
			check(rt_set_static_geometry(ctx, 0));
		}
		fill_scene_order_tables();
		check(rt_build_tlas(ctx, scene_order_roots.data(), scene_order_materials.data(), scene_order_transforms[0].cells, scene_order_transforms_inv[0].cells,
		                    scene_order_transforms_prev[0].cells, scene_order_boxes.data(), mesh_count));
		tlas_on_device = true; tlas_host_view_stale = true;
		return;
	}
	tlas_on_device = false;
	StaticGeometry & flat = static_geometry;
	if (flat.active) { // the flattened instances have to stand where they stood when they were flattened
		bool moved = cpu_config.bvh_type != BVHType::BVH8;
		for (size_t j = 0; j < flat.members.size(); j++) {
			const Mesh & mesh = scene.meshes[size_t(flat.members[j])];
			const StaticGeometry::Pose & pose = flat.member_poses[j];
			if (memcmp(&mesh.position, &pose.position, sizeof(Vector3)) || memcmp(&mesh.rotation, &pose.rotation, sizeof(Quaternion)) || mesh.scale != pose.scale) {
				instance_has_moved[size_t(flat.members[j])] = 1;   // for good: it gets a TLAS leaf of its own from now on
				moved = true;
			}
		}
		if (moved) { // flatten what still stands still
			if (flatten_asynchronously && cpu_config.bvh_type == BVHType::BVH8 && cpu_config.device_blas <= 0) {
				// ... beside the frame loop: this frame and the next ones are rendered in the reference's layout (every instance a TLAS
				// leaf; the per-mesh trees and their triangles have been on the device all along), a worker builds the new tree
				start_flatten_worker();
				flat.active = false; flat.members.clear(); flat.movers.clear();
				alias_mesh_ids.clear(); alias_triangle_ids.clear();          // (the copies stay where they are, unreachable: no TLAS leaf leads to their tree)
				if (ctx) check(rt_upload_triangle_aliases(ctx, nullptr, nullptr));
				tlas_raw.indices.resize(mesh_count); tlas_raw.nodes.resize(mesh_count * 2);
				tlas_builder = std::make_unique<SAHBuilder>(tlas_raw, mesh_count);
			} else {   // ... now: a one-off stall of a build + upload
				init_geometry();
				if (cpu_config.device_blas > 0) geometry_was_rebuilt();   // (host-built trees: the originals keep their places, the copies are the tail)
			}
		}
	} else if (pending_flatten) {
		if (!pending_flatten_is_current()) start_flatten_worker();   // one of its members moved meanwhile: a new build for those that are left (the old one is not waited for)
		else if (pending_flatten->ready.load()) {
			// the worker is done: stage the flattened layout again (init_geometry recognises its own member set and input and takes the tree)
			init_geometry();
			drop_flatten_worker();
		}
	}
	bool whole_scene = flat.active && flat.movers.empty();   // everything is in the flattened tree: rays start inside it, there is no TLAS
	if (flat.active) { // one TLAS leaf for the flattened tree (leaf 0 of the build), one per instance that has moved
		std::vector<AABB> leaf_boxes(size_t(flat.leaves()));
		flat.aabb = AABB::create_empty();   // (Mesh::update fills in the world boxes: they are not known when the geometry is set up)
		for (int member : flat.members) flat.aabb.expand(scene.meshes[member].aabb);
		leaf_boxes[0] = flat.aabb;
		for (size_t k = 0; k < flat.movers.size(); k++) leaf_boxes[1 + k] = scene.meshes[flat.movers[k]].aabb;
		if (!whole_scene) tlas_builder->build(leaf_boxes);
	} else {
		tlas_builder->build(scene.meshes);
	}

	bool use_bvh8 = cpu_config.bvh_type == BVHType::BVH8;
	if (cpu_config.bvh_type == BVHType::BVH4) {
		BVH4Converter(tlas_4, tlas_raw).convert();
		memcpy((void *)aggregated_bvh_nodes_4.data(), tlas_4.nodes.data(), tlas_4.nodes.size() * sizeof(BVHNode4));
		if (ctx) check(rt_upload_tlas_bvh4(ctx, tlas_4.nodes.data(), tlas_4.nodes.size()));
		tlas.indices = tlas_4.indices;
	} else if (use_bvh8) {
		if (whole_scene) { tlas.indices.assign(1, 0); tlas.nodes.assign(1, aggregated_bvh_nodes_8[size_t(flat.root)]); }   // node 0 = the tree's root (its indices are absolute): rt_set_static_geometry
		else tlas_converter->convert();
		memcpy(aggregated_bvh_nodes_8.data(), tlas.nodes.data(), tlas.nodes.size() * sizeof(BVHNode8));
		if (ctx) check(rt_upload_tlas(ctx, tlas.nodes.data(), tlas.nodes.size()));
	} else {
		memcpy(aggregated_bvh_nodes_2.data(), tlas_raw.nodes.data(), tlas_raw.nodes.size() * sizeof(BVHNode2));
		if (ctx) check(rt_upload_tlas_bvh2(ctx, tlas_raw.nodes.data(), tlas_raw.nodes.size()));
		tlas.indices = tlas_raw.indices;
	}
	reference_tlas_order.clear();
	if (flat.active) { // rows: the TLAS leaves (leaf 0 of the build was the flattened tree: no scene mesh of its own), then the members
		for (int & leaf : tlas.indices) leaf = leaf == 0 ? -1 : flat.movers[size_t(leaf) - 1];
		tlas.indices.insert(tlas.indices.end(), flat.members.begin(), flat.members.end());
		if (use_bvh8) { // the top-level tree of the reference's layout, for its leaf order alone (a few hundred boxes: microseconds)
			BVH2 raw; raw.indices.resize(mesh_count); raw.nodes.resize(mesh_count * 2);
			SAHBuilder(raw, mesh_count).build(scene.meshes);
			BVH8 wide;
			BVH8Converter(wide, raw).convert();
			reference_tlas_order = wide.indices;
		}
	}

	size_t rows = tlas.indices.size();
	mesh_bvh_root_indices.resize(rows); mesh_material_ids.resize(rows);
	mesh_transforms.resize(rows); mesh_transforms_inv.resize(rows); mesh_transforms_prev.resize(rows);
	for (size_t i = 0; i < rows; i++) {
		if (tlas.indices[i] < 0) { // the flattened static geometry: world space, no material of its own (hits name the members' rows)
			mesh_bvh_root_indices[i] = flat.root | int(0x80000000u);
			mesh_material_ids[i] = 0;
			Matrix4 identity;
			memcpy(mesh_transforms[i].cells, identity.cells, sizeof(Matrix3x4)); memcpy(mesh_transforms_inv[i].cells, identity.cells, sizeof(Matrix3x4)); memcpy(mesh_transforms_prev[i].cells, identity.cells, sizeof(Matrix3x4));
			continue;
		}
		const Mesh & mesh = scene.meshes[size_t(tlas.indices[i])];
		mesh_bvh_root_indices[i] = mesh_data_bvh_offsets[mesh.mesh_data_handle.handle] | (int(mesh.has_identity_transform()) << 31);
		mesh_material_ids[i] = mesh.material_handle.handle;
		memcpy(mesh_transforms     [i].cells, mesh.transform     .cells, sizeof(Matrix3x4));
		memcpy(mesh_transforms_inv [i].cells, mesh.transform_inv .cells, sizeof(Matrix3x4));
		memcpy(mesh_transforms_prev[i].cells, mesh.transform_prev.cells, sizeof(Matrix3x4));
	}
	if (ctx) check(rt_upload_instances(ctx, mesh_bvh_root_indices.data(), mesh_material_ids.data(),
		mesh_transforms[0].cells, mesh_transforms_inv[0].cells, mesh_transforms_prev[0].cells, rows));
	if (ctx && cpu_config.bvh_type == BVHType::BVH8) {
		check(rt_set_static_geometry(ctx, whole_scene ? 1 : 0));
		check(rt_set_skip_behind_hit(ctx, cpu_config.skip_behind_hit ? 1 : 0));
		// rays start inside the one tree: its top levels (breadth-first: the first nodes from its root) may live in LDS
	}
}

rt_gpu_config Integrator::make_device_config() const {
	rt_gpu_config c = { };
	c.reconstruction_filter               = int(gpu_config.reconstruction_filter);
	c.aov_mask                            = gpu_config.aov_mask;
	c.num_bounces                         = gpu_config.num_bounces;
	c.enable_mipmapping                   = gpu_config.enable_mipmapping;
	c.enable_next_event_estimation        = gpu_config.enable_next_event_estimation;
	c.enable_multiple_importance_sampling = gpu_config.enable_multiple_importance_sampling;
	c.enable_russian_roulette             = gpu_config.enable_russian_roulette;
	c.enable_svgf                         = gpu_config.enable_svgf;
	c.enable_spatial_variance             = gpu_config.enable_spatial_variance;
	c.enable_taa                          = gpu_config.enable_taa;
	c.alpha_colour                        = gpu_config.alpha_colour;
	c.alpha_moment                        = gpu_config.alpha_moment;
	c.num_atrous_iterations               = gpu_config.num_atrous_iterations;
	c.sigma_z                             = gpu_config.sigma_z;
	c.sigma_n                             = gpu_config.sigma_n;
	c.sigma_l                             = gpu_config.sigma_l;
	return c;
}

// reference: Integrator::update (Integrator.cpp:432-528)
void Integrator::update(float delta) {
	if (invalidated_gpu_config && gpu_config.enable_svgf && scene.camera.aperture_radius > 0.0f) {
		fprintf(stderr, "WARNING: SVGF and DoF cannot simultaneously be enabled!\n");
		scene.camera.aperture_radius = 0.0f;
		invalidated_camera = true;
	}

	if (ctx && cpu_config.enable_scene_update != scheduler_for_scene_updates) {
		// A scene that uploads a new TLAS every frame keeps several frames in flight only under the slot scheduler
		// (each chain reads the scene version it was submitted with); everything else feeds the merged wavefront.
		scheduler_for_scene_updates = cpu_config.enable_scene_update;
		check(rt_set_scheduler(ctx, scheduler_for_scene_updates ? RT_SCHEDULER_SLOTS : RT_SCHEDULER_MERGED));
	}
	if (cpu_config.enable_scene_update) {
		if (!scene_advanced_by_another_integrator) scene.update(delta);
		invalidated_scene = true;
	} else if ((gpu_config.enable_svgf || invalidated_scene) && !scene_advanced_by_another_integrator) {
		scene.camera.update(0.0f);
		scene.update(0.0f);
	}

	if (pixel_query_status == PixelQueryStatus::OUTPUT_READY) { // reference: Integrator.cpp:483-495
		if (ctx) check(rt_get_pixel_query(ctx, &pixel_query.mesh_id, &pixel_query.triangle_id));
		sync_host_view_of_device_tlas();
		if (pixel_query.mesh_id != INVALID) pixel_query.mesh_id = tlas.indices[pixel_query.mesh_id]; // TLAS order -> scene mesh index
		pixel_query.pixel_index = INVALID;
		if (ctx) check(rt_set_pixel_query(ctx, INVALID));
		pixel_query_status = PixelQueryStatus::INACTIVE;
	}

	if (pending_flatten && pending_flatten->ready.load()) invalidated_scene = true;   // the tree built beside the frame loop is there: build_tlas installs it
	if (pending_reseat && pending_reseat->ready.load() && install_reseat()) invalidated_scene = true;   // the tree seated beside the frame loop is there: its root's copy in node 0 follows (build_tlas)
	if (invalidated_scene) {
		invalidated_scene = false;
		build_tlas();
	}

	scene.camera.update(delta);

	// Seat the flattened tree (again) beside the frame loop: a tree the device built has never been seated; a tree seated for a viewpoint is seated again once
	// the camera has travelled static_reseat_distance x the geometry's diagonal from there (Integrator.h: PendingReseat)
	if (static_geometry.active && cpu_config.bvh_type == BVHType::BVH8 && cpu_config.static_slot_learning_rays > 0 && !pending_reseat && !pending_flatten) {
		bool wanted = flattened_tree_needs_seating;
		if (!wanted && cpu_config.static_reseat_distance > 0.0f && cpu_config.static_slot_learning_viewpoint && seated_for_a_viewpoint) {
			const float diagonal = static_geometry.diagonal;
			wanted = std::isfinite(diagonal) && diagonal > 0.0f && Vector3::length(scene.camera.position - seated_for_position) > cpu_config.static_reseat_distance * diagonal;
		}
		if (wanted) {
			// (the first tree of an integrator is seated inside this update -- nothing has been rendered yet, the scene's load took longer than the 0.3 s --;
			// later ones, a device rebuild after a member moved or a camera that has travelled, beside the frame loop)
			const bool inside_this_update = !reseat_asynchronously || (flattened_tree_needs_seating && geometry_generation <= 1);
			flattened_tree_needs_seating = false;
			start_reseat_worker(!inside_this_update);
			if (inside_this_update && pending_reseat) {
				if (pending_reseat->worker.joinable()) pending_reseat->worker.join();
				if (install_reseat()) build_tlas();
			}
		}
	}

	if (scene.camera.moved || invalidated_camera) {
		const Camera & c = scene.camera;
		memcpy(device_camera.position,           &c.position.x,                   12);
		memcpy(device_camera.bottom_left_corner, &c.bottom_left_corner_rotated.x, 12);
		memcpy(device_camera.x_axis,             &c.x_axis_rotated.x,             12);
		memcpy(device_camera.y_axis,             &c.y_axis_rotated.x,             12);
		device_camera.pixel_spread_angle = c.pixel_spread_angle;
		device_camera.aperture_radius    = c.aperture_radius;
		device_camera.focal_distance     = c.focal_distance;
		if (ctx) check(rt_set_camera(ctx, &device_camera));

		if (!gpu_config.enable_svgf) sample_index = 0;
		invalidated_camera = false;
	}

	if (invalidated_aovs) {
		invalidated_aovs = false;
		invalidated_gpu_config = true; // the device (re)allocates AOV buffers from aov_mask
	}

	if (invalidated_gpu_config) {
		invalidated_gpu_config = false;
		sample_index = 0;
		rt_gpu_config c = make_device_config();
		if (ctx) check(rt_set_config(ctx, &c));
		if (ctx) check(rt_set_svgf_tiles(ctx, cpu_config.svgf_lds_tiles ? 1 : 0));
	} else if (scene.camera.moved && !gpu_config.enable_svgf) {
		sample_index = 0;
	} else {
		sample_index++;
	}
	scene.camera.moved = false;
}

void Integrator::set_pixel_query(int x, int y) {
	if (x < 0 || y < 0 || x >= screen_width || y >= screen_height) return;
	y = screen_height - y; // window coordinates are top-down
	pixel_query.pixel_index = x + y * screen_pitch;
	pixel_query.mesh_id     = INVALID;
	pixel_query.triangle_id = INVALID;
	if (ctx) check(rt_set_pixel_query(ctx, pixel_query.pixel_index));
	pixel_query_status = PixelQueryStatus::PENDING;
}

std::vector<float> Integrator::read_aov(AOVType type, bool accumulated) {
	require_device();
	std::vector<float> image(size_t(screen_pitch) * screen_height * 4);
	check(rt_read_aov(ctx, int(type), image.data(), accumulated ? 1 : 0));
	return image;
}

static std::vector<Vector3> rgb_of(const std::vector<float> & rgba) {
	std::vector<Vector3> rgb(rgba.size() / 4);
	for (size_t i = 0; i < rgb.size(); i++) rgb[i] = Vector3(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2]);
	return rgb;
}

void Integrator::save_image(const std::string & filename) {
	std::string error;
	if (!Exporters::save(filename, screen_pitch, screen_width, screen_height, rgb_of(read_framebuffer()), &error)) throw std::runtime_error(error);

	size_t slash = filename.find_last_of('/');
	std::string directory = slash == std::string::npos ? std::string() : filename.substr(0, slash + 1);
	const struct { AOVType type; const char * name; } extras[] = {
		{ AOVType::ALBEDO, "albedo.exr" }, { AOVType::NORMAL, "normal.exr" }, { AOVType::POSITION, "position.exr" } };
	for (const auto & extra : extras) {
		if (!aov_is_enabled(extra.type)) continue;
		if (!EXRExporter::save(directory + extra.name, screen_pitch, screen_width, screen_height, rgb_of(read_aov(extra.type, true)))) {
			throw std::runtime_error("failed to write '" + directory + extra.name + "'");
		}
	}
}

std::vector<float> Integrator::read_framebuffer() {
	require_device();
	std::vector<float> image(size_t(screen_pitch) * screen_height * 4);
	check(rt_read_framebuffer(ctx, image.data()));
	return image;
}
