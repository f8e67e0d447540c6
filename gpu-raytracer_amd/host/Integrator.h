// Host side of the hot path: turns a Scene into the flat device data formats and
// drives the device layer through the C ABI (include/gpu_raytracer_amd.h).
// Same role, member names and invalidation protocol as the reference's
// `struct Integrator` (Src/Renderer/Integrators/Integrator.h:56-296, Integrator.cpp):
//   cuda_init/cuda_free  -> gpu_init/gpu_free   (cuda_* kept as aliases)
//   resize_init/resize_free, update(delta), render(), invalidated_* flags,
//   sample_index, aov_enable/aov_disable/aov_is_enabled, set_pixel_query.
// GL interop, ImGui and NVRTC hot-reload have no counterpart (headless).
#pragma once
#include <atomic>
#include <memory>
#include <thread>
#include <vector>

#include "Scene.h"
#include "PMJ.h"

struct DeviceTriangle { // CUDA/Raytracing/Triangle.h:4-11, Integrator.h:139-151
	Vector3 position_0, position_edge_1, position_edge_2;
	Vector3 normal_0,   normal_edge_1,   normal_edge_2;
	Vector2 tex_coord_0, tex_coord_edge_1, tex_coord_edge_2;
};
static_assert(sizeof(DeviceTriangle) == 96, "device triangle is 6 x float4");

struct Matrix3x4 { float cells[12]; };

union alignas(16) DeviceMaterial { // CUDA/Material.h:21-39
	struct { Vector3 emission; } light;
	struct { Vector3 diffuse; int texture_id; } diffuse;
	struct { Vector3 diffuse; int texture_id; float linear_roughness; } plastic;
	struct { int medium_id; float ior; float linear_roughness; } dielectric;
	struct { Vector3 eta; float linear_roughness; Vector3 k; } conductor;
	float raw[8];
	DeviceMaterial() { for (float & f : raw) f = 0.0f; }
};
static_assert(sizeof(DeviceMaterial) == 32, "device material is 2 x float4");

struct alignas(16) DeviceMedium { Vector3 sigma_a; float g; Vector3 sigma_s; float pad = 0.0f; };
static_assert(sizeof(DeviceMedium) == 32, "device medium is 2 x float4");

struct PixelQuery { int pixel_index, mesh_id, triangle_id; };

struct Integrator {
	Scene & scene;
	rt_context * ctx = nullptr; // null = host-only baking (tests, CPU oracle); render() then throws

	bool invalidated_scene      = true;
	bool invalidated_sky        = true;
	bool invalidated_materials  = true;
	bool invalidated_mediums    = true;
	bool invalidated_camera     = true;
	bool invalidated_gpu_config = true;
	bool invalidated_aovs       = true;

	int screen_width = 0, screen_height = 0, screen_pitch = 0;
	int pixel_count = 0;
	int sample_index = 0;

	// Multi-GPU: scan-order pixel range rendered by this integrator (default: all)
	int pixel_range_offset = 0, pixel_range_count = -1;

	PixelQuery pixel_query = { INVALID, INVALID, INVALID };
	enum struct PixelQueryStatus { INACTIVE, PENDING, OUTPUT_READY } pixel_query_status = PixelQueryStatus::INACTIVE; // Integrator.h:75-79
	bool scheduler_for_scene_updates = false;
	// Several integrators on ONE scene (FrameSplit: one per GPU): Camera::update and Mesh::update shift "current" into
	// "previous" on every call, so only the first integrator of a frame may advance the scene; the others upload what it left.
	bool scene_advanced_by_another_integrator = false;
	// The current TLAS was built by the device (rt_build_tlas): the host holds no TLAS nodes, `tlas.indices` and the
	// TLAS-ordered instance tables are fetched from the device when something on the host needs them (pixel queries,
	// the parity checker's view of the scene).
	bool tlas_on_device = false, tlas_host_view_stale = false;
	float device_blas_build_ms = 0.0f;   // cpu_config.device_blas: what rt_build_geometry took on the device (sort + levels + gather)
	bool wants_device_tlas() const;
	void sync_host_view_of_device_tlas();
	void fill_scene_order_tables();
	std::vector<int> scene_order_roots, scene_order_materials; std::vector<float> scene_order_boxes;
	std::vector<Matrix3x4> scene_order_transforms, scene_order_transforms_inv, scene_order_transforms_prev; // the device context schedules for per-frame scene uploads (rt_set_scheduler)

	// ---- host staging of everything the device consumes (filled by init_* / build_tlas) ----
	std::vector<DeviceTriangle> aggregated_triangles;
	std::vector<BVHNode8>       aggregated_bvh_nodes_8;   // slots [0, 2*meshes) = TLAS
	std::vector<BVHNode2>       aggregated_bvh_nodes_2;   // same layout, binary BVH
	std::vector<BVHNode4>       aggregated_bvh_nodes_4;   // same layout, 4-wide BVH
	std::vector<int>            reverse_indices;          // original triangle -> position in aggregated_triangles
	std::vector<int>            mesh_data_bvh_offsets;
	std::vector<int>            mesh_data_triangle_offsets;
	std::vector<int>            mesh_data_index_offsets;

	// ---- flattened static geometry (cpu_config.merge_static; no counterpart in the reference) ----
	// Every instance that has not been seen moving is copied, triangle by triangle (in world space), into ONE extra
	// bottom-level tree. The TLAS has one leaf for that tree and one per remaining instance ("movers"); when there are no
	// movers there is no TLAS at all and rays start inside the tree (rt_set_static_geometry). The instance tables have a row
	// per TLAS leaf (in TLAS order, as ever) followed by a row per flattened instance in fixed order: a hit on a copy is
	// reported by the device as (row of its instance, its original triangle), so materials, transforms, light tables,
	// pixel queries and the SVGF ids keep naming the scene's own instances.
	struct StaticGeometry {
		bool built  = false;          // the merged tree and the triangle copies are part of the uploaded geometry
		bool active = false;          // ... and the TLAS / instance tables use them
		struct Pose { Vector3 position; Quaternion rotation; float scale; };
		std::vector<int> members;     // scene mesh indices, in the order of their table rows
		std::vector<Pose> member_poses;   // where they stood when their triangles were copied
		std::vector<int> movers;      // every other instance
		int    root = 0;              // root node of the merged tree
		size_t copy_bytes = 0;        // what the copies and the tree's nodes add to the device's geometry
		int    top_nodes = 0;         // its nodes are in breadth-first order; so many of them (from the root) make up its top levels
		AABB   aabb;
		float  diagonal = 0.0f;       // of the box around the copies: what cpu_config.static_reseat_distance is a fraction of
		double build_seconds = 0.0;   // host SAH + CWBVH conversion (0 when the device built it)
		int leaves() const { return 1 + int(movers.size()); }   // rows in front of the members' rows
	} static_geometry;
	// A flattened instance that starts to move leaves the tree -- WITHOUT stalling the frame loop for the rebuild (0.5 s for Sponza):
	// the frame in which it is noticed, and every frame until the new tree is there, is rendered in the reference's layout (a TLAS
	// over all instances; their trees and triangles are still on the device, untouched by the flattening), while a worker thread
	// builds the tree of the members that are left. update() notices the finished build, init_geometry() takes the tree instead of
	// building one (what remains on the frame loop's thread is staging and the upload).
	struct PendingFlatten {
		std::thread worker;
		std::atomic<bool> ready { false };
		std::vector<int> members;               // the member set the tree was built for
		std::vector<StaticGeometry::Pose> poses; // ... and where they stood
		std::vector<Triangle> world;            // their triangles in world space, in member order (the build's input)
		std::vector<int> source_member, source_triangle;   // per world triangle: its member, its index among all original triangles
		std::vector<DeviceTriangle> copy_triangles; std::vector<int> copy_member, copy_original;   // the tree's leaf triangles as they go behind the staged originals
		BVH8 wide; int top_nodes = 0; double build_seconds = 0.0;
		bool failed = false;
	};
	std::unique_ptr<PendingFlatten> pending_flatten;
	std::vector<std::unique_ptr<PendingFlatten>> retired_flattens;   // builds whose input went out of date while they ran: joined when they are done, never waited for
	bool pending_flatten_is_current();
	size_t staged_index_total = 0, staged_node_total = 0;   // triangles / nodes of the reference part of the staged arrays (in front of the copies / the flattened tree)
	bool flatten_asynchronously = true;     // (false: rebuild inside build_tlas, as round 3 did -- tests compare the two)
	int  reflattens_completed = 0;
	void start_flatten_worker();
	void drop_flatten_worker();
	std::vector<int> flatten_candidates() const;
	std::vector<Triangle> world_triangles_of(const std::vector<int> & members, std::vector<int> * source_member, std::vector<int> * source_triangle) const;
	// Seating again (SlotOrder.cpp) beside the frame loop: a copy of the flattened tree and its triangles goes to a worker, which deals the children of every node
	// to octant slots for the camera as it stands now; the finished nodes replace the tree's nodes in place (same count, same boxes, same leaves) between two
	// frames. Started when the camera has travelled (cpu_config.static_reseat_distance) and for a tree the device built (its collapse knows no seating).
	struct PendingReseat {
		std::thread worker;
		std::atomic<bool> ready { false };
		bool failed = false;
		BVH8 tree;                               // the flattened tree, root = node 0, indices relative to it
		std::vector<Triangle> triangles;         // its leaf triangles in leaf order (world space)
		std::vector<unsigned> absolute;          // node of `tree` -> its place in aggregated_bvh_nodes_8
		size_t triangle_base = 0;                // leaf position 0 of `tree` in aggregated_triangles
		unsigned long long generation = 0;       // of the staged geometry it was copied from
		Vector3 camera_position;                 // what it is seated for
		double seconds = 0.0;
	};
	std::unique_ptr<PendingReseat> pending_reseat;
	std::vector<std::unique_ptr<PendingReseat>> retired_reseats;
	unsigned long long geometry_generation = 0;   // every init_geometry is another one
	Vector3 seated_for_position; bool seated_for_a_viewpoint = false;
	bool flattened_tree_needs_seating = false;    // a tree the device built: never seated yet
	int reseats_completed = 0; double last_reseat_seconds = 0.0;
	bool reseat_asynchronously = true;            // (false: seat inside update(), for tests)
	void start_reseat_worker(bool beside_frame_loop);
	bool install_reseat();                        // true when finished nodes went in
	void drop_reseat_worker();
	SlotLearningView slot_learning_view() const;   // the camera as it stands: what bvh8_learn_slot_order samples its paths from
	std::vector<char> instance_has_moved;   // per scene mesh: seen with a changed transform since the scene was loaded -> never flattened again
	std::vector<int> alias_mesh_ids, alias_triangle_ids;   // per device triangle (-1: not a copy): what rt_upload_triangle_aliases was given

	std::vector<int>       mesh_bvh_root_indices;   // TLAS order; MSB = identity transform
	std::vector<int>       mesh_material_ids;
	std::vector<Matrix3x4> mesh_transforms, mesh_transforms_inv, mesh_transforms_prev;

	std::vector<unsigned char>  material_types;
	std::vector<DeviceMaterial> materials;
	std::vector<DeviceMedium>   media;

	BVH2 tlas_raw;
	// Flattened static geometry: the order the REFERENCE'S top-level tree would list the scene's instances in (its SAH build over
	// the instance boxes + 8-wide collapse, Integrator.cpp:399-430), kept so that tables whose ORDER enters the rendered image --
	// the light meshes' cumulative distribution: which light a random number picks -- are the reference's in every layout
	std::vector<int> reference_tlas_order;
	BVH8 tlas;                                   // tlas.indices[i] = scene mesh index of instance-table row i (TLAS leaf i; -1: the flattened static geometry)
	BVH4 tlas_4;                                 // bvh_type = BVH4
	std::unique_ptr<SAHBuilder>    tlas_builder;
	std::unique_ptr<BVH8Converter> tlas_converter;

	std::vector<float>         pmj_samples;
	std::vector<unsigned char> blue_noise;

	rt_camera device_camera = { };

	explicit Integrator(Scene & scene, int device_ordinal = 0);
	virtual ~Integrator();

	virtual void gpu_init(int screen_width, int screen_height);
	virtual void gpu_free();
	void cuda_init(unsigned /*frame_buffer_handle*/, int w, int h) { gpu_init(w, h); }
	void cuda_free() { gpu_free(); }

	void init_materials();
	void init_geometry();
	void init_sky();
	void init_rng();

	virtual void resize_free() = 0;
	virtual void resize_init(int width, int height) = 0;

	void aov_enable (AOVType t) { gpu_config.aov_mask |=  (1u << int(t)); invalidated_aovs = true; }
	void aov_disable(AOVType t) { gpu_config.aov_mask &= ~(1u << int(t)); invalidated_aovs = true; }
	bool aov_is_enabled(AOVType t) const { return gpu_config.aov_mask & (1u << int(t)); }

	void build_tlas();
	// build_tlas has run init_geometry again (a flattened instance moved). With trees built on the device the triangles of the
	// other meshes may have changed places too (leaf positions are dealt over all trees together): whatever names device
	// triangles by index has to follow -- the path tracer's light tables.
	virtual void geometry_was_rebuilt() { }

	virtual void update(float delta);
	virtual void render() = 0;

	void set_pixel_query(int x, int y);
	void set_pixel_range(int offset, int count) { pixel_range_offset = offset; pixel_range_count = count; if (ctx) check(rt_set_pixel_range(ctx, offset, count)); }

	// Reads an AOV accumulator (pitch*height float4) back to the host.
	std::vector<float> read_aov(AOVType type, bool accumulated = true);
	std::vector<float> read_framebuffer();

	// Screenshot as the reference takes it (Main.cpp:199-246): the frame to `filename` (.ppm tone-mapped,
	// .exr raw radiance) and, for each enabled auxiliary AOV, albedo.exr / normal.exr / position.exr
	// next to it. Throws on an unsupported extension or a write failure.
	void save_image(const std::string & filename);

	rt_gpu_config make_device_config() const;

	// queue sizes per bounce and stage times of the last render() (rt_get_counters)
	rt_counters counters() { require_device(); rt_counters c; check(rt_get_counters(ctx, &c)); return c; }

protected:
	void check(int status) const; // throws std::runtime_error with rt_last_error on failure
	void require_device() const;
};
