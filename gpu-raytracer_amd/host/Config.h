// Global render settings, same field names as the reference so callers that
// poke `gpu_config.num_bounces` / `cpu_config.bvh_type` keep working
// (reference: Src/Config.h:7-64, Src/CUDA/Common.h:18-67).
#pragma once
#include <string>
#include <vector>

#include "../../include/gpu_raytracer_amd.h"

enum struct ReconstructionFilter : int { BOX = RT_FILTER_BOX, TENT = RT_FILTER_TENT, GAUSSIAN = RT_FILTER_GAUSSIAN };

enum struct AOVType : int {
	RADIANCE = RT_AOV_RADIANCE,
	RADIANCE_DIRECT = RT_AOV_RADIANCE_DIRECT,
	RADIANCE_INDIRECT = RT_AOV_RADIANCE_INDIRECT,
	ALBEDO = RT_AOV_ALBEDO,
	NORMAL = RT_AOV_NORMAL,
	POSITION = RT_AOV_POSITION,
	COUNT = RT_AOV_COUNT
};

// Mirrors the reference GPUConfig field for field; converted to the C-ABI's
// rt_gpu_config by Integrator::update.
struct GPUConfig {
	ReconstructionFilter reconstruction_filter = ReconstructionFilter::GAUSSIAN;
	unsigned aov_mask = 0;

	int num_bounces = 10;

	bool enable_mipmapping                   = true;
	bool enable_next_event_estimation        = true;
	bool enable_multiple_importance_sampling = true;
	bool enable_russian_roulette             = true;
	bool enable_svgf                         = false;
	bool enable_spatial_variance             = true;
	bool enable_taa                          = true;

	float alpha_colour = 0.1f;
	float alpha_moment = 0.1f;
	int   num_atrous_iterations = 6;
	float sigma_z =  4.0f;
	float sigma_n = 16.0f;
	float sigma_l = 10.0f;
};

enum struct BVHType { BVH, SBVH, BVH4, BVH8 };
enum struct IntegratorType { PATHTRACER, AO };
enum struct MipmapFilterType { BOX, LANCZOS, KAISER };

struct CPUConfig {
	int initial_width  = 900;
	int initial_height = 600;

	std::vector<std::string> scene_filenames;
	std::string              sky_filename;

	int         output_sample_index = INVALID_SAMPLE;
	std::string output_filename     = "render.ppm";

	bool enable_scene_update = false;
	// Where the TLAS is built: 0 = on the host (SAH + CWBVH conversion, byte-identical to the reference's), 1 = on the device
	// (rt_build_tlas: one kernel launch, CWBVH only, up to 4096 instances), -1 = on the device when the scene is rebuilt
	// every frame (enable_scene_update) and has at least 1024 instances: there the host build is the CPU work inside the frame loop
	int  device_tlas = -1;
	// Where the bottom-level trees of a CWBVH scene are built: 0 = on the host (SAH / SBVH builder + BVH8Converter, byte-identical
	// to the reference's), 1 = on the device (rt_build_geometry: a linear BVH over all meshes at once; fast to build, dearer to traverse)
	int  device_blas = 0;
	// ... and, for the flattened tree of such a build, early split clipping in front of it: a triangle longer than this fraction of the flattened
	// geometry's longest side is cut into pieces, each piece a reference with its own box (StaticBVHBuilder::presplit, rt_set_build_boxes) -- the
	// Morton-order builder has no spatial splits of its own. 0: off.
	float device_presplit = 0.08f;   // (Sponza, flattened, ms per step of the benchmark: 0 1.72, 0.3 1.67, 0.15 1.61, 0.1 1.56, 0.075 1.55, 0.05 1.62, 0.025 1.62; host tree 1.43: profiles/r05_device_blas_presplit.txt)
	// Static geometry: 1 = every instance that has not been seen moving and whose mesh is not instanced more than twice (two or
	// more such instances) is flattened into ONE bottom-level tree -- a ray then walks one well-built tree (StaticBVHBuilder:
	// SAH object and spatial splits) instead of entering a dozen overlapping per-mesh trees; hits still name the scene's
	// instances and triangles (rt_upload_triangle_aliases), and with nothing left outside the tree rays start inside it
	// (rt_set_static_geometry). CWBVH with the TLAS built on the host only; an instance that starts to move is taken out
	// again (one rebuild). 2 = the same with the per-mesh SAH builder (no spatial splits). 3 = only instances with the
	// identity transform (their copies are their triangles bit for bit; a transformed instance's copies are its triangles
	// taken to world space, which the reference's layout never does: it takes the ray to object space). 0 = one BLAS per
	// mesh under the TLAS, exactly the reference's structure (Integrator.cpp:101-283).
	int  merge_static = 1;
	// The flattening policy, as memory (a copy costs ~176 bytes per triangle: 96 B shading triangle, 48 B traversal positions, the
	// names of the original, its share of the tree's nodes): a mesh that is instanced N times joins only if the N - 1 copies BEYOND
	// the first stay under static_mesh_copy_limit_mb -- copying an instanced mesh per instance is the opposite of what instancing
	// is for (441 instances of a 102 400-triangle mesh: 7.9 GB and a 45 M-triangle tree to build) --, and meshes join in scene order
	// until all copies together reach static_copy_budget_mb. What does not join keeps its TLAS leaf.
	int  static_mesh_copy_limit_mb = 64;
	int  static_copy_budget_mb = 2048;
	// ... and what a triangle test costs relative to a node step when that tree's binary form is collapsed into 8-wide nodes
	// (BVH8Converter: 1 in the reference; here a triangle test runs with a quarter of a wave's lanes, a node step with most)
	float static_primitive_cost = 1.0f;
	// ... and how that collapse deals a node's children to the eight octant slots (BVH8Converter::slot_assignment): 0 = the reference's greedy rule by centres
	// (BVH8Converter.cpp:146-205), 5 = inner children to the slots of least total cost by the corner a ray enters them at, leaves take what is left.
	// The flattened tree is this renderer's own layout: its slot order is free.
	int   static_slot_assignment = 5;
	// ... and then re-seated by what this many sample rays over the flattened geometry say (bvh8_learn_slot_order, SlotOrder.cpp: a seeded, pure function of the
	// geometry). 0: off.
	int   static_slot_learning_rays = 1000000;
	// The seating is trained for the camera as it stood (static_slot_learning_viewpoint): when the camera has travelled further than this fraction of the flattened
	// geometry's diagonal from there, the tree is seated again for the new viewpoint BESIDE the frame loop (a worker thread, 0.3 s for Sponza) and its nodes are
	// swapped in between two frames (rt_update_nodes). 0: never. Trees the DEVICE built (device_blas) get their first seating the same way.
	float static_reseat_distance = 0.1f;
	// Closest-hit rays of a one-tree scene drop stacked groups of children that lie behind the hit they already hold (rt_set_skip_behind_hit): same hits,
	// 12 % fewer node visits on Sponza. false: the reference's walk, node for node.
	bool  skip_behind_hit = true;
	int   static_slot_learning_viewpoint = 1;   // a quarter of those rays are paths from the camera as it stands when the tree is built (0: none are; half come from points of the free space, the rest from the surface, either way)
	// ... and early split clipping (StaticBVHBuilder::presplit, as in front of the device build) in front of that builder's own SAH + spatial splits: fraction of
	// the geometry's longest side above which a triangle is cut blindly first. 0: off.
	float static_presplit = 0.0f;

	IntegratorType integrator = IntegratorType::PATHTRACER; // read by the command-line front end

	MipmapFilterType mipmap_filter = MipmapFilterType::BOX;
	// BC1-quantise power-of-two textures like the reference does by default (BlockCompression.cpp). Off by
	// default here until the textured GPU parity tests have been re-run with it.
	bool enable_block_compression = true;  // the reference's default (Config.h:55): every power-of-two texture is stored as BC1
	// Block-compressed textures are decoded ONCE, when they are uploaded (rt_set_texture_expansion): the device keeps 64 bytes per
	// 4x4 block instead of 8 and the shade kernels fetch texels instead of decoding a block per fetch -- the same texel values.
	// false: the 8-byte blocks stay compressed on the device (an eighth of the memory, a third more shade time).
	bool expand_block_compressed_textures = true;
	// The SVGF filter's a-trous passes stage a workgroup's taps in LDS (rt_set_svgf_tiles); false: every tap is a global load, as in the
	// reference's kernel_svgf_atrous. Bit-identical images either way.
	bool svgf_lds_tiles = true;
	BVHType bvh_type = BVHType::BVH8;

	// "<mesh file>.bvh" caches (BVHCache.h). The reference always reads and writes them; a library
	// that may sit on a read-only asset tree only does so when asked to.
	bool enable_bvh_cache        = false;
	bool bvh_force_rebuild       = false; // ignore existing caches (they are still rewritten)
	bool enable_bvh_optimization = false; // BVHOptimizer::optimize on every binary tree after it is built (-O)
	int  bvh_optimizer_max_time        = 60000; // milliseconds
	int  bvh_optimizer_max_num_batches = 1000;

	float sah_cost_node = 4.0f;   // BVH8 conversion and leaf collapse
	float sah_cost_leaf = 1.0f;
	float sbvh_alpha    = 10e-5f; // 1: never split spatially, 0: always consider it

	static constexpr int INVALID_SAMPLE = -1;
};

extern GPUConfig gpu_config;
extern CPUConfig cpu_config;
