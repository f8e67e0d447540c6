// Geometry containers: MeshData = triangles + BLAS, Mesh = one placed instance
// (reference: Src/Renderer/MeshData.h, Mesh.h, Mesh.cpp).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "BVH.h"
#include "Config.h"
#include "Material.h"

struct MeshData {
	std::vector<Triangle> triangles;
	bool from_file = false; // loaded from a mesh file (as opposed to generated shapes)
	std::string filename;   // of that file
	std::string bvh_filename; // its BVH cache: filename + ".bvh", or "<archive>.shape_<i>.bvh" for serialized meshes

	BVH2 bvh2; // binary SAH BVH, one triangle per leaf
	BVH8 bvh8; // its CWBVH collapse
	BVH4 bvh4; // its 4-wide collapse

	// What the device traverses when cpu_config.bvh_type is not BVH8 (reference:
	// AssetManager.cpp:57-95, BVH.cpp:14-59): the SAH or spatial-split binary tree, leaf-collapsed
	// for file-loaded meshes, and for BVH4 the 4-wide collapse of that tree. Built on first use.
	BVH2 sbvh;        // the uncollapsed spatial-split tree (built, or read from the mesh's .bvh cache)
	BVH2 device_bvh2;
	BVH4 device_bvh4;
	int  device_bvh_type = -1; // BVHType the two members above were built for

	void prepare_device_bvh(BVHType type);
};

struct Scene;

struct Mesh {
	std::string name;

	AABB aabb_untransformed;
	AABB aabb;

	Handle<MeshData> mesh_data_handle;

	Vector3    position;
	Quaternion rotation;
	float      scale = 1.0f;

	Handle<Material> material_handle;

	Matrix4 transform;
	Matrix4 transform_inv;
	Matrix4 transform_prev;

	struct {
		float weight = 0.0f;
		int first_triangle_index = 0;
		int triangle_count = 0;
	} light;

	Mesh(std::string name, Handle<MeshData> mesh_data_handle, Handle<Material> material_handle)
		: name(std::move(name)), mesh_data_handle(mesh_data_handle), material_handle(material_handle) { }

	void calc_aabb(const Scene & scene);
	void update();
	bool has_identity_transform() const;

	Vector3 get_center() const { return aabb.get_center(); }
	AABB    get_aabb()   const { return aabb; }
};
