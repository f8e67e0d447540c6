// Scene = asset tables + placed meshes + camera + sky, with the reference's member
// names (Src/Renderer/Scene.h, Src/Assets/AssetManager.h) so that code written
// against `scene.asset_manager.materials` / `scene.meshes` / `scene.camera` ports over.
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "Camera.h"
#include "Config.h"
#include "Mesh.h"

struct Sky {
	std::vector<Vector4> data;
	int   width  = 0;
	int   height = 0;
	float scale  = 1.0f;

	// Radiance .hdr (RGBE) loader; an empty file name gives a 1x1 white sky, the
	// stand-in for the HDR files missing from the reference mount.
	void load(const std::string & filename);
};

struct AssetManager {
	std::vector<MeshData> mesh_datas;
	std::vector<Material> materials;
	std::vector<Medium>   media;
	std::vector<Texture>  textures;

	AssetManager();

	using FallbackLoader = std::function<std::vector<Triangle>(const std::string & filename)>;

	Handle<MeshData> add_mesh_data(const std::string & filename, FallbackLoader loader);
	// `filename` keys the mesh (and is what the loader receives); `bvh_filename` names its cache file
	Handle<MeshData> add_mesh_data(const std::string & filename, const std::string & bvh_filename, FallbackLoader loader);
	Handle<MeshData> add_mesh_data(std::vector<Triangle> triangles);
	Handle<Material> add_material(Material material);
	Handle<Medium>   add_medium(Medium medium);
	Handle<Texture>  add_texture(const std::string & filename, const std::string & name);

	// Builds every pending BLAS (one job per mesh file on a thread pool) and decodes textures.
	void wait_until_loaded();
	void prepare_device_bvhs(BVHType type); // builds MeshData::device_bvh* for a non-BVH8 type, in parallel

	MeshData & get_mesh_data(Handle<MeshData> h) { return mesh_datas[h.handle]; }
	Material & get_material (Handle<Material> h) { return materials [h.handle]; }
	Medium   & get_medium   (Handle<Medium>   h) { return media     [h.handle]; }
	Texture  & get_texture  (Handle<Texture>  h) { return textures  [h.handle]; }
	const MeshData & get_mesh_data(Handle<MeshData> h) const { return mesh_datas[h.handle]; }
	const Material & get_material (Handle<Material> h) const { return materials [h.handle]; }

	double bvh_build_ms = 0.0; // wall time of the parallel BLAS build

private:
	std::map<std::string, Handle<MeshData>> mesh_data_cache;
	std::map<std::string, Handle<Texture>>  texture_cache;

	struct PendingMesh    { int handle; std::string filename; FallbackLoader loader; std::string bvh_filename; };
	struct PendingTexture { int handle; std::string filename; };
	std::vector<PendingMesh>    pending_meshes;
	std::vector<PendingTexture> pending_textures;
	bool assets_loaded = false;
};

struct Scene {
	AssetManager asset_manager;

	Camera            camera;
	std::vector<Mesh> meshes;
	Sky               sky;

	bool has_diffuse    = false;
	bool has_plastic    = false;
	bool has_dielectric = false;
	bool has_conductor  = false;
	bool has_lights     = false;

	// Loads everything named in cpu_config.scene_filenames (.xml / .obj) and the sky.
	Scene();

	Mesh & add_mesh(std::string name, Handle<MeshData> mesh_data_handle, Handle<Material> material_handle = Handle<Material>::get_default());

	void check_materials();
	void update(float delta);
};

namespace OBJLoader     { std::vector<Triangle> load(const std::string & filename); }
namespace PLYLoader     { std::vector<Triangle> load(const std::string & filename); }
namespace SerializedLoader { std::vector<Triangle> load(const std::string & filename, int shape_index); }
namespace MitshairLoader   { std::vector<Triangle> load(const std::string & filename, float radius); }
namespace MitsubaLoader { void load(const std::string & filename, Scene & scene); }
namespace TextureLoader {
	bool load(const std::string & filename, Texture * texture);
	// One mip step with the box / lanczos / kaiser kernel of the reference (Src/Math/Mipmap.cpp)
	void downsample(MipmapFilterType filter, int w_src, int h_src, int w_dst, int h_dst, const Vector4 * src, Vector4 * dst, std::vector<Vector4> & temp);
}

namespace Geometry {
	std::vector<Triangle> rectangle(const Matrix4 & transform);
	std::vector<Triangle> cube     (const Matrix4 & transform);
	std::vector<Triangle> disk     (const Matrix4 & transform, int num_segments = 32);
	std::vector<Triangle> cylinder (const Matrix4 & transform, const Vector3 & p0, const Vector3 & p1, float radius, int num_segments = 32);
	std::vector<Triangle> sphere   (const Matrix4 & transform, int num_subdivisions = 3);
}
