#include "Scene.h"
#include "BVHCache.h"
#include "Parser.h"

#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <stdexcept>
#include <thread>

GPUConfig gpu_config;
CPUConfig cpu_config;

// ---- Mesh -------------------------------------------------------------------------------------

void Mesh::calc_aabb(const Scene & scene) {
	const MeshData & mesh_data = scene.asset_manager.get_mesh_data(mesh_data_handle);
	aabb_untransformed = AABB::create_empty();
	for (const Triangle & t : mesh_data.triangles) aabb_untransformed.expand(t.get_aabb());
}

// T*R*S and its inverse S^-1 * R^-1 * T^-1 (reference: Renderer/Mesh.cpp:16-34)
void Mesh::update() {
	transform_prev = transform;
	transform =
		Matrix4::create_translation(position) *
		Matrix4::create_rotation(rotation) *
		Matrix4::create_scale(scale);
	transform_inv =
		Matrix4::create_scale(1.0f / scale) *
		Matrix4::create_rotation(Quaternion::conjugate(rotation)) *
		Matrix4::create_translation(-position);

	aabb = AABB::transform(aabb_untransformed, transform);
	const float * box = &aabb.min.x;
	for (int k = 0; k < 6; k++) {
		// (checked before fix_if_needed, whose widening loop would not terminate on NaN / infinite extents)
		if (!std::isfinite(box[k])) throw std::runtime_error("mesh '" + name + "' has a non-finite transform or bounds");
	}
	aabb.fix_if_needed();
}

bool Mesh::has_identity_transform() const {
	constexpr float eps = 1e-6f;
	return
		Math::approx_equal(scale, 1.0f, eps) &&
		Math::approx_equal(position.x, 0.0f, eps) && Math::approx_equal(position.y, 0.0f, eps) && Math::approx_equal(position.z, 0.0f, eps) &&
		Math::approx_equal(rotation.x, 0.0f, eps) && Math::approx_equal(rotation.y, 0.0f, eps) && Math::approx_equal(rotation.z, 0.0f, eps) &&
		(Math::approx_equal(rotation.w, 1.0f, eps) || Math::approx_equal(rotation.w, -1.0f, eps)); // quaternion double cover
}

// ---- AssetManager -----------------------------------------------------------------------------

AssetManager::AssetManager() {
	Material default_material;
	default_material.name    = "Default";
	default_material.diffuse = Vector3(1.0f, 0.0f, 1.0f);
	add_material(std::move(default_material));

	Medium default_medium;
	default_medium.name = "Default";
	add_medium(std::move(default_medium));
}

Handle<MeshData> AssetManager::add_mesh_data(const std::string & filename, FallbackLoader loader) {
	return add_mesh_data(filename, BVHCache::get_bvh_filename(filename), std::move(loader));
}

Handle<MeshData> AssetManager::add_mesh_data(const std::string & filename, const std::string & bvh_filename, FallbackLoader loader) {
	auto it = mesh_data_cache.find(filename);
	if (it != mesh_data_cache.end()) return it->second;

	Handle<MeshData> handle { int(mesh_datas.size()) };
	mesh_datas.emplace_back();
	mesh_data_cache[filename] = handle;
	pending_meshes.push_back({ handle.handle, filename, std::move(loader), bvh_filename });
	return handle;
}

static void build_blas(MeshData & mesh_data);

// One mesh file: triangles and tree come from the "<file>.bvh" cache when caching is on and the
// cache is current, else from the loader and the builder (reference: AssetManager.cpp:57-95). The
// cache holds the tree kind of the current bvh_type -- SAH or spatial-split -- so with SBVH
// selected the SAH tree (which the CWBVH is made from) is still built here.
static void load_mesh_file(MeshData & mesh_data, const std::string & filename, const std::string & bvh_filename, const AssetManager::FallbackLoader & loader) {
	mesh_data.from_file    = true;
	mesh_data.filename     = filename;
	mesh_data.bvh_filename = bvh_filename;

	bool use_cache = cpu_config.enable_bvh_cache;
	BVH2 cached;
	bool cache_hit = use_cache && BVHCache::try_to_load(filename, bvh_filename, &mesh_data.triangles, &cached);
	if (!cache_hit) mesh_data.triangles = loader(filename);

	for (const Triangle & triangle : mesh_data.triangles) { // NaN / infinite vertices would poison every SAH comparison of the builders
		const Vector3 * p = &triangle.position_0;
		for (int v = 0; v < 3; v++) {
			if (!std::isfinite(p[v].x) || !std::isfinite(p[v].y) || !std::isfinite(p[v].z)) throw ParseError("'" + filename + "': mesh has a non-finite vertex position");
		}
	}

	bool cache_is_sbvh = BVHCache::underlying_bvh_type() == BVHType::SBVH;
	if (cache_hit && cache_is_sbvh) mesh_data.sbvh = std::move(cached);
	if (cache_hit && !cache_is_sbvh) {
		mesh_data.bvh2 = std::move(cached);
		BVH8Converter(mesh_data.bvh8, mesh_data.bvh2).convert();
		BVH4Converter(mesh_data.bvh4, mesh_data.bvh2).convert();
		return;
	}
	build_blas(mesh_data);
	if (use_cache && !cache_is_sbvh) BVHCache::save(bvh_filename, mesh_data.triangles, mesh_data.bvh2);
}

static void build_blas(MeshData & mesh_data) {
	if (mesh_data.triangles.empty()) {
		// An empty mesh is replaced by one dummy triangle (reference: AssetManager.cpp:64-78)
		mesh_data.triangles.push_back(Triangle(
			Vector3(-1.0f, -1.0f, 0.0f), Vector3(0.0f, +1.0f, 0.0f), Vector3(+1.0f, -1.0f, 0.0f),
			Vector3(0.0f, 0.0f, 1.0f), Vector3(0.0f, 0.0f, 1.0f), Vector3(0.0f, 0.0f, 1.0f),
			Vector2(0.0f, 1.0f), Vector2(0.5f, 0.0f), Vector2(1.0f, 1.0f)));
	}
	mesh_data.bvh2 = BVH::create_sah_from_triangles(mesh_data.triangles);
	BVH8Converter(mesh_data.bvh8, mesh_data.bvh2).convert();
	BVH4Converter(mesh_data.bvh4, mesh_data.bvh2).convert();
}

void MeshData::prepare_device_bvh(BVHType type) {
	if (type == BVHType::BVH8 || device_bvh_type == int(type)) return;
	if (type == BVHType::SBVH) {
		if (sbvh.nodes.empty()) {
			SBVHBuilder(sbvh, triangles.size()).build(triangles);
			if (cpu_config.enable_bvh_optimization) BVHOptimizer::optimize(sbvh);
			if (from_file && cpu_config.enable_bvh_cache && cpu_config.bvh_type == BVHType::SBVH) BVHCache::save(bvh_filename, triangles, sbvh);
		}
		device_bvh2 = sbvh;
	} else {
		device_bvh2 = bvh2;
	}
	if (from_file) BVHCollapser::collapse(device_bvh2);
	device_bvh4 = BVH4();
	if (type == BVHType::BVH4) BVH4Converter(device_bvh4, device_bvh2).convert();
	device_bvh_type = int(type);
}

void AssetManager::prepare_device_bvhs(BVHType type) {
	wait_until_loaded();
	std::atomic<size_t> next { 0 };
	std::string failure;
	std::mutex  failure_mutex;
	auto work = [&]() {
		while (true) {
			size_t i = next.fetch_add(1);
			if (i >= mesh_datas.size()) break;
			try {
				mesh_datas[i].prepare_device_bvh(type);
			} catch (const std::exception & e) {
				std::lock_guard<std::mutex> lock(failure_mutex);
				failure = e.what();
			}
		}
	};
	unsigned worker_count = std::max(1u, std::thread::hardware_concurrency());
	std::vector<std::thread> workers;
	for (unsigned w = 1; w < worker_count; w++) workers.emplace_back(work);
	work();
	for (std::thread & t : workers) t.join();
	if (!failure.empty()) throw std::runtime_error(failure);
}

Handle<MeshData> AssetManager::add_mesh_data(std::vector<Triangle> triangles) {
	Handle<MeshData> handle { int(mesh_datas.size()) };
	mesh_datas.emplace_back();
	mesh_datas.back().triangles = std::move(triangles);
	pending_meshes.push_back({ handle.handle, std::string(), nullptr, std::string() });
	return handle;
}

Handle<Material> AssetManager::add_material(Material material) {
	Handle<Material> handle { int(materials.size()) };
	materials.emplace_back(std::move(material));
	return handle;
}

Handle<Medium> AssetManager::add_medium(Medium medium) {
	Handle<Medium> handle { int(media.size()) };
	media.emplace_back(std::move(medium));
	return handle;
}

Handle<Texture> AssetManager::add_texture(const std::string & filename, const std::string & name) {
	auto it = texture_cache.find(filename);
	if (it != texture_cache.end()) return it->second;

	Handle<Texture> handle { int(textures.size()) };
	textures.emplace_back();
	textures.back().name = name;
	texture_cache[filename] = handle;
	pending_textures.push_back({ handle.handle, filename });
	return handle;
}

void AssetManager::wait_until_loaded() {
	if (assets_loaded) return;

	unsigned worker_count = std::max(1u, std::thread::hardware_concurrency());
	auto t0 = std::chrono::steady_clock::now();
	{
		std::atomic<size_t> next { 0 };
		std::string failure;
		std::mutex  failure_mutex;
		auto work = [&]() {
			while (true) {
				size_t i = next.fetch_add(1);
				if (i >= pending_meshes.size()) break;
				PendingMesh & job = pending_meshes[i];
				MeshData & mesh_data = mesh_datas[job.handle];
				try {
					if (job.loader) load_mesh_file(mesh_data, job.filename, job.bvh_filename, job.loader);
					else            build_blas(mesh_data);
				} catch (const std::exception & e) { // a worker must not let it escape: report it from the calling thread
					std::lock_guard<std::mutex> lock(failure_mutex);
					if (failure.empty()) failure = e.what();
				}
			}
		};
		std::vector<std::thread> workers;
		for (unsigned w = 1; w < worker_count; w++) workers.emplace_back(work);
		work();
		for (std::thread & t : workers) t.join();
		if (!failure.empty()) {
			pending_meshes.clear();
			throw ParseError(failure);
		}
	}
	bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	{
		std::atomic<size_t> next { 0 };
		auto work = [&]() {
			while (true) {
				size_t i = next.fetch_add(1);
				if (i >= pending_textures.size()) break;
				PendingTexture & job = pending_textures[i];
				Texture & texture = textures[job.handle];
				bool loaded = false;
				try { loaded = TextureLoader::load(job.filename, &texture); } catch (const std::exception &) { /* e.g. out of memory on a hostile header: fall back */ }
				if (!loaded) {
					fprintf(stderr, "WARNING: Failed to load Texture '%s'!\n", job.filename.c_str());
					// 1x1 fallback (reference: AssetManager.cpp:157-169). The reference means it to be pink but stores the
					// colour as a float4 in a texture it then uploads as RGBA8: the one texel the device gets is the first
					// four bytes of the float 1.0f. Same texel here, so that a scene with a missing map renders alike.
					texture.width = texture.height = 1;
					texture.texels = { 0x00, 0x00, 0x80, 0x3f };
					texture.mip_offsets = { 0 };
				}
			}
		};
		std::vector<std::thread> workers;
		for (unsigned w = 1; w < worker_count; w++) workers.emplace_back(work);
		work();
		for (std::thread & t : workers) t.join();
	}
	pending_meshes.clear();
	pending_textures.clear();
	mesh_data_cache.clear();
	texture_cache.clear();
	assets_loaded = true;
}

// ---- Sky --------------------------------------------------------------------------------------

// Radiance RGBE (.hdr) reader: "#?RADIANCE" header, -Y h +X w, flat or new-style RLE scanlines.
void Sky::load(const std::string & filename) {
	auto fallback = [this]() { width = height = 1; data = { Vector4(1.0f, 1.0f, 1.0f, 0.0f) }; };
	if (filename.empty()) { fallback(); return; }

	FILE * f = fopen(filename.c_str(), "rb");
	if (!f) {
		fprintf(stderr, "WARNING: unable to load hdr Sky from file '%s', using a constant white sky\n", filename.c_str());
		fallback();
		return;
	}
	char line[256];
	bool have_size = false;
	while (fgets(line, sizeof(line), f)) {
		if (sscanf(line, "-Y %d +X %d", &height, &width) == 2) { have_size = true; break; }
	}
	if (!have_size || width <= 0 || height <= 0 || width > (1 << 15) || height > (1 << 15)) { fclose(f); fallback(); return; }

	data.resize(size_t(width) * height);
	std::vector<unsigned char> scan(size_t(width) * 4);
	for (int y = 0; y < height; y++) {
		unsigned char head[4];
		if (fread(head, 1, 4, f) != 4) break;
		bool rle = head[0] == 2 && head[1] == 2 && !(head[2] & 0x80) && ((head[2] << 8) | head[3]) == width && width >= 8 && width < 32768;
		if (rle) {
			for (int c = 0; c < 4; c++) {
				int x = 0;
				while (x < width) {
					int count = fgetc(f);
					if (count > 128) { int v = fgetc(f); count -= 128; while (count-- > 0 && x < width) scan[size_t(x++) * 4 + c] = (unsigned char)v; }
					else             { while (count-- > 0 && x < width) scan[size_t(x++) * 4 + c] = (unsigned char)fgetc(f); }
				}
			}
		} else {
			memcpy(scan.data(), head, 4);
			if (fread(scan.data() + 4, 1, size_t(width - 1) * 4, f) != size_t(width - 1) * 4) break;
		}
		for (int x = 0; x < width; x++) {
			const unsigned char * p = &scan[size_t(x) * 4];
			float s = p[3] ? ldexpf(1.0f, int(p[3]) - (128 + 8)) : 0.0f;
			data[size_t(x) + size_t(y) * width] = Vector4(p[0] * s, p[1] * s, p[2] * s, 0.0f);
		}
	}
	fclose(f);
}

// ---- Scene ------------------------------------------------------------------------------------

static std::string file_extension(const std::string & filename) {
	size_t dot = filename.find_last_of('.');
	return dot == std::string::npos ? std::string() : filename.substr(dot + 1);
}

Scene::Scene() : camera(Math::deg_to_rad(85.0f)) {
	for (const std::string & scene_filename : cpu_config.scene_filenames) {
		std::string ext = file_extension(scene_filename);
		if (ext == "obj") {
			add_mesh(scene_filename, asset_manager.add_mesh_data(scene_filename, OBJLoader::load));
		} else if (ext == "ply") {
			add_mesh(scene_filename, asset_manager.add_mesh_data(scene_filename, PLYLoader::load));
		} else if (ext == "xml") {
			MitsubaLoader::load(scene_filename, *this);
		} else {
			throw ParseError("'" + scene_filename + "': file format is not supported (expected .xml, .obj or .ply)");
		}
	}
	sky.load(cpu_config.sky_filename);
}

Mesh & Scene::add_mesh(std::string name, Handle<MeshData> mesh_data_handle, Handle<Material> material_handle) {
	meshes.emplace_back(std::move(name), mesh_data_handle, material_handle);
	return meshes.back();
}

void Scene::check_materials() {
	has_diffuse = has_plastic = has_dielectric = has_conductor = has_lights = false;
	for (const Material & material : asset_manager.materials) {
		switch (material.type) {
			case Material::Type::DIFFUSE:    has_diffuse    = true; break;
			case Material::Type::PLASTIC:    has_plastic    = true; break;
			case Material::Type::DIELECTRIC: has_dielectric = true; break;
			case Material::Type::CONDUCTOR:  has_conductor  = true; break;
			case Material::Type::LIGHT:      has_lights    |= material.is_light(); break;
		}
	}
}

void Scene::update(float delta) {
	(void)delta;
	for (Mesh & mesh : meshes) mesh.update();
}
