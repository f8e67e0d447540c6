#include "Pathtracer.h"

#include <map>

void Pathtracer::gpu_init(int width, int height) {
	Integrator::gpu_init(width, height);
	lights_total_weight = 0.0f;
}

void Pathtracer::gpu_free() {
	Integrator::gpu_free();
}

// reference: Pathtracer::resize_init (Pathtracer.cpp:255-301)
void Pathtracer::resize_init(int width, int height) {
	screen_width  = width;
	screen_height = height;
	screen_pitch  = Math::round_up(width, 32);
	pixel_count   = width * height;

	if (ctx) check(rt_resize(ctx, width, height));
	aov_enable(AOVType::RADIANCE);

	scene.camera.resize(width, height);
	invalidated_camera = true;
	sample_index = 0;

	if (gpu_config.enable_svgf) { // SVGF needs these inputs (Pathtracer.cpp:331-334)
		aov_enable(AOVType::RADIANCE_DIRECT);
		aov_enable(AOVType::RADIANCE_INDIRECT);
		aov_enable(AOVType::ALBEDO);
	}
}

void Pathtracer::resize_free() { }

// Per light-emitting MeshData: triangle areas -> per-mesh CDF over triangles; per Mesh:
// weight = luminance(emission) * total area (reference: Pathtracer.cpp:384-500).
void Pathtracer::calc_light_power() {
	std::map<int, std::vector<Mesh *>> light_users; // mesh_data handle -> meshes, ascending handle order
	for (Mesh & mesh : scene.meshes) {
		const Material & material = scene.asset_manager.get_material(mesh.material_handle);
		if (material.is_light()) light_users[mesh.mesh_data_handle.handle].push_back(&mesh);
		else mesh.light.weight = 0.0f;
	}

	light_triangle_indices.clear();
	light_triangle_cumulative_probability.clear();
	std::vector<double> triangle_area;

	for (auto & [mesh_data_handle, meshes] : light_users) {
		const MeshData & mesh_data = scene.asset_manager.mesh_datas[mesh_data_handle];

		size_t first = light_triangle_indices.size();
		size_t count = mesh_data.triangles.size();
		double total_area = 0.0;
		for (size_t t = 0; t < count; t++) {
			const Triangle & tri = mesh_data.triangles[t];
			float area = 0.5f * Vector3::length(Vector3::cross(tri.position_1 - tri.position_0, tri.position_2 - tri.position_0));
			light_triangle_indices.push_back(reverse_indices[mesh_data_triangle_offsets[mesh_data_handle] + int(t)]);
			triangle_area.push_back(area);
			total_area += area;
		}

		for (Mesh * mesh : meshes) {
			const Material & material = scene.asset_manager.get_material(mesh->material_handle);
			mesh->light.weight               = Math::luminance(material.emission) * float(total_area);
			mesh->light.first_triangle_index = int(first);
			mesh->light.triangle_count       = int(count);
		}

		light_triangle_cumulative_probability.resize(first + count);
		double cumulative = 0.0;
		for (size_t i = first; i < first + count; i++) {
			cumulative += triangle_area[i] / total_area;
			light_triangle_cumulative_probability[i] = float(cumulative);
		}
		for (size_t i = first; i < first + count; i++) light_triangle_cumulative_probability[i] /= float(cumulative);
	}

	if (!light_triangle_indices.empty()) invalidated_scene = true; // mesh tables are filled once the TLAS exists
}

// Second CDF level, over light meshes in TLAS order, weight scaled by scale^2
// (reference: Pathtracer.cpp:503-534).
void Pathtracer::calc_light_mesh_weights() {
	light_mesh_cumulative_probability.clear();
	light_mesh_triangle_span.clear();
	light_mesh_transform_indices.clear();

	// Order of the light meshes in the CDF: TLAS order, as in the reference -- unless the TLAS is built on the device, whose
	// order the host does not know; then scene order, with scene indices as transform indices (the device maps them)
	// Flattened static geometry: the instance tables' rows are in another order (TLAS leaves, then the flattened members), but the
	// light meshes keep the ORDER the reference gives them -- that of its own top-level tree's leaves (reference_tlas_order) --:
	// the position of a light in the distribution decides which light a random number selects, i.e. it enters the image
	double total = 0.0;
	size_t rows = tlas_on_device ? scene.meshes.size() : tlas.indices.size();   // the instance tables' rows
	std::vector<int> row_of_mesh;
	const bool reference_order = !tlas_on_device && !reference_tlas_order.empty();
	if (reference_order) {
		row_of_mesh.assign(scene.meshes.size(), -1);
		for (size_t i = 0; i < rows; i++) if (tlas.indices[i] >= 0) row_of_mesh[size_t(tlas.indices[i])] = int(i);
	}
	size_t entries = reference_order ? reference_tlas_order.size() : rows;
	for (size_t k = 0; k < entries; k++) {
		size_t i = reference_order ? size_t(row_of_mesh[size_t(reference_tlas_order[k])]) : k;
		if (!tlas_on_device && tlas.indices[i] < 0) continue;   // the row of the flattened static geometry: its members have rows of their own
		const Mesh & mesh = scene.meshes[tlas_on_device ? int(i) : tlas.indices[i]];
		if (mesh.light.weight > 0.0f) {
			total += double(mesh.light.weight * mesh.scale * mesh.scale);
			light_mesh_cumulative_probability.push_back(float(total));
			light_mesh_triangle_span.push_back(mesh.light.first_triangle_index);
			light_mesh_triangle_span.push_back(mesh.light.first_triangle_index + mesh.light.triangle_count - 1);
			light_mesh_transform_indices.push_back(int(i));
		}
	}
	for (float & p : light_mesh_cumulative_probability) p /= float(total);
	lights_total_weight = float(total);

	if (ctx) check(rt_upload_lights(ctx,
		light_triangle_indices.data(), light_triangle_cumulative_probability.data(), light_triangle_indices.size(),
		light_mesh_cumulative_probability.data(), light_mesh_triangle_span.data(), light_mesh_transform_indices.data(), light_mesh_transform_indices.size(),
		lights_total_weight));
}

// reference: Pathtracer::update (Pathtracer.cpp:536-736)
void Pathtracer::update(float delta) {
	if (invalidated_sky) {
		invalidated_sky = false;
		if (ctx) check(rt_set_sky(ctx, &scene.sky.data[0].x, scene.sky.width, scene.sky.height, scene.sky.scale));
		sample_index = 0;
	}

	if (invalidated_materials) {
		const std::vector<Material> & scene_materials = scene.asset_manager.materials;
		material_types.assign(scene_materials.size(), 0);
		materials.assign(scene_materials.size(), DeviceMaterial());

		for (size_t i = 0; i < scene_materials.size(); i++) {
			const Material & m = scene_materials[i];
			material_types[i] = (unsigned char)m.type;
			switch (m.type) {
				case Material::Type::LIGHT:
					materials[i].light.emission = m.emission;
					break;
				case Material::Type::DIFFUSE:
					materials[i].diffuse.diffuse    = m.diffuse;
					materials[i].diffuse.texture_id = m.texture_handle.handle;
					break;
				case Material::Type::PLASTIC:
					materials[i].plastic.diffuse          = m.diffuse;
					materials[i].plastic.texture_id       = m.texture_handle.handle;
					materials[i].plastic.linear_roughness = m.linear_roughness;
					break;
				case Material::Type::DIELECTRIC:
					materials[i].dielectric.medium_id        = m.medium_handle.handle;
					materials[i].dielectric.ior              = Math::max(m.index_of_refraction, 1.0001f);
					materials[i].dielectric.linear_roughness = m.linear_roughness;
					break;
				case Material::Type::CONDUCTOR:
					materials[i].conductor.eta              = m.eta;
					materials[i].conductor.linear_roughness = m.linear_roughness;
					materials[i].conductor.k                = m.k;
					break;
			}
		}
		if (ctx) check(rt_upload_materials(ctx, material_types.data(), materials.data(), materials.size()));

		bool had_lights = scene.has_lights;
		scene.check_materials();
		if (had_lights != scene.has_lights) {
			if (scene.has_lights) {
				invalidated_scene = true;
			} else {
				lights_total_weight = 0.0f;
				for (Mesh & mesh : scene.meshes) mesh.light.weight = 0.0f;
				if (ctx) check(rt_upload_lights(ctx, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0.0f));
			}
		}
		if (scene.has_lights) calc_light_power();

		sample_index = 0;
		invalidated_materials = false;
	}

	if (invalidated_mediums) {
		const std::vector<Medium> & scene_media = scene.asset_manager.media;
		media.assign(scene_media.size(), DeviceMedium());
		for (size_t i = 0; i < scene_media.size(); i++) {
			scene_media[i].to_sigmas(media[i].sigma_a, media[i].sigma_s);
			media[i].g = scene_media[i].g;
		}
		if (ctx && !media.empty()) check(rt_upload_media(ctx, media.data(), media.size()));
		sample_index = 0;
		invalidated_mediums = false;
	}

	// The per-mesh light tables (CDF, transform indices) are in TLAS order. With enable_scene_update the TLAS is
	// rebuilt every frame (Integrator::update sets invalidated_scene only after this point), so the tables follow
	// it every frame too. The reference captures the flag here as well but without this term (Pathtracer.cpp:694),
	// i.e. after a TLAS reorder its NEE samples stale instance transforms: a reference flaw, not reproduced.
	bool invalidated_light_mesh_weights = invalidated_scene || cpu_config.enable_scene_update;

	if (gpu_config.enable_svgf) {
		memcpy(&svgf_matrices[0],  scene.camera.view_projection.cells,      64);
		memcpy(&svgf_matrices[16], scene.camera.view_projection_prev.cells, 64);
		if (ctx) check(rt_set_svgf_matrices(ctx, &svgf_matrices[0], &svgf_matrices[16]));
		if (invalidated_aovs && !aov_is_enabled(AOVType::ALBEDO)) aov_enable(AOVType::ALBEDO);
	}

	Integrator::update(delta);

	if (gpu_config.enable_svgf && ctx) {
		// the matrices of THIS frame are only known after camera.update(); the reference uploads
		// the pre-update pair (Pathtracer.cpp:707-717), which lags by one frame for a moving camera
		// and is identical for a static one.
	}

	if (invalidated_light_mesh_weights) {
		calc_light_mesh_weights();
		if (!gpu_config.enable_svgf) sample_index = 0;
	}
}

void Pathtracer::render() {
	require_device();
	check(rt_render_sample(ctx, sample_index));
	if (pixel_query_status == PixelQueryStatus::PENDING) pixel_query_status = PixelQueryStatus::OUTPUT_READY; // Pathtracer.cpp:852-854
}

void Pathtracer::render_samples(int count) {
	require_device();
	if (count < 1) return;
	if (pixel_query_status == PixelQueryStatus::PENDING) pixel_query_status = PixelQueryStatus::OUTPUT_READY;
	if (gpu_config.enable_svgf) { // SVGF frames feed each other's history: one at a time
		for (int i = 0; i < count; i++) { if (i) sample_index++; check(rt_render_sample(ctx, sample_index)); }
		return;
	}
	while (count > 0) {
		int n = count < 16 ? count : 16;
		check(rt_render_samples(ctx, sample_index, n));
		sample_index += n - 1;
		count -= n;
		if (count > 0) sample_index++;
	}
}
