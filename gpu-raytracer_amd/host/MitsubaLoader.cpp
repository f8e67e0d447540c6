// Mitsuba 0.5 XML scene -> Scene. Supported subset and every default follow the
// reference loader (Src/Assets/Mitsuba/MitsubaLoader.cpp): bsdf (diffuse, plastic,
// roughplastic, roughdiffuse, phong, (rough)dielectric, thindielectric, (rough)conductor,
// difftrans; twosided/mask/bumpmap/coating are peeled), shapes (obj, rectangle, cube,
// disk, cylinder, sphere, shapegroup/instance), homogeneous media, perspective/thinlens
// sensors, film size, integrator maxDepth, area/point/envmap emitters, <include>.
// Scene XML overrides the CLI: film width/height and maxDepth are written into
// cpu_config / gpu_config (MitsubaLoader.cpp:610-616).
#include "Scene.h"
#include "XMLParser.h"

#include <cstdio>
#include <map>

namespace {

struct ShapeGroup {
	Handle<MeshData> mesh_data_handle;
	Handle<Material> material_handle;
};

struct LoadState {
	std::map<std::string, ShapeGroup>       shape_groups;
	std::map<std::string, Handle<Material>> materials;
	std::map<std::string, Handle<Texture>>  textures;
	std::string directory; // of the xml file, with trailing separator
};

void warn(const XMLNode & node, const std::string & message) {
	fprintf(stderr, "%s: WARNING: %s\n", node.location.c_str(), message.c_str());
}

std::string directory_of(const std::string & filename) {
	size_t slash = filename.find_last_of("/\\");
	return slash == std::string::npos ? std::string("./") : filename.substr(0, slash + 1);
}

std::string strip_directory(const std::string & filename) {
	size_t slash = filename.find_last_of("/\\");
	return slash == std::string::npos ? filename : filename.substr(slash + 1);
}

// Scene files written on Windows use "textures\\name.tga" (Data/Sponza/scene.xml:13)
std::string join_path(const std::string & directory, std::string_view relative) {
	std::string rel(relative);
	for (size_t i = 0; i < rel.size(); i++) {
		if (rel[i] == '\\') {
			if (i + 1 < rel.size() && rel[i + 1] == '\\') rel.erase(i, 1);
			rel[i] = '/';
		}
	}
	return directory + rel;
}

Handle<Texture> parse_texture(const XMLNode * node, LoadState & state, Scene & scene, Vector3 * rgb) {
	std::string_view type = node->get_attribute_value("type");

	if (type == "scale") {
		if (const XMLNode * scale = node->get_child_by_name("scale")) {
			if      (scale->tag == "float") *rgb *= scale->require_attribute("value").as_float();
			else if (scale->tag == "rgb")   *rgb *= scale->require_attribute("value").as_vector3();
			else warn(*scale, "invalid scale tag <" + scale->tag + ">");
		}
		node = node->get_child_by_tag("texture");
		if (!node) return Handle<Texture> { INVALID };
		type = node->get_attribute_value("type");
	}

	if (type == "bitmap") {
		std::string filename = join_path(state.directory, node->require_child_by_name("filename").get_attribute_value("value"));
		Handle<Texture> handle = scene.asset_manager.add_texture(filename, strip_directory(filename));
		if (const XMLAttribute * id = node->get_attribute("id")) state.textures[id->value] = handle;
		return handle;
	}
	warn(*node, "only bitmap textures are supported");
	return Handle<Texture> { INVALID };
}

void parse_rgb_or_texture(const XMLNode * node, const char * name, LoadState & state, Scene & scene, Vector3 * rgb, Handle<Texture> * texture_handle) {
	const XMLNode * colour = node->get_child_by_name(name);
	if (!colour) { *rgb = Vector3(1.0f); return; }

	if (colour->tag == "rgb") {
		*rgb = colour->get_attribute_optional("value", Vector3(1.0f));
	} else if (colour->tag == "srgb") {
		*rgb = colour->get_attribute_optional("value", Vector3(1.0f));
		rgb->x = Math::gamma_to_linear(rgb->x);
		rgb->y = Math::gamma_to_linear(rgb->y);
		rgb->z = Math::gamma_to_linear(rgb->z);
	} else if (colour->tag == "texture") {
		*texture_handle = parse_texture(colour, state, scene, rgb);
		if (const XMLNode * scale = colour->get_child_by_name("scale")) *rgb = scale->get_attribute_optional("value", Vector3(1.0f));
	} else if (colour->tag == "ref") {
		std::string id(colour->get_attribute_value("id"));
		auto it = state.textures.find(id);
		if (it != state.textures.end()) *texture_handle = it->second;
		else warn(*colour, "invalid texture ref '" + id + "'");
	}
}

// Every child of <transform> is applied on the left, in document order.
Matrix4 parse_transform_matrix(const XMLNode * node) {
	Matrix4 world;
	const XMLNode * transform = node->get_child_by_tag("transform");
	if (!transform) return world;

	for (const XMLNode & op : transform->children) {
		if (op.tag == "matrix") {
			world = op.require_attribute("value").as_matrix4() * world;
		} else if (op.tag == "lookat") {
			Vector3 origin = op.get_attribute_optional("origin", Vector3(0.0f, 0.0f,  0.0f));
			Vector3 target = op.get_attribute_optional("target", Vector3(0.0f, 0.0f, -1.0f));
			Vector3 up     = op.get_attribute_optional("up",     Vector3(0.0f, 1.0f,  0.0f));
			world = Matrix4::create_translation(origin) * Matrix4::create_rotation(Quaternion::look_rotation(target - origin, up)) * world;
		} else if (op.tag == "scale") {
			if (const XMLAttribute * uniform = op.get_attribute("value")) {
				world = Matrix4::create_scale(uniform->as_float()) * world;
			} else {
				world = Matrix4::create_scale(op.get_attribute_optional("x", 1.0f), op.get_attribute_optional("y", 1.0f), op.get_attribute_optional("z", 1.0f)) * world;
			}
		} else if (op.tag == "rotate") {
			float x = op.get_attribute_optional("x", 0.0f), y = op.get_attribute_optional("y", 0.0f), z = op.get_attribute_optional("z", 0.0f);
			if (x == 0.0f && y == 0.0f && z == 0.0f) {
				warn(op, "rotation without axis specified");
			} else {
				float angle = op.get_attribute_optional("angle", 0.0f);
				world = Matrix4::create_rotation(Quaternion::axis_angle(Vector3(x, y, z), Math::deg_to_rad(angle))) * world;
			}
		} else if (op.tag == "translate") {
			world = Matrix4::create_translation(Vector3(op.get_attribute_optional("x", 0.0f), op.get_attribute_optional("y", 0.0f), op.get_attribute_optional("z", 0.0f))) * world;
		} else {
			warn(op, "node <" + op.tag + "> is not a valid transformation");
		}
	}
	return world;
}

void parse_transform(const XMLNode * node, Vector3 * position, Quaternion * rotation, float * scale, const Vector3 & forward = Vector3(0.0f, 0.0f, 1.0f)) {
	Matrix4::decompose(parse_transform_matrix(node), position, rotation, scale, forward);
}

bool lookup_known_ior(std::string_view name, float * ior) {
	// Mitsuba 0.5 documentation, page 58
	static const struct { const char * name; float ior; } table[] = {
		{ "vacuum", 1.0f }, { "helium", 1.00004f }, { "hydrogen", 1.00013f }, { "air", 1.00028f }, { "carbon dioxide", 1.00045f },
		{ "water", 1.3330f }, { "acetone", 1.36f }, { "ethanol", 1.361f }, { "carbon tetrachloride", 1.461f }, { "glycerol", 1.4729f },
		{ "benzene", 1.501f }, { "silicone oil", 1.52045f }, { "bromine", 1.661f }, { "water ice", 1.31f }, { "fused quartz", 1.458f },
		{ "pyrex", 1.470f }, { "acrylic glass", 1.49f }, { "polypropylene", 1.49f }, { "bk7", 1.5046f }, { "sodium chloride", 1.544f },
		{ "amber", 1.55f }, { "pet", 1.575f }, { "diamond", 2.419f }
	};
	for (const auto & entry : table) if (name == entry.name) { *ior = entry.ior; return true; }
	return false;
}

float parse_ior(const XMLNode * bsdf, const char * name, float fallback) {
	const XMLNode * child = bsdf->get_child_by_name(name);
	if (child && child->tag == "string") {
		float ior = 0.0f;
		std::string_view ior_name = child->get_attribute_value("value");
		if (!lookup_known_ior(ior_name, &ior)) throw ParseError(child->location + ": index of refraction not known for '" + std::string(ior_name) + "'");
		return ior;
	}
	return bsdf->get_child_value_optional(name, fallback);
}

Handle<Material> parse_material(const XMLNode * node, Scene & scene, LoadState & state) {
	Material material;
	const XMLNode * bsdf;

	if (node->tag != "bsdf") {
		// A shape: an <emitter> child wins, then <ref>, then an inline <bsdf>
		if (const XMLNode * emitter = node->get_child_by_tag("emitter")) {
			material.type = Material::Type::LIGHT;
			material.name = "emitter";
			material.emission = emitter->require_child_by_name("radiance").require_attribute("value").as_vector3();
			return scene.asset_manager.add_material(std::move(material));
		}
		if (const XMLNode * ref = node->get_child_by_tag("ref")) {
			std::string id(ref->get_attribute_value("id"));
			auto it = state.materials.find(id);
			if (it != state.materials.end()) return it->second;
			warn(*ref, "invalid material ref '" + id + "'");
			return Handle<Material>::get_default();
		}
		bsdf = node->get_child_by_tag("bsdf");
		if (!bsdf) { warn(*node, "unable to parse BSDF"); return Handle<Material>::get_default(); }
	} else {
		bsdf = node;
	}

	const XMLAttribute * name = bsdf->get_attribute("id");
	const XMLNode * inner = bsdf;
	std::string_view inner_type = inner->get_attribute_value("type");

	// Only the innermost BSDF of adapter BSDFs matters
	while (inner_type == "twosided" || inner_type == "mask" || inner_type == "bumpmap" || inner_type == "coating") {
		if (const XMLNode * child = inner->get_child_by_tag("bsdf")) {
			inner = child;
		} else if (const XMLNode * ref = inner->get_child_by_tag("ref")) {
			std::string id(ref->get_attribute_value("id"));
			auto it = state.materials.find(id);
			if (it != state.materials.end()) return it->second;
			warn(*ref, "invalid material ref '" + id + "'");
			return Handle<Material>::get_default();
		} else {
			return Handle<Material>::get_default();
		}
		inner_type = inner->get_attribute_value("type");
		if (!name) name = inner->get_attribute("id");
	}
	material.name = name ? name->value : "Material";

	if (inner_type == "diffuse") {
		material.type = Material::Type::DIFFUSE;
		parse_rgb_or_texture(inner, "reflectance", state, scene, &material.diffuse, &material.texture_handle);
	} else if (inner_type == "conductor" || inner_type == "roughconductor") {
		material.type = Material::Type::CONDUCTOR;
		material.linear_roughness = inner_type == "conductor" ? 0.0f : inner->get_child_value_optional("alpha", 0.5f);
		const XMLNode * preset = inner->get_child_by_name("material");
		if (preset && preset->get_attribute_value("value") == "none") {
			material.eta = Vector3(0.0f);
			material.k   = Vector3(1.0f);
		} else {
			material.eta = inner->get_child_value_optional("eta", Vector3(1.33f));
			material.k   = inner->get_child_value_optional("k",   Vector3(1.0f));
		}
	} else if (inner_type == "plastic" || inner_type == "roughplastic" || inner_type == "roughdiffuse") {
		material.type = Material::Type::PLASTIC;
		parse_rgb_or_texture(inner, "diffuseReflectance", state, scene, &material.diffuse, &material.texture_handle);
		material.linear_roughness = inner_type == "plastic" ? 0.0f : inner->get_child_value_optional("alpha", 0.5f);
	} else if (inner_type == "phong") {
		material.type = Material::Type::PLASTIC;
		parse_rgb_or_texture(inner, "diffuseReflectance", state, scene, &material.diffuse, &material.texture_handle);
		float exponent = inner->get_child_value_optional("exponent", 1.0f);
		material.linear_roughness = powf(0.5f * exponent + 1.0f, 0.25f);
	} else if (inner_type == "thindielectric" || inner_type == "dielectric" || inner_type == "roughdielectric") {
		float int_ior = parse_ior(inner, "intIOR", 1.33f);
		float ext_ior = parse_ior(inner, "extIOR", 1.0f);
		material.type = Material::Type::DIELECTRIC;
		material.index_of_refraction = ext_ior == 0.0f ? int_ior : int_ior / ext_ior;
		material.linear_roughness = inner_type == "roughdielectric" ? inner->get_child_value_optional("alpha", 0.5f) : 0.0f;
	} else if (inner_type == "difftrans") {
		material.type = Material::Type::DIFFUSE;
		parse_rgb_or_texture(inner, "transmittance", state, scene, &material.diffuse, &material.texture_handle);
	} else {
		warn(*inner, "BSDF type '" + std::string(inner_type) + "' not supported");
		return Handle<Material>::get_default();
	}
	return scene.asset_manager.add_material(std::move(material));
}

Handle<Medium> parse_medium(const XMLNode * node, Scene & scene) {
	const XMLNode * xml_medium = node->get_child_by_tag("medium");
	if (!xml_medium) return Handle<Medium> { INVALID };

	std::string_view medium_type = xml_medium->get_attribute_value("type");
	if (medium_type != "homogeneous") {
		warn(*xml_medium, "medium type '" + std::string(medium_type) + "' not supported");
		return Handle<Medium> { INVALID };
	}

	Medium medium;
	if (const XMLAttribute * name = xml_medium->get_attribute("name")) medium.name = name->value;

	const XMLNode * xml_sigma_a = xml_medium->get_child_by_name("sigmaA");
	const XMLNode * xml_sigma_s = xml_medium->get_child_by_name("sigmaS");
	const XMLNode * xml_sigma_t = xml_medium->get_child_by_name("sigmaT");
	const XMLNode * xml_albedo  = xml_medium->get_child_by_name("albedo");

	Vector3 sigma_a, sigma_s;
	bool has_as = xml_sigma_a && xml_sigma_s, has_ta = xml_sigma_t && xml_albedo;
	if (!(has_as ^ has_ta)) {
		warn(*xml_medium, "provide EITHER sigmaA and sigmaS OR sigmaT and albedo");
	} else if (has_as) {
		sigma_a = xml_sigma_a->require_attribute("value").as_vector3();
		sigma_s = xml_sigma_s->require_attribute("value").as_vector3();
	} else {
		Vector3 sigma_t = xml_sigma_t->require_attribute("value").as_vector3();
		Vector3 albedo  = xml_albedo ->require_attribute("value").as_vector3();
		sigma_s = albedo * sigma_t;
		sigma_a = sigma_t - sigma_s;
	}

	float scale = xml_medium->get_child_value_optional("scale", 1.0f);
	medium.from_sigmas(scale * sigma_a, scale * sigma_s); // note: uses g = 0 here, the phase function is parsed afterwards

	if (const XMLNode * phase = xml_medium->get_child_by_tag("phase")) {
		std::string_view phase_type = phase->get_attribute_value("type");
		if      (phase_type == "isotropic") medium.g = 0.0f;
		else if (phase_type == "hg")        medium.g = phase->get_child_value_optional("g", 0.0f);
		else warn(*xml_medium, "phase function type '" + std::string(phase_type) + "' not supported");
	}
	return scene.asset_manager.add_medium(std::move(medium));
}

bool is_primitive_shape(std::string_view type) {
	return type == "rectangle" || type == "cube" || type == "disk" || type == "cylinder" || type == "sphere";
}

Handle<MeshData> parse_shape(const XMLNode * node, Scene & scene, LoadState & state, std::string * name) {
	std::string_view type = node->get_attribute_value("type");

	if (type == "obj" || type == "ply") { // reference: MitsubaLoader.cpp:434-442
		std::string filename = join_path(state.directory, node->require_child_by_name("filename").get_attribute_value("value"));
		*name = strip_directory(filename);
		return scene.asset_manager.add_mesh_data(filename, type == "obj" ? OBJLoader::load : PLYLoader::load);
	}
	if (is_primitive_shape(type)) {
		Matrix4 transform = parse_transform_matrix(node);
		std::vector<Triangle> triangles;
		if (type == "rectangle") {
			triangles = Geometry::rectangle(transform);
		} else if (type == "cube") {
			triangles = Geometry::cube(transform);
		} else if (type == "disk") {
			triangles = Geometry::disk(transform);
		} else if (type == "cylinder") {
			Vector3 p0 = node->get_child_value_optional("p0", Vector3(0.0f, 0.0f, 0.0f));
			Vector3 p1 = node->get_child_value_optional("p1", Vector3(0.0f, 0.0f, 1.0f));
			float radius = node->get_child_value_optional("radius", 1.0f);
			triangles = Geometry::cylinder(transform, p0, p1, radius);
		} else {
			float radius = node->get_child_value_optional("radius", 1.0f);
			Vector3 center(0.0f);
			if (const XMLNode * c = node->get_child_by_name("center")) {
				center = Vector3(c->get_attribute_optional("x", 0.0f), c->get_attribute_optional("y", 0.0f), c->get_attribute_optional("z", 0.0f));
			}
			transform = transform * Matrix4::create_translation(center) * Matrix4::create_scale(radius);
			triangles = Geometry::sphere(transform);
		}
		*name = std::string(type);
		return scene.asset_manager.add_mesh_data(std::move(triangles));
	}
	if (type == "serialized") { // reference: MitsubaLoader.cpp:487-500
		std::string relative = std::string(node->require_child_by_name("filename").get_attribute_value("value"));
		std::string filename = join_path(state.directory, relative);
		int shape_index = node->get_child_value_optional("shapeIndex", 0);
		*name = relative + "_" + std::to_string(shape_index);
		// one archive holds many meshes: the reference keys (and caches) each by "<archive>.shape_<i>.bvh"
		std::string key = filename + ".shape_" + std::to_string(shape_index) + ".bvh";
		return scene.asset_manager.add_mesh_data(key, key, [filename, shape_index](const std::string &) { return SerializedLoader::load(filename, shape_index); });
	}
	if (type == "hair") { // reference: MitsubaLoader.cpp:501-512
		std::string relative = std::string(node->require_child_by_name("filename").get_attribute_value("value"));
		std::string filename = join_path(state.directory, relative);
		*name = relative;
		float radius = node->get_child_value_optional("radius", 0.0025f);
		return scene.asset_manager.add_mesh_data(filename, [radius](const std::string & f) { return MitshairLoader::load(f, radius); });
	}
	warn(*node, "shape type '" + std::string(type) + "' not supported");
	return Handle<MeshData> { INVALID };
}

void walk(const XMLNode * node, Scene & scene, LoadState & state) {
	if (node->tag == "bsdf") {
		Handle<Material> handle = parse_material(node, scene, state);
		state.materials[scene.asset_manager.get_material(handle).name] = handle;
	} else if (node->tag == "texture") {
		Vector3 scale = 1.0f;
		parse_texture(node, state, scene, &scale);
	} else if (node->tag == "shape") {
		std::string_view type = node->get_attribute_value("type");
		if (type == "shapegroup") {
			if (!node->children.empty()) {
				const XMLNode * shape = node->get_child_by_tag("shape");
				if (!shape) throw ParseError(node->location + ": shapegroup needs a <shape> child");
				std::string name;
				Handle<MeshData> mesh_data_handle = parse_shape(shape, scene, state, &name);
				Handle<Material> material_handle  = parse_material(shape, scene, state);
				state.shape_groups[std::string(node->get_attribute_value("id"))] = { mesh_data_handle, material_handle };
			}
		} else if (type == "instance") {
			const XMLNode * ref = node->get_child_by_tag("ref");
			if (!ref) { warn(*node, "instance without ref"); return; }
			std::string id(ref->get_attribute_value("id"));
			auto it = state.shape_groups.find(id);
			if (it != state.shape_groups.end() && it->second.mesh_data_handle.handle != INVALID) {
				Mesh & mesh = scene.add_mesh(id, it->second.mesh_data_handle, it->second.material_handle);
				parse_transform(node, &mesh.position, &mesh.rotation, &mesh.scale);
			}
		} else {
			std::string name;
			Handle<MeshData> mesh_data_handle = parse_shape(node, scene, state, &name);
			Handle<Material> material_handle  = parse_material(node, scene, state);
			Handle<Medium>   medium_handle    = parse_medium(node, scene);

			if (material_handle.handle != INVALID) {
				Material & material = scene.asset_manager.get_material(material_handle);
				if (material.medium_handle.handle != INVALID && material.medium_handle.handle != medium_handle.handle) {
					// Material already bound to another medium: clone it for this shape
					Material copy = material;
					copy.medium_handle = medium_handle;
					material_handle = scene.asset_manager.add_material(std::move(copy));
				} else {
					material.medium_handle = medium_handle;
				}
			}
			if (mesh_data_handle.handle != INVALID) {
				Mesh & mesh = scene.add_mesh(std::move(name), mesh_data_handle, material_handle);
				// Primitive shapes have their transform baked into the vertices
				if (!is_primitive_shape(type)) parse_transform(node, &mesh.position, &mesh.rotation, &mesh.scale);
			}
		}
	} else if (node->tag == "sensor") {
		std::string_view camera_type = node->get_attribute_value("type");
		if (camera_type == "perspective" || camera_type == "perspective_rdist" || camera_type == "thinlens") {
			if (const XMLNode * fov = node->get_child_by_name("fov")) scene.camera.set_fov(Math::deg_to_rad(fov->require_attribute("value").as_float()));
			if (camera_type == "perspective") {
				scene.camera.aperture_radius = 0.0f;
			} else {
				scene.camera.aperture_radius = node->get_child_value_optional("apertureRadius", 0.05f);
				scene.camera.focal_distance  = node->get_child_value_optional("focusDistance", 10.0f);
			}
			parse_transform(node, &scene.camera.position, &scene.camera.rotation, nullptr, Vector3(0.0f, 0.0f, -1.0f));
		} else {
			warn(*node, "camera type '" + std::string(camera_type) + "' not supported");
		}
		if (const XMLNode * film = node->get_child_by_tag("film")) {
			cpu_config.initial_width  = film->get_child_value_optional("width",  cpu_config.initial_width);
			cpu_config.initial_height = film->get_child_value_optional("height", cpu_config.initial_height);
			scene.camera.resize(cpu_config.initial_width, cpu_config.initial_height);
		}
	} else if (node->tag == "integrator") {
		gpu_config.num_bounces = node->get_child_value_optional("maxDepth", gpu_config.num_bounces);
	} else if (node->tag == "emitter") {
		std::string_view emitter_type = node->get_attribute_value("type");
		if (emitter_type == "area") {
			if (const XMLAttribute * id = node->get_attribute("id")) {
				Material material;
				material.type = Material::Type::LIGHT;
				material.name = id->value;
				material.emission = node->require_child_by_name("radiance").require_attribute("value").as_vector3();
				state.materials[id->value] = scene.asset_manager.add_material(std::move(material));
			} else {
				warn(*node, "emitter defined without an id that is also not attached to any geometry");
			}
		} else if (emitter_type == "envmap") {
			std::string filename(node->require_child_by_name("filename").get_attribute_value("value"));
			size_t dot = filename.find_last_of('.');
			if (dot == std::string::npos)            warn(*node, "environment map '" + filename + "' has no file extension");
			else if (filename.substr(dot + 1) != "hdr") warn(*node, "only HDR environment maps are supported");
			else cpu_config.sky_filename = join_path(state.directory, filename);
		} else if (emitter_type == "point") {
			constexpr float RADIUS = 0.0001f; // a point light becomes a tiny emissive icosahedron
			Matrix4 transform = parse_transform_matrix(node) * Matrix4::create_scale(RADIUS);
			Handle<MeshData> mesh_data_handle = scene.asset_manager.add_mesh_data(Geometry::sphere(transform, 0));
			Material material;
			material.type = Material::Type::LIGHT;
			material.emission = node->get_child_value_optional("intensity", Vector3(1.0f));
			scene.add_mesh("PointLight", mesh_data_handle, scene.asset_manager.add_material(std::move(material)));
		} else {
			warn(*node, "emitter type '" + std::string(emitter_type) + "' is not supported");
		}
	} else if (node->tag == "include") {
		MitsubaLoader::load(join_path(state.directory, node->get_attribute_value("filename")), scene);
	} else {
		for (const XMLNode & child : node->children) walk(&child, scene, state);
	}
}

} // namespace

void MitsubaLoader::load(const std::string & filename, Scene & scene) {
	XMLParser xml_parser(filename);
	XMLNode root = xml_parser.parse_root();

	const XMLNode * scene_node = root.get_child_by_tag("scene");
	if (!scene_node) throw ParseError(filename + ": file does not contain a <scene> tag");

	{
		std::string version(scene_node->get_attribute_value("version"));
		Parser v(version);
		int major = v.parse_int(); v.expect('.');
		int minor = v.parse_int(); v.expect('.');
		int patch = v.parse_int();
		if (major * 100 + minor * 10 + patch >= 200) throw ParseError(filename + ": Mitsuba 2 files are not supported");
	}

	LoadState state;
	state.directory = directory_of(filename);
	walk(scene_node, scene, state);
}
