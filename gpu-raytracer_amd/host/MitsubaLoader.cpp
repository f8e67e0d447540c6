// Mitsuba 0.5 scene files -> Scene.
//
// What is accepted, and every default, is the behaviour of the reference's loader
// (Src/Assets/Mitsuba/MitsubaLoader.cpp) -- tests/test_scene_load.py compares whole scenes, float by float,
// with that loader compiled verbatim. How it is organised is this project's: the file format is described by
// TABLES (which element does what, which BSDF plug-in maps to which material model and where its parameters
// live, which transform operation builds which matrix, which shape plug-in produces geometry how), and one
// small interpreter, SceneFile, walks the document and looks things up. Adding a plug-in is adding a row.
//
//   elements    bsdf, texture, shape, sensor, integrator, emitter, include; anything else is a container
//   bsdf        diffuse, difftrans, plastic, roughplastic, roughdiffuse, phong, conductor, roughconductor,
//               dielectric, thindielectric, roughdielectric; twosided / mask / bumpmap / coating are wrappers
//   shapes      obj, ply, serialized, hair (file meshes); rectangle, cube, disk, cylinder, sphere (generated, the
//               transform is baked into the vertices); shapegroup + instance
//   emitters    area (named, or attached to a shape), point, envmap
//   sensors     perspective, perspective_rdist, thinlens (+ film size); integrator maxDepth
// The scene file overrides the command line: film width / height and maxDepth are written into cpu_config /
// gpu_config (reference: MitsubaLoader.cpp:610-616).
#include "Scene.h"
#include "XMLParser.h"

#include <cstdio>
#include <map>

namespace {

// ---- small helpers -------------------------------------------------------------------------------------

void complain(const XMLNode & where, const std::string & what) {
	fprintf(stderr, "%s: WARNING: %s\n", where.location.c_str(), what.c_str());
}

struct PathName {
	static size_t last_separator(const std::string & path) { return path.find_last_of("/\\"); }
	static std::string folder(const std::string & path) { // with trailing separator
		size_t at = last_separator(path);
		return at == std::string::npos ? std::string("./") : path.substr(0, at + 1);
	}
	static std::string leaf(const std::string & path) {
		size_t at = last_separator(path);
		return at == std::string::npos ? path : path.substr(at + 1);
	}
	// scene files written on Windows say "textures\\name.tga" (Data/Sponza/scene.xml:13): both separators, single or doubled
	static std::string below(const std::string & folder, std::string_view relative) {
		std::string out = folder;
		for (size_t i = 0; i < relative.size(); i++) {
			char c = relative[i];
			if (c != '\\') { out += c; continue; }
			out += '/';
			if (i + 1 < relative.size() && relative[i + 1] == '\\') i++;
		}
		return out;
	}
};

// `value` of the child named `name`, or the fallback
template<typename T> T property(const XMLNode & node, const char * name, T fallback) { return node.get_child_value_optional(name, fallback); }
std::string_view plugin(const XMLNode & node) { return node.get_attribute_value("type"); }
std::string filename_property(const XMLNode & node) { return std::string(node.require_child_by_name("filename").get_attribute_value("value")); }

// ---- transforms: each operation of a <transform> multiplies from the left, in document order ---------------

struct TransformOp { const char * tag; bool (*build)(const XMLNode & op, Matrix4 & out); };

const TransformOp TRANSFORM_OPS[] = {
	{ "matrix", [](const XMLNode & op, Matrix4 & out) { out = op.require_attribute("value").as_matrix4(); return true; } },
	{ "lookat", [](const XMLNode & op, Matrix4 & out) {
		Vector3 eye = op.get_attribute_optional("origin", Vector3(0.0f, 0.0f,  0.0f));
		Vector3 at  = op.get_attribute_optional("target", Vector3(0.0f, 0.0f, -1.0f));
		Vector3 up  = op.get_attribute_optional("up",     Vector3(0.0f, 1.0f,  0.0f));
		out = Matrix4::create_translation(eye) * Matrix4::create_rotation(Quaternion::look_rotation(at - eye, up));
		return true;
	} },
	{ "scale", [](const XMLNode & op, Matrix4 & out) {
		const XMLAttribute * uniform = op.get_attribute("value");
		out = uniform ? Matrix4::create_scale(uniform->as_float())
		              : Matrix4::create_scale(op.get_attribute_optional("x", 1.0f), op.get_attribute_optional("y", 1.0f), op.get_attribute_optional("z", 1.0f));
		return true;
	} },
	{ "rotate", [](const XMLNode & op, Matrix4 & out) {
		Vector3 axis(op.get_attribute_optional("x", 0.0f), op.get_attribute_optional("y", 0.0f), op.get_attribute_optional("z", 0.0f));
		if (axis.x == 0.0f && axis.y == 0.0f && axis.z == 0.0f) { complain(op, "rotation without axis specified"); return false; }
		out = Matrix4::create_rotation(Quaternion::axis_angle(axis, Math::deg_to_rad(op.get_attribute_optional("angle", 0.0f))));
		return true;
	} },
	{ "translate", [](const XMLNode & op, Matrix4 & out) {
		out = Matrix4::create_translation(Vector3(op.get_attribute_optional("x", 0.0f), op.get_attribute_optional("y", 0.0f), op.get_attribute_optional("z", 0.0f)));
		return true;
	} },
};

Matrix4 to_world(const XMLNode & owner) {
	Matrix4 world;
	const XMLNode * list = owner.get_child_by_tag("transform");
	if (!list) return world;
	for (const XMLNode & op : list->children) {
		const TransformOp * known = nullptr;
		for (const TransformOp & candidate : TRANSFORM_OPS) if (op.tag == candidate.tag) { known = &candidate; break; }
		if (!known) { complain(op, "node <" + op.tag + "> is not a valid transformation"); continue; }
		Matrix4 step;
		if (known->build(op, step)) world = step * world;
	}
	return world;
}

void place(const XMLNode & owner, Vector3 * position, Quaternion * rotation, float * scale, const Vector3 & forward = Vector3(0.0f, 0.0f, 1.0f)) {
	Matrix4::decompose(to_world(owner), position, rotation, scale, forward);
}

// ---- BSDF plug-ins -----------------------------------------------------------------------------------------

enum class Roughness { UNUSED, SMOOTH, ALPHA, PHONG_EXPONENT };   // where linear_roughness comes from (UNUSED: the model has none, the field keeps its default)
enum class Optics    { NONE, CONDUCTOR, DIELECTRIC };         // which extra parameters the model reads

struct BsdfModel {
	const char *   plugin;
	Material::Type material;
	const char *   colour;       // name of the reflectance-like parameter (rgb / srgb / texture / ref), or null
	Roughness      roughness;
	Optics         optics;
};

const BsdfModel BSDF_MODELS[] = {
	{ "diffuse",         Material::Type::DIFFUSE,    "reflectance",        Roughness::UNUSED,         Optics::NONE       },
	{ "difftrans",       Material::Type::DIFFUSE,    "transmittance",      Roughness::UNUSED,         Optics::NONE       },
	{ "plastic",         Material::Type::PLASTIC,    "diffuseReflectance", Roughness::SMOOTH,         Optics::NONE       },
	{ "roughplastic",    Material::Type::PLASTIC,    "diffuseReflectance", Roughness::ALPHA,          Optics::NONE       },
	{ "roughdiffuse",    Material::Type::PLASTIC,    "diffuseReflectance", Roughness::ALPHA,          Optics::NONE       },
	{ "phong",           Material::Type::PLASTIC,    "diffuseReflectance", Roughness::PHONG_EXPONENT, Optics::NONE       },
	{ "conductor",       Material::Type::CONDUCTOR,  nullptr,              Roughness::SMOOTH,         Optics::CONDUCTOR  },
	{ "roughconductor",  Material::Type::CONDUCTOR,  nullptr,              Roughness::ALPHA,          Optics::CONDUCTOR  },
	{ "dielectric",      Material::Type::DIELECTRIC, nullptr,              Roughness::SMOOTH,         Optics::DIELECTRIC },
	{ "thindielectric",  Material::Type::DIELECTRIC, nullptr,              Roughness::SMOOTH,         Optics::DIELECTRIC },
	{ "roughdielectric", Material::Type::DIELECTRIC, nullptr,              Roughness::ALPHA,          Optics::DIELECTRIC },
};
const char * const BSDF_WRAPPERS[] = { "twosided", "mask", "bumpmap", "coating" };   // only what they wrap matters

// named indices of refraction (Mitsuba 0.5 documentation, page 58)
const struct { const char * medium; float index; } NAMED_IOR[] = {
	{ "vacuum", 1.0f }, { "helium", 1.00004f }, { "hydrogen", 1.00013f }, { "air", 1.00028f }, { "carbon dioxide", 1.00045f },
	{ "water", 1.3330f }, { "acetone", 1.36f }, { "ethanol", 1.361f }, { "carbon tetrachloride", 1.461f }, { "glycerol", 1.4729f },
	{ "benzene", 1.501f }, { "silicone oil", 1.52045f }, { "bromine", 1.661f }, { "water ice", 1.31f }, { "fused quartz", 1.458f },
	{ "pyrex", 1.470f }, { "acrylic glass", 1.49f }, { "polypropylene", 1.49f }, { "bk7", 1.5046f }, { "sodium chloride", 1.544f },
	{ "amber", 1.55f }, { "pet", 1.575f }, { "diamond", 2.419f },
};

float index_of_refraction(const XMLNode & bsdf, const char * name, float fallback) {
	const XMLNode * given = bsdf.get_child_by_name(name);
	if (!given || given->tag != "string") return property(bsdf, name, fallback);
	std::string_view medium = given->get_attribute_value("value");
	for (const auto & row : NAMED_IOR) if (medium == row.medium) return row.index;
	throw ParseError(given->location + ": index of refraction not known for '" + std::string(medium) + "'");
}

// ---- shape plug-ins ------------------------------------------------------------------------------------------

struct SceneFile;
struct ShapeKind {
	const char * plugin;
	bool         transform_is_baked;   // generated geometry: the <transform> goes into the vertices, the Mesh keeps the identity
	Handle<MeshData> (*make)(SceneFile & file, const XMLNode & shape, std::string & mesh_name);
};

// ---- the interpreter -------------------------------------------------------------------------------------------

struct SceneFile {
	Scene &     scene;
	std::string folder;    // of the scene file, with trailing separator

	struct Prototype { Handle<MeshData> geometry; Handle<Material> material; };   // a <shape type="shapegroup">
	std::map<std::string, Prototype>        prototypes;
	std::map<std::string, Handle<Material>> materials;     // by material NAME (the bsdf's id, or "Material")
	std::map<std::string, Handle<Texture>>  textures;      // by texture id

	SceneFile(Scene & scene, const std::string & filename) : scene(scene), folder(PathName::folder(filename)) { }

	// -- textures and colours
	Handle<Texture> texture(const XMLNode * node, Vector3 & tint) {
		std::string_view kind = plugin(*node);
		if (kind == "scale") { // a scale node multiplies the colour and wraps the real texture
			if (const XMLNode * factor = node->get_child_by_name("scale")) {
				if      (factor->tag == "float") tint *= factor->require_attribute("value").as_float();
				else if (factor->tag == "rgb")   tint *= factor->require_attribute("value").as_vector3();
				else complain(*factor, "invalid scale tag <" + factor->tag + ">");
			}
			node = node->get_child_by_tag("texture");
			if (!node) return Handle<Texture> { INVALID };
			kind = plugin(*node);
		}
		if (kind != "bitmap") { complain(*node, "only bitmap textures are supported"); return Handle<Texture> { INVALID }; }
		std::string path = PathName::below(folder, node->require_child_by_name("filename").get_attribute_value("value"));
		Handle<Texture> handle = scene.asset_manager.add_texture(path, PathName::leaf(path));
		if (const XMLAttribute * id = node->get_attribute("id")) textures[id->value] = handle;
		return handle;
	}

	void colour(const XMLNode & bsdf, const char * name, Vector3 & rgb, Handle<Texture> & map) {
		const XMLNode * given = bsdf.get_child_by_name(name);
		if (!given) { rgb = Vector3(1.0f); return; }
		const std::string & form = given->tag;
		if (form == "rgb" || form == "srgb") {
			rgb = given->get_attribute_optional("value", Vector3(1.0f));
			if (form == "srgb") rgb = Vector3(Math::gamma_to_linear(rgb.x), Math::gamma_to_linear(rgb.y), Math::gamma_to_linear(rgb.z));
		} else if (form == "texture") {
			map = texture(given, rgb);
			if (const XMLNode * factor = given->get_child_by_name("scale")) rgb = factor->get_attribute_optional("value", Vector3(1.0f));
		} else if (form == "ref") {
			std::string id(given->get_attribute_value("id"));
			auto known = textures.find(id);
			if (known != textures.end()) map = known->second;
			else complain(*given, "invalid texture ref '" + id + "'");
		}
	}

	// -- materials
	Handle<Material> emitter_material(const XMLNode & emitter, const std::string & name) {
		Material light;
		light.type = Material::Type::LIGHT;
		light.name = name;
		light.emission = emitter.require_child_by_name("radiance").require_attribute("value").as_vector3();
		return scene.asset_manager.add_material(std::move(light));
	}

	// `ref` names a material declared earlier; an unknown id falls back to the default material
	Handle<Material> referenced_material(const XMLNode & ref) {
		std::string id(ref.get_attribute_value("id"));
		auto known = materials.find(id);
		if (known != materials.end()) return known->second;
		complain(ref, "invalid material ref '" + id + "'");
		return Handle<Material>::get_default();
	}

	// The material of `owner`: a <bsdf> itself, or a shape (whose <emitter> wins over a <ref>, which wins over an inline <bsdf>)
	Handle<Material> material(const XMLNode & owner) {
		const XMLNode * bsdf = &owner;
		if (owner.tag != "bsdf") {
			if (const XMLNode * emitter = owner.get_child_by_tag("emitter")) return emitter_material(*emitter, "emitter");
			if (const XMLNode * ref = owner.get_child_by_tag("ref")) return referenced_material(*ref);
			bsdf = owner.get_child_by_tag("bsdf");
			if (!bsdf) { complain(owner, "unable to parse BSDF"); return Handle<Material>::get_default(); }
		}

		// peel the wrappers; the name is the outermost id there is
		const XMLAttribute * id = bsdf->get_attribute("id");
		auto is_wrapper = [](std::string_view kind) { for (const char * w : BSDF_WRAPPERS) if (kind == w) return true; return false; };
		while (is_wrapper(plugin(*bsdf))) {
			const XMLNode * wrapped = bsdf->get_child_by_tag("bsdf");
			if (!wrapped) {
				const XMLNode * ref = bsdf->get_child_by_tag("ref");
				return ref ? referenced_material(*ref) : Handle<Material>::get_default();
			}
			bsdf = wrapped;
			if (!id) id = bsdf->get_attribute("id");
		}

		const std::string_view kind = plugin(*bsdf);
		const BsdfModel * model = nullptr;
		for (const BsdfModel & candidate : BSDF_MODELS) if (kind == candidate.plugin) { model = &candidate; break; }
		if (!model) { complain(*bsdf, "BSDF type '" + std::string(kind) + "' not supported"); return Handle<Material>::get_default(); }

		Material made;
		made.name = id ? id->value : "Material";
		made.type = model->material;
		if (model->colour) colour(*bsdf, model->colour, made.diffuse, made.texture_handle);
		if (model->optics == Optics::DIELECTRIC) { // (before the roughness: an unknown named index of refraction ends the load)
			float inside = index_of_refraction(*bsdf, "intIOR", 1.33f), outside = index_of_refraction(*bsdf, "extIOR", 1.0f);
			made.index_of_refraction = outside == 0.0f ? inside : inside / outside;
		}
		switch (model->roughness) {
			case Roughness::UNUSED:         break;
			case Roughness::SMOOTH:         made.linear_roughness = 0.0f; break;
			case Roughness::ALPHA:          made.linear_roughness = property(*bsdf, "alpha", 0.5f); break;
			case Roughness::PHONG_EXPONENT: made.linear_roughness = powf(0.5f * property(*bsdf, "exponent", 1.0f) + 1.0f, 0.25f); break;
		}
		if (model->optics == Optics::CONDUCTOR) {
			const XMLNode * preset = bsdf->get_child_by_name("material");
			bool mirror = preset && preset->get_attribute_value("value") == "none";
			made.eta = mirror ? Vector3(0.0f) : property(*bsdf, "eta", Vector3(1.33f));
			made.k   = mirror ? Vector3(1.0f) : property(*bsdf, "k",   Vector3(1.0f));
		}
		return scene.asset_manager.add_material(std::move(made));
	}

	// -- media
	Handle<Medium> interior_medium(const XMLNode & shape) {
		const XMLNode * given = shape.get_child_by_tag("medium");
		if (!given) return Handle<Medium> { INVALID };
		if (plugin(*given) != "homogeneous") {
			complain(*given, "medium type '" + std::string(plugin(*given)) + "' not supported");
			return Handle<Medium> { INVALID };
		}
		Medium medium;
		if (const XMLAttribute * name = given->get_attribute("name")) medium.name = name->value;

		// coefficients: (sigmaA, sigmaS) or (sigmaT, albedo), exactly one of the pairs
		auto coefficient = [&](const char * name) { return given->get_child_by_name(name); };
		const XMLNode * a = coefficient("sigmaA"), * s = coefficient("sigmaS"), * t = coefficient("sigmaT"), * albedo = coefficient("albedo");
		Vector3 absorb, scatter;
		const bool as_pair = a && s, t_pair = t && albedo;
		if (as_pair == t_pair) {
			complain(*given, "provide EITHER sigmaA and sigmaS OR sigmaT and albedo");
		} else if (as_pair) {
			absorb  = a->require_attribute("value").as_vector3();
			scatter = s->require_attribute("value").as_vector3();
		} else {
			Vector3 extinction = t->require_attribute("value").as_vector3();
			scatter = albedo->require_attribute("value").as_vector3() * extinction;
			absorb  = extinction - scatter;
		}
		const float density = property(*given, "scale", 1.0f);
		medium.from_sigmas(density * absorb, density * scatter);   // (with g = 0: the phase function is read afterwards, as in the reference)

		if (const XMLNode * phase = given->get_child_by_tag("phase")) {
			std::string_view kind = plugin(*phase);
			if      (kind == "isotropic") medium.g = 0.0f;
			else if (kind == "hg")        medium.g = property(*phase, "g", 0.0f);
			else complain(*given, "phase function type '" + std::string(kind) + "' not supported");
		}
		return scene.asset_manager.add_medium(std::move(medium));
	}

	// -- shapes
	static const ShapeKind * shape_kind(std::string_view kind);

	Handle<MeshData> geometry(const XMLNode & shape, std::string & mesh_name) {
		const ShapeKind * kind = shape_kind(plugin(shape));
		if (!kind) { complain(shape, "shape type '" + std::string(plugin(shape)) + "' not supported"); return Handle<MeshData> { INVALID }; }
		return kind->make(*this, shape, mesh_name);
	}

	void shape(const XMLNode & node) {
		const std::string_view kind = plugin(node);
		if (kind == "shapegroup") { // a prototype: geometry + material under an id, placed by <shape type="instance">
			if (node.children.empty()) return;
			const XMLNode * inner = node.get_child_by_tag("shape");
			if (!inner) throw ParseError(node.location + ": shapegroup needs a <shape> child");
			std::string unused;
			Prototype prototype;
			prototype.geometry = geometry(*inner, unused);
			prototype.material = material(*inner);
			prototypes[std::string(node.get_attribute_value("id"))] = prototype;
			return;
		}
		if (kind == "instance") {
			const XMLNode * ref = node.get_child_by_tag("ref");
			if (!ref) { complain(node, "instance without ref"); return; }
			std::string id(ref->get_attribute_value("id"));
			auto known = prototypes.find(id);
			if (known == prototypes.end() || known->second.geometry.handle == INVALID) return;
			Mesh & mesh = scene.add_mesh(id, known->second.geometry, known->second.material);
			place(node, &mesh.position, &mesh.rotation, &mesh.scale);
			return;
		}

		std::string mesh_name;
		Handle<MeshData> mesh_data = geometry(node, mesh_name);
		Handle<Material> surface   = material(node);
		Handle<Medium>   inside    = interior_medium(node);

		if (surface.handle != INVALID) { // the medium hangs on the material; a material already bound to another medium is cloned for this shape
			Material & bound = scene.asset_manager.get_material(surface);
			if (bound.medium_handle.handle != INVALID && bound.medium_handle.handle != inside.handle) {
				Material clone = bound;
				clone.medium_handle = inside;
				surface = scene.asset_manager.add_material(std::move(clone));
			} else {
				bound.medium_handle = inside;
			}
		}
		if (mesh_data.handle == INVALID) return;
		const ShapeKind * made_by = shape_kind(kind);
		Mesh & mesh = scene.add_mesh(std::move(mesh_name), mesh_data, surface);
		if (!made_by->transform_is_baked) place(node, &mesh.position, &mesh.rotation, &mesh.scale);
	}

	// -- the other top-level elements
	void declare_bsdf(const XMLNode & node) {
		Handle<Material> handle = material(node);
		materials[scene.asset_manager.get_material(handle).name] = handle;
	}
	void declare_texture(const XMLNode & node) { Vector3 unused = 1.0f; texture(&node, unused); }

	void sensor(const XMLNode & node) {
		const std::string_view kind = plugin(node);
		const bool pinhole = kind == "perspective", lens = kind == "perspective_rdist" || kind == "thinlens";
		if (pinhole || lens) {
			if (const XMLNode * fov = node.get_child_by_name("fov")) scene.camera.set_fov(Math::deg_to_rad(fov->require_attribute("value").as_float()));
			scene.camera.aperture_radius = lens ? property(node, "apertureRadius", 0.05f) : 0.0f;
			if (lens) scene.camera.focal_distance = property(node, "focusDistance", 10.0f);
			place(node, &scene.camera.position, &scene.camera.rotation, nullptr, Vector3(0.0f, 0.0f, -1.0f));
		} else {
			complain(node, "camera type '" + std::string(kind) + "' not supported");
		}
		if (const XMLNode * film = node.get_child_by_tag("film")) {
			cpu_config.initial_width  = property(*film, "width",  cpu_config.initial_width);
			cpu_config.initial_height = property(*film, "height", cpu_config.initial_height);
			scene.camera.resize(cpu_config.initial_width, cpu_config.initial_height);
		}
	}

	void integrator(const XMLNode & node) { gpu_config.num_bounces = property(node, "maxDepth", gpu_config.num_bounces); }

	void emitter(const XMLNode & node) {
		const std::string_view kind = plugin(node);
		if (kind == "area") { // a named emitter is a light material that shapes can <ref>
			const XMLAttribute * id = node.get_attribute("id");
			if (id) materials[id->value] = emitter_material(node, id->value);
			else complain(node, "emitter defined without an id that is also not attached to any geometry");
		} else if (kind == "envmap") {
			std::string file = filename_property(node);
			size_t dot = file.find_last_of('.');
			if (dot == std::string::npos)            complain(node, "environment map '" + file + "' has no file extension");
			else if (file.substr(dot + 1) != "hdr") complain(node, "only HDR environment maps are supported");
			else cpu_config.sky_filename = PathName::below(folder, file);
		} else if (kind == "point") { // a point light becomes a tiny emissive icosahedron
			constexpr float RADIUS = 0.0001f;
			Handle<MeshData> ball = scene.asset_manager.add_mesh_data(Geometry::sphere(to_world(node) * Matrix4::create_scale(RADIUS), 0));
			Material light;
			light.type = Material::Type::LIGHT;
			light.emission = property(node, "intensity", Vector3(1.0f));
			scene.add_mesh("PointLight", ball, scene.asset_manager.add_material(std::move(light)));
		} else {
			complain(node, "emitter type '" + std::string(kind) + "' is not supported");
		}
	}

	void include(const XMLNode & node) { MitsubaLoader::load(PathName::below(folder, node.get_attribute_value("filename")), scene); }

	// -- the walk: an element with a handler is handled, anything else is a container of elements
	void element(const XMLNode & node);
};

struct ElementHandler { const char * tag; void (SceneFile::*handle)(const XMLNode &); };
const ElementHandler ELEMENTS[] = {
	{ "bsdf",       &SceneFile::declare_bsdf    },
	{ "texture",    &SceneFile::declare_texture },
	{ "shape",      &SceneFile::shape           },
	{ "sensor",     &SceneFile::sensor          },
	{ "integrator", &SceneFile::integrator      },
	{ "emitter",    &SceneFile::emitter         },
	{ "include",    &SceneFile::include         },
};

void SceneFile::element(const XMLNode & node) {
	for (const ElementHandler & known : ELEMENTS) if (node.tag == known.tag) { (this->*known.handle)(node); return; }
	for (const XMLNode & child : node.children) element(child);
}

// file meshes: loaded (and cached) by the asset manager under their path
template<std::vector<Triangle> (*LOAD)(const std::string &)>
Handle<MeshData> file_mesh(SceneFile & file, const XMLNode & shape, std::string & mesh_name) { // reference: MitsubaLoader.cpp:434-442
	std::string path = PathName::below(file.folder, filename_property(shape));
	mesh_name = PathName::leaf(path);
	return file.scene.asset_manager.add_mesh_data(path, LOAD);
}
// generated meshes: `TESSELLATE(shape, transform)` with the transform baked in
template<std::vector<Triangle> (*TESSELLATE)(const XMLNode &, const Matrix4 &)>
Handle<MeshData> generated_mesh(SceneFile & file, const XMLNode & shape, std::string & mesh_name) {
	mesh_name = std::string(plugin(shape));
	return file.scene.asset_manager.add_mesh_data(TESSELLATE(shape, to_world(shape)));
}
std::vector<Triangle> rectangle_of(const XMLNode &, const Matrix4 & m) { return Geometry::rectangle(m); }
std::vector<Triangle> cube_of     (const XMLNode &, const Matrix4 & m) { return Geometry::cube(m); }
std::vector<Triangle> disk_of     (const XMLNode &, const Matrix4 & m) { return Geometry::disk(m); }
std::vector<Triangle> cylinder_of (const XMLNode & shape, const Matrix4 & m) {
	return Geometry::cylinder(m, property(shape, "p0", Vector3(0.0f, 0.0f, 0.0f)), property(shape, "p1", Vector3(0.0f, 0.0f, 1.0f)), property(shape, "radius", 1.0f));
}
std::vector<Triangle> sphere_of(const XMLNode & shape, const Matrix4 & m) {
	const float radius = property(shape, "radius", 1.0f);
	Vector3 centre(0.0f);
	if (const XMLNode * c = shape.get_child_by_name("center")) centre = Vector3(c->get_attribute_optional("x", 0.0f), c->get_attribute_optional("y", 0.0f), c->get_attribute_optional("z", 0.0f));
	return Geometry::sphere(m * Matrix4::create_translation(centre) * Matrix4::create_scale(radius));
}
// one archive holds many meshes: the reference keys (and caches) each by "<archive>.shape_<i>.bvh" (MitsubaLoader.cpp:487-500)
Handle<MeshData> serialized_mesh(SceneFile & file, const XMLNode & shape, std::string & mesh_name) {
	std::string relative = filename_property(shape), archive = PathName::below(file.folder, relative);
	const int index = property(shape, "shapeIndex", 0);
	mesh_name = relative + "_" + std::to_string(index);
	std::string key = archive + ".shape_" + std::to_string(index) + ".bvh";
	return file.scene.asset_manager.add_mesh_data(key, key, [archive, index](const std::string &) { return SerializedLoader::load(archive, index); });
}
Handle<MeshData> hair_mesh(SceneFile & file, const XMLNode & shape, std::string & mesh_name) { // reference: MitsubaLoader.cpp:501-512
	mesh_name = filename_property(shape);
	const float radius = property(shape, "radius", 0.0025f);
	return file.scene.asset_manager.add_mesh_data(PathName::below(file.folder, mesh_name), [radius](const std::string & path) { return MitshairLoader::load(path, radius); });
}

const ShapeKind SHAPE_KINDS[] = {
	{ "obj",        false, file_mesh<OBJLoader::load>   },
	{ "ply",        false, file_mesh<PLYLoader::load>   },
	{ "rectangle",  true,  generated_mesh<rectangle_of> },
	{ "cube",       true,  generated_mesh<cube_of>      },
	{ "disk",       true,  generated_mesh<disk_of>      },
	{ "cylinder",   true,  generated_mesh<cylinder_of>  },
	{ "sphere",     true,  generated_mesh<sphere_of>    },
	{ "serialized", false, serialized_mesh              },
	{ "hair",       false, hair_mesh                    },
};
const ShapeKind * SceneFile::shape_kind(std::string_view kind) {
	for (const ShapeKind & candidate : SHAPE_KINDS) if (kind == candidate.plugin) return &candidate;
	return nullptr;
}

} // namespace

void MitsubaLoader::load(const std::string & filename, Scene & scene) {
	XMLParser reader(filename);
	XMLNode document = reader.parse_root();
	const XMLNode * root = document.get_child_by_tag("scene");
	if (!root) throw ParseError(filename + ": file does not contain a <scene> tag");

	// version "a.b.c": anything from 2.0.0 on is the other file format
	std::string version(root->get_attribute_value("version"));
	Parser digits(version);
	int major = digits.parse_int(); digits.expect('.');
	int minor = digits.parse_int(); digits.expect('.');
	int patch = digits.parse_int();
	if (major * 100 + minor * 10 + patch >= 200) throw ParseError(filename + ": Mitsuba 2 files are not supported");

	SceneFile(scene, filename).element(*root);
}
