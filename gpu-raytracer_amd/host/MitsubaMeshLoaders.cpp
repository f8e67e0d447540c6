// The two Mitsuba-specific mesh containers a scene file can name besides obj / ply:
//   serialized  -- Mitsuba's binary mesh archive: per-mesh zlib streams + an end-of-file dictionary
//                  (reference: Src/Assets/Mitsuba/SerializedLoader.cpp:9-221)
//   hair        -- strands as ascii or "BINARY_HAIR" polylines, turned into flat ribbons that taper
//                  to a point (reference: Src/Assets/Mitsuba/MitshairLoader.cpp:10-123)
#include "Scene.h"
#include "XMLParser.h"

#include <cstdint>
#include <cstring>
#include <zlib.h>

namespace {

// Bounds-checked little-endian reads from a byte buffer.
struct ByteReader {
	const unsigned char * data;
	size_t size;
	size_t pos = 0;
	const std::string & what;

	template<typename T> T read() {
		if (pos + sizeof(T) > size) throw ParseError("'" + what + "': unexpected end of data");
		T value;
		memcpy(&value, data + pos, sizeof(T));
		pos += sizeof(T);
		return value;
	}
	const unsigned char * span(size_t bytes) {
		if (bytes > size - pos) throw ParseError("'" + what + "': unexpected end of data");
		const unsigned char * p = data + pos;
		pos += bytes;
		return p;
	}
};

std::vector<unsigned char> inflate_zlib(const unsigned char * src, size_t src_bytes, const std::string & what) {
	z_stream z;
	memset(&z, 0, sizeof(z));
	if (inflateInit(&z) != Z_OK) throw ParseError("'" + what + "': zlib initialisation failed");
	std::vector<unsigned char> out(src_bytes * 3 + 64);
	z.next_in  = const_cast<Bytef *>(src);
	z.avail_in = uInt(src_bytes);
	int status = Z_OK;
	while (status != Z_STREAM_END) {
		if (z.total_out == out.size()) out.resize(out.size() * 2);
		z.next_out  = out.data() + z.total_out;
		z.avail_out = uInt(std::min<size_t>(out.size() - z.total_out, 1u << 30));
		status = inflate(&z, Z_NO_FLUSH);
		if (status != Z_OK && status != Z_STREAM_END) {
			inflateEnd(&z);
			throw ParseError("'" + what + "': failed to decompress the mesh stream");
		}
	}
	out.resize(z.total_out);
	inflateEnd(&z);
	return out;
}

// Element i of a packed float or double array, as float
struct RealArray {
	const unsigned char * base;
	bool doubles;
	float operator[](uint64_t i) const {
		if (doubles) { double v; memcpy(&v, base + i * 8, 8); return float(v); }
		float v; memcpy(&v, base + i * 4, 4); return v;
	}
	Vector3 vector3(uint64_t i) const { return Vector3((*this)[3 * i], (*this)[3 * i + 1], (*this)[3 * i + 2]); }
	Vector2 vector2(uint64_t i) const { return Vector2((*this)[2 * i], (*this)[2 * i + 1]); }
};

} // namespace

std::vector<Triangle> SerializedLoader::load(const std::string & filename, int shape_index) {
	std::string file = read_text_file(filename);
	ByteReader archive { (const unsigned char *)file.data(), file.size(), 0, filename };

	if (archive.read<uint16_t>() != 0x041c) throw ParseError("serialized file '" + filename + "' does not start with format ID 0x041c");
	uint16_t version = archive.read<uint16_t>();

	// End-of-file dictionary: ..., offset[0 .. n), uint32 n. 32-bit offsets up to version 3, 64-bit after.
	if (file.size() < 8) throw ParseError("'" + filename + "': too short for a serialized file");
	archive.pos = file.size() - 4;
	uint32_t mesh_count = archive.read<uint32_t>();
	size_t   offset_size = version <= 3 ? 4 : 8;
	if (size_t(mesh_count) * offset_size + 4 > file.size()) throw ParseError("'" + filename + "': corrupt end-of-file dictionary");
	size_t dictionary = file.size() - 4 - size_t(mesh_count) * offset_size;
	if (shape_index < 0 || uint32_t(shape_index) >= mesh_count) throw ParseError("'" + filename + "' has no shape #" + std::to_string(shape_index));

	std::vector<uint64_t> offsets(mesh_count + 1);
	archive.pos = dictionary;
	for (uint32_t i = 0; i < mesh_count; i++) offsets[i] = version <= 3 ? uint64_t(archive.read<uint32_t>()) : archive.read<uint64_t>();
	offsets[mesh_count] = dictionary;

	uint64_t begin = offsets[shape_index], end = offsets[shape_index + 1];
	if (begin + 4 > end || end > dictionary) throw ParseError("'" + filename + "': corrupt mesh offsets");
	// each mesh repeats the 4-byte format / version header before its zlib stream
	std::vector<unsigned char> mesh_bytes = inflate_zlib((const unsigned char *)file.data() + begin + 4, size_t(end - begin - 4), filename);
	ByteReader mesh { mesh_bytes.data(), mesh_bytes.size(), 0, filename };

	uint32_t flags = mesh.read<uint32_t>();
	bool has_normals      = flags & 0x0001;
	bool has_tex_coords   = flags & 0x0002;
	bool has_colours      = flags & 0x0008;
	bool use_face_normals = flags & 0x0010;
	bool single_precision = flags & 0x1000;
	bool double_precision = flags & 0x2000;
	if (version <= 3) {
		single_precision = true;
	} else {
		while (mesh.read<char>() != '\0') { } // the mesh's name
	}
	uint64_t vertex_count   = mesh.read<uint64_t>();
	uint64_t triangle_count = mesh.read<uint64_t>();
	if (vertex_count == 0 || triangle_count == 0) {
		fprintf(stderr, "WARNING: serialized mesh '%s' #%d defined without vertices or triangles!\n", filename.c_str(), shape_index);
		return { };
	}
	if (!single_precision && !double_precision) throw ParseError("'" + filename + "': neither single nor double precision specified");
	if (vertex_count > mesh_bytes.size() || triangle_count > mesh_bytes.size()) throw ParseError("'" + filename + "': corrupt mesh header");

	bool   doubles = !single_precision;
	size_t real    = doubles ? 8 : 4;
	RealArray positions  { mesh.span(vertex_count * 3 * real), doubles };
	RealArray normals    { has_normals    ? mesh.span(vertex_count * 3 * real) : nullptr, doubles };
	RealArray tex_coords { has_tex_coords ? mesh.span(vertex_count * 2 * real) : nullptr, doubles };
	if (has_colours) mesh.span(vertex_count * 3 * real);

	bool wide_indices = vertex_count > 0xffffffffull;
	const unsigned char * indices = mesh.span(triangle_count * 3 * (wide_indices ? 8 : 4));
	auto index_at = [&](uint64_t i) -> uint64_t {
		uint64_t v;
		if (wide_indices) { memcpy(&v, indices + i * 8, 8); } else { uint32_t n; memcpy(&n, indices + i * 4, 4); v = n; }
		if (v >= vertex_count) throw ParseError("'" + filename + "': vertex index out of range");
		return v;
	};

	std::vector<Triangle> triangles;
	triangles.reserve(triangle_count);
	for (uint64_t t = 0; t < triangle_count; t++) {
		uint64_t i0 = index_at(3 * t), i1 = index_at(3 * t + 1), i2 = index_at(3 * t + 2);
		Vector3 p0 = positions.vector3(i0), p1 = positions.vector3(i1), p2 = positions.vector3(i2);
		Vector3 n0(0.0f), n1(0.0f), n2(0.0f);
		if (use_face_normals) {
			n0 = n1 = n2 = Vector3::normalize(Vector3::cross(p1 - p0, p2 - p0));
		} else if (has_normals) {
			n0 = normals.vector3(i0); n1 = normals.vector3(i1); n2 = normals.vector3(i2);
		}
		Vector2 t0(0.0f, 0.0f), t1(0.0f, 0.0f), t2(0.0f, 0.0f);
		if (has_tex_coords) { t0 = tex_coords.vector2(i0); t1 = tex_coords.vector2(i1); t2 = tex_coords.vector2(i2); }
		triangles.emplace_back(p0, p1, p2, n0, n1, n2, t0, t1, t2);
	}
	return triangles;
}

namespace {
// The reference seeds each hair file's ribbon orientations from a hash of its path
// (Core/Hash.h:5-17 FNV-1a over the chars, Core/Random.h:8-50 PCG).
struct HairRNG {
	uint64_t state;
	explicit HairRNG(const std::string & key) {
		uint64_t hash = 14695981039346656037ull;
		for (char c : key) { hash ^= uint64_t(int64_t(c)); hash *= 1099511628211ull; }
		state = (hash + 2891336453u) * 747796405u + 2891336453u;
	}
	uint32_t next() {
		uint32_t x = uint32_t(((state >> 18u) ^ state) >> 27u);
		uint32_t r = uint32_t(state >> 59u);
		state = state * 6364136223846793005ull + 1;
		return (x >> r) | (x << ((~r + 1) & 31));
	}
	float next_float() {
		uint32_t bits = 0x2f7fffffu; float scale; memcpy(&scale, &bits, 4);
		return float(next()) * scale;
	}
};
}

std::vector<Triangle> MitshairLoader::load(const std::string & filename, float radius) {
	std::string file = read_text_file(filename);

	std::vector<Vector3> vertices;
	std::vector<int>     strand_lengths;
	int current = 0;

	static const char MAGIC[] = "BINARY_HAIR";
	if (file.compare(0, sizeof(MAGIC) - 1, MAGIC) == 0) {
		ByteReader r { (const unsigned char *)file.data(), file.size(), sizeof(MAGIC) - 1, filename };
		r.read<uint32_t>(); // vertex count; the stream itself is authoritative
		while (r.pos < r.size) {
			float x = r.read<float>();
			if (std::isinf(x)) { // +inf closes a strand
				strand_lengths.push_back(current);
				current = 0;
			} else {
				float y = r.read<float>(), z = r.read<float>();
				vertices.emplace_back(x, y, z);
				current++;
			}
		}
	} else {
		Parser p(file, filename);
		while (!p.reached_end()) {
			if (is_newline(p.peek())) { // an empty line closes a strand
				strand_lengths.push_back(current);
				current = 0;
			} else {
				float x = p.parse_float(); p.skip_whitespace();
				float y = p.parse_float(); p.skip_whitespace();
				float z = p.parse_float(); p.skip_whitespace();
				vertices.emplace_back(x, y, z);
				current++;
			}
			if (p.reached_end()) break;
			if (p.match('\r')) p.match('\n'); else if (!p.match('\n')) p.fail("expected end of line");
		}
	}

	HairRNG rng(filename);
	std::vector<Triangle> triangles;
	size_t first = 0;
	for (int length : strand_lengths) {
		const Vector3 * strand = vertices.data() + first;
		first += length;
		if (length < 2) {
			fprintf(stderr, "WARNING: %s: a hair strand was defined with less than 2 vertices!\n", filename.c_str());
			continue;
		}
		float angle = PI * rng.next_float();

		Vector3 direction = Vector3::normalize(strand[1] - strand[0]);
		Vector3 side      = Quaternion::axis_angle(direction, angle) * Math::orthogonal(direction);
		Vector3 previous_a = strand[0] + radius * side;
		Vector3 previous_b = strand[0] - radius * side;

		for (int v = 1; v < length; v++) {
			direction = Vector3::normalize(strand[v] - strand[v - 1]);
			if (std::isnan(direction.x + direction.y + direction.z)) side = Vector3(1.0f, 0.0f, 0.0f); // repeated vertex
			else side = Quaternion::axis_angle(direction, angle) * Math::orthogonal(direction);

			float r = Math::lerp(radius, 0.0f, float(v) / float(length - 1)); // tapers to a point at the tip
			Vector3 current_a = strand[v] + r * side;
			Vector3 current_b = strand[v] - r * side;

			triangles.emplace_back(previous_a, previous_b, current_a, Vector3(0.0f), Vector3(0.0f), Vector3(0.0f), Vector2(0.0f, 0.0f), Vector2(1.0f, 0.0f), Vector2(0.0f, 1.0f));
			triangles.emplace_back(previous_b, current_b,  current_a, Vector3(0.0f), Vector3(0.0f), Vector3(0.0f), Vector2(0.0f, 0.0f), Vector2(1.0f, 0.0f), Vector2(0.0f, 1.0f));
			previous_a = current_a;
			previous_b = current_b;
		}
	}
	return triangles;
}
