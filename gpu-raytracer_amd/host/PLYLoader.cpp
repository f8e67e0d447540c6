// Stanford PLY reader: ascii / binary_little_endian / binary_big_endian, elements `vertex`
// (x y z [nx ny nz] [u|s v|t], any scalar type; other properties are read and dropped) and `face`
// (vertex_index | vertex_indices list, fan-triangulated). Behaviour follows the reference
// (Src/Assets/PLYLoader.cpp:155-346): v is flipped (1 - v), absent normals stay zero and are then
// replaced by the face normal in the Triangle constructor, elements other than vertex / face are
// an error. Unlike the reference, extra face properties are consumed instead of ending the face,
// and indices / sizes are range-checked.
#include "Scene.h"
#include "XMLParser.h"

#include <cstdint>
#include <cstring>

namespace {

enum class Encoding { ASCII, LITTLE_ENDIAN_BINARY, BIG_ENDIAN_BINARY };
enum class Scalar   { I8, I16, I32, U8, U16, U32, F32, F64, NONE };
enum class Slot     { X, Y, Z, NX, NY, NZ, U, V, IGNORED, VERTEX_INDEX };

struct Property {
	Slot   slot = Slot::IGNORED;
	bool   is_list = false;
	Scalar type  = Scalar::NONE; // scalar type, or the list's item type
	Scalar count = Scalar::NONE; // the list's length type
};

struct Element {
	bool is_face = false;
	int  count   = 0;
	std::vector<Property> properties;
};

Scalar parse_scalar_type(Parser & p) {
	p.skip_whitespace();
	static const struct { const char * name; Scalar type; } NAMES[] = {
		{ "int8", Scalar::I8 }, { "char", Scalar::I8 }, { "int16", Scalar::I16 }, { "short", Scalar::I16 },
		{ "int32", Scalar::I32 }, { "int", Scalar::I32 }, { "uint8", Scalar::U8 }, { "uchar", Scalar::U8 },
		{ "uint16", Scalar::U16 }, { "ushort", Scalar::U16 }, { "uint32", Scalar::U32 }, { "uint", Scalar::U32 },
		{ "float32", Scalar::F32 }, { "float", Scalar::F32 }, { "float64", Scalar::F64 }, { "double", Scalar::F64 },
	};
	for (const auto & n : NAMES) if (p.match(n.name)) return n.type;
	return Scalar::NONE;
}

std::string parse_word(Parser & p) {
	p.skip_whitespace();
	const char * begin = p.cur;
	while (!p.reached_end() && !is_whitespace(p.peek()) && !is_newline(p.peek())) p.advance();
	return std::string(begin, p.cur);
}

template<typename T> T read_binary(Parser & p, bool swap) {
	if (size_t(p.end - p.cur) < sizeof(T)) p.fail("unexpected end of file inside the binary data");
	unsigned char bytes[sizeof(T)];
	memcpy(bytes, p.cur, sizeof(T));
	p.cur += sizeof(T); // not advance(): binary data has no lines to count
	if (swap) for (size_t i = 0; i < sizeof(T) / 2; i++) std::swap(bytes[i], bytes[sizeof(T) - 1 - i]);
	T value;
	memcpy(&value, bytes, sizeof(T));
	return value;
}

double read_scalar(Parser & p, Scalar type, Encoding encoding) {
	if (encoding == Encoding::ASCII) {
		p.skip_whitespace();
		return (type == Scalar::F32 || type == Scalar::F64) ? double(p.parse_float()) : double(p.parse_int());
	}
	bool swap = encoding == Encoding::BIG_ENDIAN_BINARY; // the host is little endian
	switch (type) {
		case Scalar::I8:  return read_binary<int8_t>  (p, swap);
		case Scalar::I16: return read_binary<int16_t> (p, swap);
		case Scalar::I32: return read_binary<int32_t> (p, swap);
		case Scalar::U8:  return read_binary<uint8_t> (p, swap);
		case Scalar::U16: return read_binary<uint16_t>(p, swap);
		case Scalar::U32: return read_binary<uint32_t>(p, swap);
		case Scalar::F32: return read_binary<float>   (p, swap);
		case Scalar::F64: return read_binary<double>  (p, swap);
		default: p.fail("invalid property type");
	}
}

void end_ascii_record(Parser & p, Encoding encoding) {
	if (encoding != Encoding::ASCII || p.reached_end()) return;
	p.skip_whitespace();
	if (p.match('\r')) p.match('\n'); else if (!p.match('\n') && !p.reached_end()) p.fail("expected end of line");
}

std::vector<Element> parse_header(Parser & p, Encoding * encoding) {
	if (!p.match("ply")) p.fail("not a PLY file");
	p.skip_whitespace_or_newline();
	if (!p.match("format")) p.fail("expected 'format'");
	p.skip_whitespace();
	if      (p.match("ascii"))                *encoding = Encoding::ASCII;
	else if (p.match("binary_little_endian")) *encoding = Encoding::LITTLE_ENDIAN_BINARY;
	else if (p.match("binary_big_endian"))    *encoding = Encoding::BIG_ENDIAN_BINARY;
	else p.fail("invalid PLY format");
	std::string version = parse_word(p);
	if (version != "1.0") fprintf(stderr, "WARNING: %s: PLY format version is not 1.0!\n", p.filename.c_str());

	std::vector<Element> elements;
	while (true) {
		p.skip_whitespace_or_newline();
		if (p.reached_end()) p.fail("header without end_header");
		if (p.match("end_header")) break;
		std::string keyword = parse_word(p);
		if (keyword == "comment" || keyword == "obj_info") {
			p.skip_line();
		} else if (keyword == "element") {
			std::string name = parse_word(p);
			if (name != "vertex" && name != "face") p.fail("unsupported element type '" + name + "'");
			Element element;
			element.is_face = name == "face";
			p.skip_whitespace();
			element.count = p.parse_int();
			if (element.count < 0) p.fail("negative element count");
			elements.push_back(std::move(element));
		} else if (keyword == "property") {
			if (elements.empty()) p.fail("property defined without element");
			Property property;
			p.skip_whitespace();
			if (p.match("list")) {
				property.is_list = true;
				property.count = parse_scalar_type(p);
				property.type  = parse_scalar_type(p);
				if (property.count == Scalar::NONE || property.count == Scalar::F32 || property.count == Scalar::F64) p.fail("invalid list length type");
			} else {
				property.type = parse_scalar_type(p);
			}
			if (property.type == Scalar::NONE) p.fail("invalid type");
			std::string name = parse_word(p);
			static const struct { const char * name; Slot slot; } SLOTS[] = {
				{ "x", Slot::X }, { "y", Slot::Y }, { "z", Slot::Z }, { "nx", Slot::NX }, { "ny", Slot::NY }, { "nz", Slot::NZ },
				{ "u", Slot::U }, { "s", Slot::U }, { "v", Slot::V }, { "t", Slot::V },
				{ "vertex_index", Slot::VERTEX_INDEX }, { "vertex_indices", Slot::VERTEX_INDEX },
			};
			for (const auto & s : SLOTS) if (name == s.name) property.slot = s.slot;
			if (property.slot == Slot::IGNORED) fprintf(stderr, "WARNING: %s: ignoring unsupported property '%s'!\n", p.filename.c_str(), name.c_str());
			elements.back().properties.push_back(property);
		} else {
			p.fail("unexpected '" + keyword + "' in the header");
		}
	}
	// exactly one line end separates the header from the (possibly binary) body
	p.skip_whitespace();
	if (p.match('\r')) p.match('\n'); else p.match('\n');
	return elements;
}

} // namespace

std::vector<Triangle> PLYLoader::load(const std::string & filename) {
	std::string file = read_text_file(filename); // opened in binary mode: the bytes are untouched
	Parser p(file, filename);

	Encoding encoding;
	std::vector<Element> elements = parse_header(p, &encoding);

	std::vector<Vector3> positions, normals;
	std::vector<Vector2> tex_coords;
	std::vector<Triangle> triangles;

	for (const Element & element : elements) {
		if (!element.is_face) {
			for (int i = 0; i < element.count; i++) {
				float value[int(Slot::VERTEX_INDEX) + 1] = { };
				for (const Property & property : element.properties) {
					if (property.is_list) { // a list on a vertex carries nothing we use: read past it
						size_t n = size_t(read_scalar(p, property.count, encoding));
						for (size_t k = 0; k < n; k++) read_scalar(p, property.type, encoding);
						continue;
					}
					value[int(property.slot)] = float(read_scalar(p, property.type, encoding));
				}
				positions .emplace_back(value[int(Slot::X)],  value[int(Slot::Y)],  value[int(Slot::Z)]);
				normals   .emplace_back(value[int(Slot::NX)], value[int(Slot::NY)], value[int(Slot::NZ)]);
				tex_coords.emplace_back(value[int(Slot::U)], 1.0f - value[int(Slot::V)]);
				end_ascii_record(p, encoding);
			}
			continue;
		}
		for (int i = 0; i < element.count; i++) {
			for (const Property & property : element.properties) {
				if (!property.is_list) { read_scalar(p, property.type, encoding); continue; }
				double size = read_scalar(p, property.count, encoding);
				if (size < 0) p.fail("negative list length");
				size_t n = size_t(size);
				if (property.slot != Slot::VERTEX_INDEX) {
					for (size_t k = 0; k < n; k++) read_scalar(p, property.type, encoding);
					continue;
				}
				if (n <= 2) p.fail("a face needs at least 3 indices");
				auto next_vertex = [&]() {
					double index = read_scalar(p, property.type, encoding);
					if (index < 0 || index >= double(positions.size())) p.fail("vertex index out of range");
					return size_t(index);
				};
				size_t first = next_vertex(), previous = next_vertex();
				for (size_t k = 2; k < n; k++) { // fan around the first corner
					size_t current = next_vertex();
					triangles.emplace_back(
						positions [first], positions [previous], positions [current],
						normals   [first], normals   [previous], normals   [current],
						tex_coords[first], tex_coords[previous], tex_coords[current]);
					previous = current;
				}
			}
			end_ascii_record(p, encoding);
		}
	}
	return triangles;
}
