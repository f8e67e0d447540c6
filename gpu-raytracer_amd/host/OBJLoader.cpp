// Wavefront OBJ reader: v / vt / vn / f with fan triangulation, negative (relative)
// indices, uv v-flip; everything else is skipped (reference: Src/Assets/OBJLoader.cpp).
#include "Scene.h"
#include "XMLParser.h"

namespace {
struct Corner { int v = 0, t = 0, n = 0; };
struct Face   { Corner c[3]; };

float next_float(Parser & p) { p.skip_whitespace(); return p.parse_float(); }
int   next_int  (Parser & p) { p.skip_whitespace(); return p.parse_int(); }

Corner parse_corner(Parser & p) {
	Corner c;
	c.v = next_int(p);
	if (p.match('/')) {
		if (p.match('/')) {
			c.n = next_int(p);
		} else {
			c.t = next_int(p);
			if (p.match('/')) c.n = next_int(p);
		}
	}
	return c;
}

bool starts_number(char ch) { return is_digit(ch) || ch == '+' || ch == '-' || ch == '.'; }

void skip_extra_numbers(Parser & p) { // w coordinate / vertex colours
	p.skip_whitespace();
	while (!p.reached_end() && starts_number(p.peek())) { p.parse_float(); p.skip_whitespace(); }
}

// 1-based, negative = relative to the end, 0 / out of range = absent
int resolve(int count, int index) {
	if (count == 0) return INVALID;
	int r = INVALID;
	if (index > 0) r = index - 1; else if (index < 0) r = count + index;
	return (r < 0 || r >= count) ? INVALID : r;
}
}

std::vector<Triangle> OBJLoader::load(const std::string & filename) {
	std::string text = read_text_file(filename);
	Parser p(text, filename);

	std::vector<Vector3> positions, normals;
	std::vector<Vector2> tex_coords;
	std::vector<Face> faces;

	while (!p.reached_end()) {
		if (p.match('#') || p.match("o ")) {
			p.skip_line();
		} else if (p.match("v ")) {
			float x = next_float(p), y = next_float(p), z = next_float(p);
			positions.emplace_back(x, y, z);
			skip_extra_numbers(p);
		} else if (p.match("vt ")) {
			float u = next_float(p), v = next_float(p);
			tex_coords.emplace_back(u, v);
			skip_extra_numbers(p);
		} else if (p.match("vn ")) {
			float x = next_float(p), y = next_float(p), z = next_float(p);
			normals.emplace_back(x, y, z);
		} else if (p.match("f ")) {
			Corner first = parse_corner(p), prev = parse_corner(p), curr = parse_corner(p);
			faces.push_back({ { first, prev, curr } });
			while (true) { // fan-triangulate polygons
				prev = curr;
				p.skip_whitespace();
				if (p.reached_end() || !(p.peek() == '-' || is_digit(p.peek()))) break;
				curr = parse_corner(p);
				faces.push_back({ { first, prev, curr } });
			}
		} else {
			p.skip_line();
		}
		p.skip_whitespace();
		p.match('\r');
		if (!p.reached_end()) p.expect('\n');
	}

	std::vector<Triangle> triangles(faces.size());
	for (size_t f = 0; f < faces.size(); f++) {
		Vector3 pos[3], nor[3];
		Vector2 tex[3];
		for (int i = 0; i < 3; i++) {
			int iv = resolve(int(positions .size()), faces[f].c[i].v);
			int it = resolve(int(tex_coords.size()), faces[f].c[i].t);
			int in = resolve(int(normals   .size()), faces[f].c[i].n);
			if (iv != INVALID) pos[i] = positions[iv];
			if (it != INVALID) { tex[i] = tex_coords[it]; tex[i].y = 1.0f - tex[i].y; }
			if (in != INVALID) nor[i] = normals[in];
		}
		triangles[f] = Triangle(pos[0], pos[1], pos[2], nor[0], nor[1], nor[2], tex[0], tex[1], tex[2]);
	}
	return triangles;
}
