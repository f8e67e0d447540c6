// Tessellated Mitsuba primitive shapes with the transform baked into the vertices.
// Vertex order, normals (via the cofactor matrix) and uv assignment follow the
// reference (Src/Util/Geometry.cpp) because triangle order feeds the BVH and the
// light CDFs.
#include "Scene.h"

namespace {
Vector3 xform_normal(const Matrix4 & cofactor, const Vector3 & n) {
	return Vector3::normalize(Matrix4::transform_direction(cofactor, n));
}
}

std::vector<Triangle> Geometry::rectangle(const Matrix4 & transform) {
	Vector3 v0 = Matrix4::transform_position(transform, Vector3(-1.0f, +1.0f, 0.0f));
	Vector3 v1 = Matrix4::transform_position(transform, Vector3(+1.0f, +1.0f, 0.0f));
	Vector3 v2 = Matrix4::transform_position(transform, Vector3(+1.0f, -1.0f, 0.0f));
	Vector3 v3 = Matrix4::transform_position(transform, Vector3(-1.0f, -1.0f, 0.0f));
	Vector3 n = xform_normal(Matrix4::cofactor(transform), Vector3(0.0f, 0.0f, 1.0f));
	Vector2 t0(0.0f, 0.0f), t1(1.0f, 0.0f), t2(1.0f, 1.0f), t3(0.0f, 1.0f);
	return {
		Triangle(v0, v1, v2, n, n, n, t0, t1, t2),
		Triangle(v0, v2, v3, n, n, n, t0, t2, t3),
	};
}

std::vector<Triangle> Geometry::cube(const Matrix4 & transform) {
	Matrix4 cof = Matrix4::cofactor(transform);
	const Vector3 corner[8] = {
		Vector3(-1.0f, +1.0f, -1.0f), Vector3(+1.0f, +1.0f, -1.0f), Vector3(+1.0f, +1.0f, +1.0f), Vector3(-1.0f, +1.0f, +1.0f),
		Vector3(-1.0f, -1.0f, -1.0f), Vector3(+1.0f, -1.0f, -1.0f), Vector3(+1.0f, -1.0f, +1.0f), Vector3(-1.0f, -1.0f, +1.0f)
	};
	const Vector3 face_normal[6] = {
		Vector3(0.0f, +1.0f, 0.0f), Vector3(0.0f, 0.0f, -1.0f), Vector3(+1.0f, 0.0f, 0.0f),
		Vector3(0.0f, 0.0f, +1.0f), Vector3(-1.0f, 0.0f, 0.0f), Vector3(0.0f, -1.0f, 0.0f)
	};
	const int face_corner[6][4] = { { 0, 1, 2, 3 }, { 0, 1, 5, 4 }, { 1, 2, 6, 5 }, { 2, 3, 7, 6 }, { 3, 0, 4, 7 }, { 4, 5, 6, 7 } };
	const Vector2 uv[4] = { Vector2(0.0f, 0.0f), Vector2(1.0f, 0.0f), Vector2(1.0f, 1.0f), Vector2(0.0f, 1.0f) };

	Vector3 world[8];
	for (int i = 0; i < 8; i++) world[i] = Matrix4::transform_position(transform, corner[i]);

	std::vector<Triangle> triangles(12);
	for (int f = 0; f < 6; f++) {
		Vector3 n = xform_normal(cof, face_normal[f]);
		Vector3 a = world[face_corner[f][0]], b = world[face_corner[f][1]], c = world[face_corner[f][2]], d = world[face_corner[f][3]];
		triangles[2 * f]     = Triangle(a, b, c, n, n, n, uv[0], uv[1], uv[2]);
		triangles[2 * f + 1] = Triangle(a, c, d, n, n, n, uv[0], uv[2], uv[3]);
	}
	return triangles;
}

std::vector<Triangle> Geometry::disk(const Matrix4 & transform, int num_segments) {
	std::vector<Triangle> triangles(num_segments);
	Vector3 center = Matrix4::transform_position(transform, Vector3(0.0f, 0.0f, 0.0f));
	Vector3 prev   = Matrix4::transform_position(transform, Vector3(1.0f, 0.0f, 0.0f));
	Vector3 n = xform_normal(Matrix4::cofactor(transform), Vector3(0.0f, 0.0f, 1.0f));
	Vector2 uv_prev(1.0f, 0.5f);

	float step = TWO_PI / float(num_segments);
	float theta = 0.0f;
	for (int i = 0; i < num_segments; i++) {
		theta += step;
		float c = cosf(theta), s = sinf(theta);
		Vector3 curr = Matrix4::transform_position(transform, Vector3(c, s, 0.0f));
		Vector2 uv_curr(0.5f + 0.5f * c, 0.5f + 0.5f * s);
		triangles[i] = Triangle(prev, curr, center, n, n, n, uv_prev, uv_curr, Vector2(0.5f, 0.5f));
		prev = curr;
		uv_prev = uv_curr;
	}
	return triangles;
}

std::vector<Triangle> Geometry::cylinder(const Matrix4 & transform, const Vector3 & p0, const Vector3 & p1, float radius, int num_segments) {
	std::vector<Triangle> triangles(size_t(2) * num_segments);

	Vector3 a = Matrix4::transform_position(transform, p0);
	Vector3 b = Matrix4::transform_position(transform, p1);
	Vector3 axis = Vector3::normalize(b - a);
	Vector3 ortho_0 = Math::orthogonal(axis);
	Vector3 ortho_1 = Vector3::cross(axis, ortho_0);
	ortho_0 *= radius;
	ortho_1 *= radius;

	Vector3 n_prev = Vector3::normalize(ortho_0);
	Vector3 off_prev = ortho_0;
	float u_prev = 0.0f;

	float step = TWO_PI / float(num_segments);
	float theta = 0.0f;
	for (int i = 0; i < num_segments; i++) {
		theta += step;
		Vector3 off_curr = cosf(theta) * ortho_0 + sinf(theta) * ortho_1;
		Vector3 n_curr = Vector3::normalize(off_curr);
		float u_curr = float(i + 1) / float(num_segments);

		triangles[2 * i]     = Triangle(a + off_prev, a + off_curr, b + off_prev, n_prev, n_curr, n_prev, Vector2(u_prev, 0.0f), Vector2(u_curr, 0.0f), Vector2(u_prev, 1.0f));
		triangles[2 * i + 1] = Triangle(a + off_curr, b + off_curr, b + off_prev, n_curr, n_curr, n_prev, Vector2(u_curr, 0.0f), Vector2(u_curr, 1.0f), Vector2(u_prev, 1.0f));

		off_prev = off_curr;
		n_prev = n_curr;
		u_prev = u_curr;
	}
	return triangles;
}

// Subdivided icosahedron; new faces are appended in 3 blocks per level, which fixes the
// triangle order (reference: Util/Geometry.cpp:196-283).
std::vector<Triangle> Geometry::sphere(const Matrix4 & transform, int num_subdivisions) {
	constexpr float X = 0.525731112119133606f;
	constexpr float Z = 0.850650808352039932f;
	const Vector3 ico_vertex[12] = {
		Vector3(-X, 0.0f, Z), Vector3(X, 0.0f, Z),  Vector3(-X, 0.0f, -Z), Vector3(X, 0.0f, -Z),
		Vector3(0.0f, Z, X),  Vector3(0.0f, Z, -X), Vector3(0.0f, -Z, X),  Vector3(0.0f, -Z, -X),
		Vector3(Z, X, 0.0f),  Vector3(-Z, X, 0.0f), Vector3(Z, -X, 0.0f),  Vector3(-Z, -X, 0.0f)
	};
	const int ico_face[20][3] = {
		{ 0, 4, 1 },  { 0, 9, 4 },  { 9, 5, 4 },  { 4, 5, 8 },  { 4, 8, 1 },
		{ 8, 10, 1 }, { 8, 3, 10 }, { 5, 3, 8 },  { 5, 2, 3 },  { 2, 7, 3 },
		{ 7, 10, 3 }, { 7, 6, 10 }, { 7, 11, 6 }, { 11, 0, 6 }, { 0, 1, 6 },
		{ 6, 1, 10 }, { 9, 0, 11 }, { 9, 11, 2 }, { 9, 2, 5 },  { 7, 2, 11 }
	};

	size_t triangle_count = size_t(20) << (2 * num_subdivisions);
	std::vector<Triangle> triangles(triangle_count);
	for (int i = 0; i < 20; i++) {
		triangles[i].position_0 = ico_vertex[ico_face[i][0]];
		triangles[i].position_1 = ico_vertex[ico_face[i][1]];
		triangles[i].position_2 = ico_vertex[ico_face[i][2]];
	}

	size_t current = 20;
	for (int s = 0; s < num_subdivisions; s++) {
		for (size_t i = 0; i < current; i++) {
			Vector3 v0 = triangles[i].position_0, v1 = triangles[i].position_1, v2 = triangles[i].position_2;
			Vector3 v01 = Vector3::normalize(v0 + v1);
			Vector3 v12 = Vector3::normalize(v1 + v2);
			Vector3 v20 = Vector3::normalize(v2 + v0);

			Triangle & t0 = triangles[i];               t0.position_0 = v0;  t0.position_1 = v01; t0.position_2 = v20;
			Triangle & t1 = triangles[i + current];     t1.position_0 = v01; t1.position_1 = v1;  t1.position_2 = v12;
			Triangle & t2 = triangles[i + 2 * current]; t2.position_0 = v20; t2.position_1 = v12; t2.position_2 = v2;
			Triangle & t3 = triangles[i + 3 * current]; t3.position_0 = v01; t3.position_1 = v12; t3.position_2 = v20;
		}
		current *= 4;
	}

	Matrix4 cof = Matrix4::cofactor(transform);
	auto sphere_uv = [](const Vector3 & n) {
		return Vector2(0.5f + atan2f(-n.z, -n.x) * ONE_OVER_TWO_PI, 0.5f + asinf(-n.y) * ONE_OVER_PI);
	};
	for (Triangle & t : triangles) {
		t.normal_0 = xform_normal(cof, t.position_0);
		t.normal_1 = xform_normal(cof, t.position_1);
		t.normal_2 = xform_normal(cof, t.position_2);
		t.position_0 = Matrix4::transform_position(transform, t.position_0);
		t.position_1 = Matrix4::transform_position(transform, t.position_1);
		t.position_2 = Matrix4::transform_position(transform, t.position_2);
		t.tex_coord_0 = sphere_uv(t.normal_0);
		t.tex_coord_1 = sphere_uv(t.normal_1);
		t.tex_coord_2 = sphere_uv(t.normal_2);
		t.fix_winding_order_if_needed();
	}
	return triangles;
}
