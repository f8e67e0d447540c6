// Host triangle: three vertices with shading normals and uvs (24 floats).
// Construction rules follow the reference (Src/Renderer/Triangle.h:47-93):
// degenerate normals are replaced by the face normal, and the winding is
// reversed only when ALL three shading normals oppose the face normal.
#pragma once
#include "Math.h"

struct Triangle {
	Vector3 position_0, position_1, position_2;
	Vector3 normal_0, normal_1, normal_2;
	Vector2 tex_coord_0, tex_coord_1, tex_coord_2;

	Triangle() = default;
	Triangle(Vector3 p0, Vector3 p1, Vector3 p2, Vector3 n0, Vector3 n1, Vector3 n2, Vector2 t0, Vector2 t1, Vector2 t2)
		: position_0(p0), position_1(p1), position_2(p2), normal_0(n0), normal_1(n1), normal_2(n2), tex_coord_0(t0), tex_coord_1(t1), tex_coord_2(t2)
	{
		bool bad_0 = Math::approx_equal(Vector3::length(normal_0), 0.0f);
		bool bad_1 = Math::approx_equal(Vector3::length(normal_1), 0.0f);
		bool bad_2 = Math::approx_equal(Vector3::length(normal_2), 0.0f);
		if (bad_0 || bad_1 || bad_2) {
			Vector3 face_normal = face_normal_unit();
			if (bad_0) normal_0 = face_normal;
			if (bad_1) normal_1 = face_normal;
			if (bad_2) normal_2 = face_normal;
		}
		fix_winding_order_if_needed();
	}

	Vector3 face_normal_unit() const {
		return Vector3::normalize(Vector3::cross(position_1 - position_0, position_2 - position_0));
	}

	void fix_winding_order_if_needed() {
		Vector3 g = face_normal_unit();
		bool all_flipped = Vector3::dot(g, normal_0) < 0.0f && Vector3::dot(g, normal_1) < 0.0f && Vector3::dot(g, normal_2) < 0.0f;
		if (all_flipped) {
			std::swap(position_1,  position_2);
			std::swap(normal_1,    normal_2);
			std::swap(tex_coord_1, tex_coord_2);
		}
	}

	Vector3 get_center() const { return (position_0 + position_1 + position_2) / 3.0f; }
	AABB get_aabb() const { Vector3 v[3] = { position_0, position_1, position_2 }; return AABB::from_points(v, 3); }
};
static_assert(sizeof(Triangle) == 96, "Triangle must stay 24 tightly packed floats");
