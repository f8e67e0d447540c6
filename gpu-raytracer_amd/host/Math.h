// Host-side math types. Semantics follow the reference's Math/ library where
// the results feed the device data formats (reference: Src/Math/Vector3.h,
// Matrix4.h:10-31 row-major cells[col + row*4], Quaternion.h, AABB.h), because
// the BVH builder has to reproduce the reference's node bytes exactly:
//   * Vector3 / scalar multiplies by the reciprocal      (Vector3.h:83,93)
//   * AABB::surface_area = 2*(dx*dy + dy*dz + dz*dx)      (AABB.h:44-50)
//   * AABB::fix_if_needed grows by eps, 2eps, 4eps...     (AABB.h:31-42)
#pragma once
#include <cmath>
#include <cfloat>
#include <cstring>
#include <algorithm>

#define PI          3.14159265359f
#define ONE_OVER_PI 0.31830988618f
#define TWO_PI          6.28318530718f
#define ONE_OVER_TWO_PI 0.15915494309f
#define INVALID -1

struct Vector2 {
	float x = 0.0f, y = 0.0f;
	Vector2() = default;
	Vector2(float f) : x(f), y(f) { }
	Vector2(float x, float y) : x(x), y(y) { }
};
inline Vector2 operator+(const Vector2 & a, const Vector2 & b) { return Vector2(a.x + b.x, a.y + b.y); }
inline Vector2 operator-(const Vector2 & a, const Vector2 & b) { return Vector2(a.x - b.x, a.y - b.y); }
inline Vector2 operator*(float s, const Vector2 & a) { return Vector2(s * a.x, s * a.y); }

struct Vector3 {
	float x = 0.0f, y = 0.0f, z = 0.0f;
	Vector3() = default;
	Vector3(float f) : x(f), y(f), z(f) { }
	Vector3(float x, float y, float z) : x(x), y(y), z(z) { }

	float & operator[](int i)       { return (&x)[i]; }
	float   operator[](int i) const { return (&x)[i]; }

	static float dot(const Vector3 & a, const Vector3 & b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
	static float length_squared(const Vector3 & v) { return dot(v, v); }
	static float length(const Vector3 & v) { return sqrtf(length_squared(v)); }
	static Vector3 normalize(const Vector3 & v) { float inv = 1.0f / length(v); return Vector3(v.x * inv, v.y * inv, v.z * inv); }
	static Vector3 cross(const Vector3 & a, const Vector3 & b) {
		return Vector3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
	}
	static Vector3 min(const Vector3 & a, const Vector3 & b) { return Vector3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }
	static Vector3 max(const Vector3 & a, const Vector3 & b) { return Vector3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }
	template<typename F> static Vector3 apply(const Vector3 & v, F f) { return Vector3(f(v.x), f(v.y), f(v.z)); }

	Vector3 & operator+=(const Vector3 & v) { x += v.x; y += v.y; z += v.z; return *this; }
	Vector3 & operator-=(const Vector3 & v) { x -= v.x; y -= v.y; z -= v.z; return *this; }
	Vector3 & operator*=(const Vector3 & v) { x *= v.x; y *= v.y; z *= v.z; return *this; }
	Vector3 & operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};
inline Vector3 operator-(const Vector3 & v) { return Vector3(-v.x, -v.y, -v.z); }
inline Vector3 operator+(const Vector3 & a, const Vector3 & b) { return Vector3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vector3 operator-(const Vector3 & a, const Vector3 & b) { return Vector3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vector3 operator*(const Vector3 & a, const Vector3 & b) { return Vector3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vector3 operator/(const Vector3 & a, const Vector3 & b) { return Vector3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline Vector3 operator+(const Vector3 & a, float s) { return Vector3(a.x + s, a.y + s, a.z + s); }
inline Vector3 operator-(const Vector3 & a, float s) { return Vector3(a.x - s, a.y - s, a.z - s); }
inline Vector3 operator*(const Vector3 & a, float s) { return Vector3(a.x * s, a.y * s, a.z * s); }
inline Vector3 operator/(const Vector3 & a, float s) { float inv = 1.0f / s; return Vector3(a.x * inv, a.y * inv, a.z * inv); }
inline Vector3 operator+(float s, const Vector3 & a) { return Vector3(s + a.x, s + a.y, s + a.z); }
inline Vector3 operator-(float s, const Vector3 & a) { return Vector3(s - a.x, s - a.y, s - a.z); }
inline Vector3 operator*(float s, const Vector3 & a) { return Vector3(s * a.x, s * a.y, s * a.z); }
inline Vector3 operator/(float s, const Vector3 & a) { return Vector3(s / a.x, s / a.y, s / a.z); }

struct Vector4 {
	float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
	Vector4() = default;
	Vector4(float x, float y, float z, float w) : x(x), y(y), z(z), w(w) { }
};

namespace Math {
	template<typename T> inline T clamp(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
	template<typename T> inline T min(T a, T b) { return a < b ? a : b; }
	template<typename T> inline T max(T a, T b) { return a > b ? a : b; }
	template<typename T> inline T divide_round_up(T n, T d) { return (n + d - 1) / d; }
	template<typename T> inline T round_up(T x, T n) { T r = x % n; return r == 0 ? x : x + (n - r); }

	// Relative-error float compare (reference: Math/Math.h:27-44)
	inline bool approx_equal(float a, float b, float epsilon = 0.0001f) {
		float diff = fabsf(a - b);
		if (a == b) return true;
		if (a == 0.0f || b == 0.0f || diff < FLT_MIN) return diff < (epsilon * FLT_MIN);
		return diff / (fabsf(a) + fabsf(b)) < epsilon;
	}
	inline float deg_to_rad(float deg) { return deg / 180.0f * PI; }
	template<typename T> inline T lerp(const T & a, const T & b, float t) { return (1.0f - t) * a + t * b; }
	inline float luminance(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }
	inline float luminance(const Vector3 & c) { return luminance(c.x, c.y, c.z); }
	inline float gamma_to_linear(float x) {
		if (x <= 0.0f) return 0.0f;
		if (x >= 1.0f) return 1.0f;
		if (x < 0.04045f) return x / 12.92f;
		return powf((x + 0.055f) / 1.055f, 2.4f);
	}
	inline Vector3 orthogonal(const Vector3 & v) {
		float s = copysignf(1.0f, v.z);
		float a = -1.0f / (s + v.z);
		float b = v.x * v.y * a;
		return Vector3(1.0f + s * v.x * v.x * a, s * b, -s * v.x);
	}
}

struct Quaternion {
	float x = 0.0f, y = 0.0f, z = 0.0f, w = 1.0f;
	Quaternion() = default;
	Quaternion(float x, float y, float z, float w) : x(x), y(y), z(z), w(w) { }

	static Quaternion conjugate(const Quaternion & q) { return Quaternion(-q.x, -q.y, -q.z, q.w); }
	static Quaternion axis_angle(const Vector3 & axis, float angle) {
		float half = 0.5f * angle, s = sinf(half);
		return Quaternion(axis.x * s, axis.y * s, axis.z * s, cosf(half));
	}
	// Rotation that looks along 'forward' (reference: Math/Quaternion.h:37-70)
	static Quaternion look_rotation(const Vector3 & forward, const Vector3 & up) {
		Vector3 f = Vector3::normalize(forward);
		Vector3 r = Vector3::normalize(Vector3::cross(up, f));
		Vector3 u = Vector3::cross(f, r);
		float m00 = r.x, m01 = r.y, m02 = r.z;
		float m10 = u.x, m11 = u.y, m12 = u.z;
		float m20 = f.x, m21 = f.y, m22 = f.z;
		if (m22 < 0.0f) {
			if (m00 > m11) {
				float t = 1.0f + m00 - m11 - m22, s = 0.5f / sqrtf(t);
				return Quaternion(s * t, s * (m01 + m10), s * (m20 + m02), s * (m12 - m21));
			} else {
				float t = 1.0f - m00 + m11 - m22, s = 0.5f / sqrtf(t);
				return Quaternion(s * (m01 + m10), s * t, s * (m12 + m21), s * (m20 - m02));
			}
		} else {
			if (m00 < -m11) {
				float t = 1.0f - m00 - m11 + m22, s = 0.5f / sqrtf(t);
				return Quaternion(s * (m20 + m02), s * (m12 + m21), s * t, s * (m01 - m10));
			} else {
				float t = 1.0f + m00 + m11 + m22, s = 0.5f / sqrtf(t);
				return Quaternion(s * (m12 - m21), s * (m20 - m02), s * (m01 - m10), s * t);
			}
		}
	}
};
inline Vector3 operator*(const Quaternion & q, const Vector3 & v) {
	Vector3 u(q.x, q.y, q.z);
	return 2.0f * Vector3::dot(u, v) * u + (q.w * q.w - Vector3::dot(u, u)) * v + 2.0f * q.w * Vector3::cross(u, v);
}
inline Quaternion operator*(const Quaternion & a, const Quaternion & b) {
	return Quaternion(
		a.x * b.w + a.w * b.x + a.y * b.z - a.z * b.y,
		a.y * b.w + a.w * b.y + a.z * b.x - a.x * b.z,
		a.z * b.w + a.w * b.z + a.x * b.y - a.y * b.x,
		a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}

// Row-major 4x4, cells[col + row*4] (reference: Math/Matrix4.h:10-31)
struct alignas(16) Matrix4 {
	float cells[16];
	Matrix4() { memset(cells, 0, sizeof(cells)); cells[0] = cells[5] = cells[10] = cells[15] = 1.0f; }
	float & operator()(int row, int col)       { return cells[col + (row << 2)]; }
	float   operator()(int row, int col) const { return cells[col + (row << 2)]; }

	static Matrix4 create_translation(const Vector3 & t) { Matrix4 m; m(0, 3) = t.x; m(1, 3) = t.y; m(2, 3) = t.z; return m; }
	static Matrix4 create_rotation(const Quaternion & q) {
		float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
		float xz = q.x * q.z, xy = q.x * q.y, yz = q.y * q.z;
		float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
		Matrix4 m;
		m(0, 0) = 1.0f - 2.0f * (yy + zz); m(1, 0) = 2.0f * (xy + wz);        m(2, 0) = 2.0f * (xz - wy);
		m(0, 1) = 2.0f * (xy - wz);        m(1, 1) = 1.0f - 2.0f * (xx + zz); m(2, 1) = 2.0f * (yz + wx);
		m(0, 2) = 2.0f * (xz + wy);        m(1, 2) = 2.0f * (yz - wx);        m(2, 2) = 1.0f - 2.0f * (xx + yy);
		return m;
	}
	static Matrix4 create_scale(float s) { Matrix4 m; m(0, 0) = s; m(1, 1) = s; m(2, 2) = s; return m; }
	static Matrix4 create_scale(float x, float y, float z) { Matrix4 m; m(0, 0) = x; m(1, 1) = y; m(2, 2) = z; return m; }
	static Matrix4 perspective(float fov, float aspect, float near_plane, float far_plane) {
		float tan_half_fov = tanf(0.5f * fov);
		Matrix4 m;
		m(0, 0) = 1.0f / tan_half_fov;
		m(1, 1) = 1.0f / (aspect * tan_half_fov);
		m(2, 2) = -(far_plane + near_plane) / (far_plane - near_plane);
		m(3, 2) = -1.0f;
		m(2, 3) = -2.0f * (far_plane * near_plane) / (far_plane - near_plane);
		m(3, 3) = 0.0f;
		return m;
	}
	static Vector3 transform_position(const Matrix4 & m, const Vector3 & p) {
		return Vector3(
			m(0, 0) * p.x + m(0, 1) * p.y + m(0, 2) * p.z + m(0, 3),
			m(1, 0) * p.x + m(1, 1) * p.y + m(1, 2) * p.z + m(1, 3),
			m(2, 0) * p.x + m(2, 1) * p.y + m(2, 2) * p.z + m(2, 3));
	}
	static Vector3 transform_direction(const Matrix4 & m, const Vector3 & d) {
		return Vector3(
			m(0, 0) * d.x + m(0, 1) * d.y + m(0, 2) * d.z,
			m(1, 0) * d.x + m(1, 1) * d.y + m(1, 2) * d.z,
			m(2, 0) * d.x + m(2, 1) * d.y + m(2, 2) * d.z);
	}
	static float minor(const Matrix4 & m, int r0, int r1, int r2, int c0, int c1, int c2) {
		return
			m(r0, c0) * (m(r1, c1) * m(r2, c2) - m(r2, c1) * m(r1, c2)) -
			m(r0, c1) * (m(r1, c0) * m(r2, c2) - m(r2, c0) * m(r1, c2)) +
			m(r0, c2) * (m(r1, c0) * m(r2, c1) - m(r2, c0) * m(r1, c1));
	}
	// Cofactor matrix: transforms normals correctly under non-uniform scale (reference: Math/Matrix4.h:170-190)
	static Matrix4 cofactor(const Matrix4 & m) {
		Matrix4 r;
		r(0, 0) =  minor(m, 1, 2, 3, 1, 2, 3); r(0, 1) = -minor(m, 1, 2, 3, 0, 2, 3); r(0, 2) =  minor(m, 1, 2, 3, 0, 1, 3); r(0, 3) = -minor(m, 1, 2, 3, 0, 1, 2);
		r(1, 0) = -minor(m, 0, 2, 3, 1, 2, 3); r(1, 1) =  minor(m, 0, 2, 3, 0, 2, 3); r(1, 2) = -minor(m, 0, 2, 3, 0, 1, 3); r(1, 3) =  minor(m, 0, 2, 3, 0, 1, 2);
		r(2, 0) =  minor(m, 0, 1, 3, 1, 2, 3); r(2, 1) = -minor(m, 0, 1, 3, 0, 2, 3); r(2, 2) =  minor(m, 0, 1, 3, 0, 1, 3); r(2, 3) = -minor(m, 0, 1, 3, 0, 1, 2);
		r(3, 0) = -minor(m, 0, 1, 2, 1, 2, 3); r(3, 1) =  minor(m, 0, 1, 2, 0, 2, 3); r(3, 2) = -minor(m, 0, 1, 2, 0, 1, 3); r(3, 3) =  minor(m, 0, 1, 2, 0, 1, 2);
		return r;
	}
	// position / rotation (as look-rotation of 'forward') / uniform scale (reference: Math/Matrix4.h:192-203)
	static void decompose(const Matrix4 & m, Vector3 * position, Quaternion * rotation, float * scale, const Vector3 & forward = Vector3(0.0f, 0.0f, -1.0f)) {
		if (position) *position = Vector3(m(0, 3), m(1, 3), m(2, 3));
		if (rotation) *rotation = Quaternion::look_rotation(transform_direction(m, forward), Vector3(0.0f, 1.0f, 0.0f));
		if (scale) {
			float sx = Vector3::length(Vector3(m(0, 0), m(0, 1), m(0, 2)));
			float sy = Vector3::length(Vector3(m(1, 0), m(1, 1), m(1, 2)));
			float sz = Vector3::length(Vector3(m(2, 0), m(2, 1), m(2, 2)));
			*scale = cbrtf(sx * sy * sz);
		}
	}
	static Matrix4 abs(const Matrix4 & m) { Matrix4 r; for (int i = 0; i < 16; i++) r.cells[i] = fabsf(m.cells[i]); return r; }
};
inline Matrix4 operator*(const Matrix4 & l, const Matrix4 & r) {
	Matrix4 out;
	for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++)
		out(i, j) = l(i, 0) * r(0, j) + l(i, 1) * r(1, j) + l(i, 2) * r(2, j) + l(i, 3) * r(3, j);
	return out;
}

struct AABB {
	Vector3 min, max;

	static AABB create_empty() { AABB b; b.min = Vector3(+INFINITY); b.max = Vector3(-INFINITY); return b; }
	bool is_valid() const { return max.x > min.x && max.y > min.y && max.z > min.z; }
	bool is_empty() const {
		return min.x == INFINITY && min.y == INFINITY && min.z == INFINITY && max.x == -INFINITY && max.y == -INFINITY && max.z == -INFINITY;
	}
	void fix_if_needed(float epsilon = 0.001f) {
		if (is_empty()) return;
		for (int d = 0; d < 3; d++) {
			float eps = epsilon;
			while (max[d] - min[d] < eps) { min[d] -= eps; max[d] += eps; eps *= 2.0f; }
		}
	}
	float surface_area() const { Vector3 d = max - min; return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x); }
	void expand(const Vector3 & p) { min = Vector3::min(min, p); max = Vector3::max(max, p); }
	void expand(const AABB & b)    { min = Vector3::min(min, b.min); max = Vector3::max(max, b.max); }
	Vector3 get_center() const { return (min + max) * 0.5f; }
	static AABB from_points(const Vector3 * points, int n) {
		AABB b = create_empty();
		for (int i = 0; i < n; i++) b.expand(points[i]);
		b.fix_if_needed();
		return b;
	}
	// Intersection of two boxes; empty unless it has volume (reference: Math/AABB.cpp:39-50)
	static AABB overlap(const AABB & a, const AABB & b) {
		AABB r; r.min = Vector3::max(a.min, b.min); r.max = Vector3::min(a.max, b.max);
		return r.is_valid() ? r : create_empty();
	}
	// AABB of an OBB via component-wise |M| (reference: Math/AABB.cpp:55-69)
	static AABB transform(const AABB & aabb, const Matrix4 & m) {
		Vector3 center = 0.5f * (aabb.min + aabb.max);
		Vector3 extent = 0.5f * (aabb.max - aabb.min);
		Vector3 c = Matrix4::transform_position(m, center);
		Vector3 e = Matrix4::transform_direction(Matrix4::abs(m), extent);
		AABB r; r.min = c - e; r.max = c + e; return r;
	}
};
