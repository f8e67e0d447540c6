// oracle_math.h -- float3/float4 helpers for the CPU restatement (TEST INFRASTRUCTURE).
// Operator semantics follow the vector header the reference compiles against
// (Src/CUDA/cudart/cuda_math.h): component-wise ops, dot/cross/normalize/length.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

static inline float2 make_float2(float x, float y) { return { x, y }; }
static inline float3 make_float3(float x, float y, float z) { return { x, y, z }; }
static inline float3 make_float3(float s) { return { s, s, s }; }
static inline float3 make_float3(const float4 & v) { return { v.x, v.y, v.z }; }
static inline float4 make_float4(float x, float y, float z, float w) { return { x, y, z, w }; }
static inline float4 make_float4(float s) { return { s, s, s, s }; }
static inline float4 make_float4(const float3 & v) { return { v.x, v.y, v.z, 0.0f }; }

static inline float2 operator+(float2 a, float2 b) { return { a.x + b.x, a.y + b.y }; }
static inline float2 operator-(float2 a, float2 b) { return { a.x - b.x, a.y - b.y }; }
static inline float2 operator*(float s, float2 a) { return { s * a.x, s * a.y }; }
static inline float2 operator*(float2 a, float s) { return { a.x * s, a.y * s }; }
static inline float  dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }

static inline float3 operator-(float3 a) { return { -a.x, -a.y, -a.z }; }
static inline float3 operator+(float3 a, float3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline float3 operator-(float3 a, float3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline float3 operator*(float3 a, float3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
static inline float3 operator/(float3 a, float3 b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
static inline float3 operator*(float3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static inline float3 operator*(float s, float3 a) { return { s * a.x, s * a.y, s * a.z }; }
static inline float3 operator/(float3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
static inline float3 operator/(float s, float3 a) { return { s / a.x, s / a.y, s / a.z }; }
static inline float3 operator+(float3 a, float s) { return { a.x + s, a.y + s, a.z + s }; }
static inline float3 operator-(float3 a, float s) { return { a.x - s, a.y - s, a.z - s }; }
static inline float3 operator-(float s, float3 a) { return { s - a.x, s - a.y, s - a.z }; }
static inline float3 & operator+=(float3 & a, float3 b) { a = a + b; return a; }
static inline float3 & operator-=(float3 & a, float3 b) { a = a - b; return a; }
static inline float3 & operator*=(float3 & a, float3 b) { a = a * b; return a; }
static inline float3 & operator*=(float3 & a, float s) { a = a * s; return a; }
static inline float3 & operator/=(float3 & a, float s) { a = a / s; return a; }

static inline float4 operator+(float4 a, float4 b) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
static inline float4 operator-(float4 a, float4 b) { return { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; }
static inline float4 operator*(float4 a, float s) { return { a.x * s, a.y * s, a.z * s, a.w * s }; }
static inline float4 operator*(float s, float4 a) { return { s * a.x, s * a.y, s * a.z, s * a.w }; }
static inline float4 operator*(float4 a, float4 b) { return { a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; }
static inline float4 operator/(float4 a, float s) { return { a.x / s, a.y / s, a.z / s, a.w / s }; }
static inline float4 & operator+=(float4 & a, float4 b) { a = a + b; return a; }
static inline float4 & operator*=(float4 & a, float s) { a = a * s; return a; }

static inline float  dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float3 cross(float3 a, float3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline float  length(float3 a) { return sqrtf(dot(a, a)); }
static inline float  length(float2 a) { return sqrtf(dot(a, a)); }
static inline float3 normalize(float3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }

static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline float saturate(float v) { return clampf(v, 0.0f, 1.0f); }

static inline uint32_t float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int32_t  float_as_int(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float    int_as_float(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
