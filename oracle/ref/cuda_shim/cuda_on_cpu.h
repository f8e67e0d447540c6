// oracle/ref/cuda_shim/cuda_on_cpu.h -- TEST INFRASTRUCTURE ONLY.
//
// Lets the reference's CUDA device code (/root/reference/Src/CUDA/*.cu, *.h) compile and run on the host CPU,
// verbatim, as the parity oracle of last resort: __device__ / __global__ functions become inline functions,
// __constant__ globals become inline variables, one "thread" at a time runs with threadIdx / blockIdx set by the
// harness, and the handful of CUDA built-ins the sources use are defined below with their documented semantics.
// Nothing here is product code and nothing of the reference is copied.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cassert>
#include <atomic>
#include <math.h>
using std::isfinite; using std::isnan; using std::isinf;   // CUDA has them in the global namespace

#include "crt/host_defines.h"
#include "cudart/vector_types.h"   // the reference's vendored copies (Src/CUDA/cudart)

// ---- vector_functions.h: constructors ------------------------------------------------------------------
#define GRT_MAKE2(T, S) inline T make_##T(S x, S y) { T v; v.x = x; v.y = y; return v; }
#define GRT_MAKE3(T, S) inline T make_##T(S x, S y, S z) { T v; v.x = x; v.y = y; v.z = z; return v; }
#define GRT_MAKE4(T, S) inline T make_##T(S x, S y, S z, S w) { T v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
GRT_MAKE2(float2, float) GRT_MAKE3(float3, float) GRT_MAKE4(float4, float)
GRT_MAKE2(int2, int) GRT_MAKE3(int3, int) GRT_MAKE4(int4, int)
GRT_MAKE2(uint2, unsigned) GRT_MAKE3(uint3, unsigned) GRT_MAKE4(uint4, unsigned)
GRT_MAKE2(uchar2, unsigned char) GRT_MAKE4(uchar4, unsigned char)
GRT_MAKE2(ushort2, unsigned short)
#undef GRT_MAKE2
#undef GRT_MAKE3
#undef GRT_MAKE4

// ---- the execution model: one thread at a time ---------------------------------------------------------
struct grt_dim3 { unsigned x = 1, y = 1, z = 1; };
inline thread_local grt_dim3 threadIdx, blockIdx, blockDim, gridDim;
constexpr int warpSize = 32;

// ---- min / max: CUDA overloads them for every arithmetic type; the reference's vendored cuda_math.h only supplies
// the int pair on a host compiler, and without these a call like max(0.0001f, x) would silently truncate to int
// (cuda_math.h is included with __CUDACC__ defined, so that its host fallbacks -- whose fminf / fmaxf return NaN
// where CUDA's return the other operand -- stay out; libm's fminf / fmaxf have CUDA's semantics)
inline int      max(int a, int b)           { return a > b ? a : b; }
inline int      min(int a, int b)           { return a < b ? a : b; }
inline float    rsqrtf(float x)             { return 1.0f / sqrtf(x); }
inline float    max(float a, float b)       { return __builtin_fmaxf(a, b); }
inline float    min(float a, float b)       { return __builtin_fminf(a, b); }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline float    max(float a, int b)         { return __builtin_fmaxf(a, float(b)); }
inline float    max(int a, float b)         { return __builtin_fmaxf(float(a), b); }
inline float    min(float a, int b)         { return __builtin_fminf(a, float(b)); }
inline float    min(int a, float b)         { return __builtin_fminf(float(a), b); }

// ---- math intrinsics (fast-math forms map to the precise libm functions; comparisons allow for that) ------
inline float __saturatef(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); } // NaN -> 0 on the GPU; not exercised
inline void  __sincosf(float x, float * s, float * c) { *s = sinf(x); *c = cosf(x); }
inline void  sincosf_(float x, float * s, float * c) { *s = sinf(x); *c = cosf(x); }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __powf(float x, float y) { return powf(x, y); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline int      __float_as_int (float x)    { int i; memcpy(&i, &x, 4); return i; }
inline unsigned __float_as_uint(float x)    { unsigned i; memcpy(&i, &x, 4); return i; }
inline float    __int_as_float (int i)      { float x; memcpy(&x, &i, 4); return x; }
inline float    __uint_as_float(unsigned i) { float x; memcpy(&x, &i, 4); return x; }
inline int      __popc(unsigned x) { return __builtin_popcount(x); }
inline int      __clz(int x) { return x ? __builtin_clz(unsigned(x)) : 32; }
inline int      __ffs(int x) { return __builtin_ffs(x); }
inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
inline unsigned __byte_perm(unsigned a, unsigned b, unsigned s) {
	unsigned long long v = (unsigned long long)(b) << 32 | a; unsigned r = 0;
	for (int i = 0; i < 4; i++) r |= unsigned((v >> (8 * ((s >> (4 * i)) & 7))) & 0xff) << (8 * i);
	return r;
}
template<typename T> inline T __ldg(const T * p) { return *p; }

// ---- atomics: single-threaded emulation ---------------------------------------------------------------------
inline int      atomicAdd(int * p, int v)           { int old = *p; *p += v; return old; }
inline unsigned atomicAdd(unsigned * p, unsigned v) { unsigned old = *p; *p += v; return old; }
inline float    atomicAdd(float * p, float v)       { float old = *p; *p += v; return old; }
inline int      atomicAgg(int * p)                  { return atomicAdd(p, 1); }

// ---- warp-level built-ins: only the traversal kernels use them, and those are never run here ---------------
inline unsigned __activemask() { return 1u; }
inline unsigned __ballot_sync(unsigned, int p) { return p ? 1u : 0u; }
inline int      __any_sync(unsigned, int p) { return p; }
inline int      __all_sync(unsigned, int p) { return p; }
template<typename T> inline T __shfl_sync(unsigned, T v, int, int = 32) { return v; }
inline void     __syncthreads() { }
inline void     __syncwarp(unsigned = 0xffffffffu) { }
inline void     __threadfence() { }

// ---- textures and surfaces -------------------------------------------------------------------------------------
// A texture / surface object is a pointer to one of these descriptors; sampling is delegated to the harness
// (ref_cuda_harness.cpp), which applies the software texture unit rules of DESIGN.md section 5 -- the reference's
// NVIDIA texture unit is exactly the part of the device path that cannot be reproduced anywhere.
typedef unsigned long long cudaTextureObject_t;
typedef unsigned long long cudaSurfaceObject_t;
enum cudaSurfaceBoundaryMode { cudaBoundaryModeZero, cudaBoundaryModeClamp, cudaBoundaryModeTrap };

extern "C" {
	void grt_tex_fetch_1d  (cudaTextureObject_t t, float s, float out[4]);
	void grt_tex_fetch_2d  (cudaTextureObject_t t, float s, float u, float out[4]);
	void grt_tex_fetch_3d  (cudaTextureObject_t t, float s, float u, float r, float out[4]);
	void grt_tex_fetch_lod (cudaTextureObject_t t, float s, float u, float lod, float out[4]);
	void grt_tex_fetch_grad(cudaTextureObject_t t, float s, float u, const float dx[2], const float dy[2], float out[4]);
	void grt_surf_read (cudaSurfaceObject_t s, int x_bytes, int y, int z, void * dst, int bytes);
	void grt_surf_write(cudaSurfaceObject_t s, int x_bytes, int y, int z, const void * src, int bytes);
}
template<typename T> inline T grt_from4(const float v[4]);
template<> inline float  grt_from4<float >(const float v[4]) { return v[0]; }
template<> inline float2 grt_from4<float2>(const float v[4]) { return make_float2(v[0], v[1]); }
template<> inline float4 grt_from4<float4>(const float v[4]) { return make_float4(v[0], v[1], v[2], v[3]); }

template<typename T> inline T tex1D(cudaTextureObject_t t, float s)                   { float v[4]; grt_tex_fetch_1d(t, s, v); return grt_from4<T>(v); }
template<typename T> inline T tex2D(cudaTextureObject_t t, float s, float u)          { float v[4]; grt_tex_fetch_2d(t, s, u, v); return grt_from4<T>(v); }
template<typename T> inline T tex3D(cudaTextureObject_t t, float s, float u, float r) { float v[4]; grt_tex_fetch_3d(t, s, u, r, v); return grt_from4<T>(v); }
template<typename T> inline T tex2DLod(cudaTextureObject_t t, float s, float u, float lod) { float v[4]; grt_tex_fetch_lod(t, s, u, lod, v); return grt_from4<T>(v); }
template<typename T> inline T tex2DGrad(cudaTextureObject_t t, float s, float u, float2 dx, float2 dy) {
	float v[4], gx[2] = { dx.x, dx.y }, gy[2] = { dy.x, dy.y };
	grt_tex_fetch_grad(t, s, u, gx, gy, v);
	return grt_from4<T>(v);
}
template<typename T> inline void surf2Dread (T * dst, cudaSurfaceObject_t s, int x_bytes, int y, cudaSurfaceBoundaryMode = cudaBoundaryModeTrap)        { grt_surf_read(s, x_bytes, y, 0, dst, sizeof(T)); }
template<typename T> inline void surf3Dread (T * dst, cudaSurfaceObject_t s, int x_bytes, int y, int z, cudaSurfaceBoundaryMode = cudaBoundaryModeTrap) { grt_surf_read(s, x_bytes, y, z, dst, sizeof(T)); }
template<typename T> inline void surf2Dwrite(T value, cudaSurfaceObject_t s, int x_bytes, int y, cudaSurfaceBoundaryMode = cudaBoundaryModeTrap)        { grt_surf_write(s, x_bytes, y, 0, &value, sizeof(T)); }
template<typename T> inline void surf3Dwrite(T value, cudaSurfaceObject_t s, int x_bytes, int y, int z, cudaSurfaceBoundaryMode = cudaBoundaryModeTrap) { grt_surf_write(s, x_bytes, y, z, &value, sizeof(T)); }
