// oracle/ref/ref_cuda_common.h -- TEST INFRASTRUCTURE ONLY.
// What both CUDA-on-CPU harnesses (ref_cuda_harness.cpp: Pathtracer.cu, ref_cuda_ao_harness.cpp: AO.cu) put in
// front of the reference's device sources: the shim, the vendored CUDA math headers, and Util.h with its six PTX
// helpers (Util.h:280-341) parked under other names and replaced by C++ with the instructions' semantics.
#pragma once
#include "cuda_on_cpu.h"

#define sign_extend_s8x4 sign_extend_s8x4_ptx
#define msb              msb_ptx
#define vmin_min         vmin_min_ptx
#define vmin_max         vmin_max_ptx
#define vmax_min         vmax_min_ptx
#define vmax_max         vmax_max_ptx
#define __CUDACC__ 1   // keeps cuda_math.h's host fallbacks of fminf / fmaxf / min / max / rsqrtf out: see cuda_on_cpu.h
#include "cudart/cuda_math.h"
#undef __CUDACC__
#include "Util.h"
#undef sign_extend_s8x4
#undef msb
#undef vmin_min
#undef vmin_max
#undef vmax_min
#undef vmax_max
inline unsigned sign_extend_s8x4(unsigned x) { unsigned r = 0; for (int i = 0; i < 4; i++) if (x & (0x80u << (8 * i))) r |= 0xffu << (8 * i); return r; }
inline unsigned msb(unsigned x) { return x ? 31u - unsigned(__builtin_clz(x)) : 0xffffffffu; }
inline float vmin_min(float a, float b, float c) { int x = __float_as_int(a), y = __float_as_int(b), z = __float_as_int(c); int m = x < y ? x : y; return __int_as_float(m < z ? m : z); }
inline float vmin_max(float a, float b, float c) { int x = __float_as_int(a), y = __float_as_int(b), z = __float_as_int(c); int m = x < y ? x : y; return __int_as_float(m > z ? m : z); }
inline float vmax_min(float a, float b, float c) { int x = __float_as_int(a), y = __float_as_int(b), z = __float_as_int(c); int m = x > y ? x : y; return __int_as_float(m < z ? m : z); }
inline float vmax_max(float a, float b, float c) { int x = __float_as_int(a), y = __float_as_int(b), z = __float_as_int(c); int m = x > y ? x : y; return __int_as_float(m > z ? m : z); }

