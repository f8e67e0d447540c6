// oracle/ref/cuda_shim/crt/host_defines.h -- TEST INFRASTRUCTURE ONLY.
// Stands in for the CUDA toolkit header that the reference's vendored cudart/vector_types.h includes, so that
// the reference's device sources can be compiled for the host CPU (see cuda_on_cpu.h).
#pragma once
#define __host__
#define __device__ inline
#define __global__ inline
#define __constant__
#define __shared__
#define __forceinline__ inline
#define __noinline__
#define __restrict__ __restrict
#define __align__(n) __attribute__((aligned(n)))
#define __builtin_align__(n) __attribute__((aligned(n)))
#define __device_builtin__
#define __device_builtin_texture_type__
#define __device_builtin_surface_type__
#define __cudart_builtin__
#define __launch_bounds__(...)
