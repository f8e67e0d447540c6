// sincosf(x) against sinf(x) / cosf(x) on the device: the shade kernels' sincos_pair (rt_shading.h) switched from the two calls
// to the one (half the instructions: each of sinf / cosf evaluates both polynomials). Do the bits agree?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/sincos_check tools/microbench/sincos_check.hip && /tmp/sincos_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void check(unsigned long long * mismatches, float lo, float hi, unsigned n) {
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float x = lo + (hi - lo) * (float(i) / float(n));
	float s, c; sincosf(x, &s, &c);
	float s2 = sinf(x), c2 = cosf(x);
	if (__float_as_uint(s) != __float_as_uint(s2)) atomicAdd(&mismatches[0], 1ull);
	if (__float_as_uint(c) != __float_as_uint(c2)) atomicAdd(&mismatches[1], 1ull);
}
int main() {
	unsigned long long * d; hipMalloc(&d, 16);
	const float ranges[3][2] = { { -0.7853982f, 2.3561945f }, { 0.0f, 6.2831855f }, { -100.0f, 100.0f } };   // sample_disk's phi, 2 pi u, and beyond
	for (auto & r : ranges) {
		hipMemset(d, 0, 16);
		unsigned n = 1u << 26;
		hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, d, r[0], r[1], n);
		unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
		printf("sincosf vs sinf / cosf on [%g, %g], %u arguments: %llu sines and %llu cosines differ\n", r[0], r[1], n, h[0], h[1]);
	}
	return 0;
}
