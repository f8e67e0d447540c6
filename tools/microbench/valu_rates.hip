// valu_rates.hip -- issue cost of the vector instructions the traversal loop is made of, on the chip the bench runs on.
// Development tool (not product). Every kernel runs ITER x 64 copies of one instruction on 8 independent register sets (no
// dependent chain shorter than 8 instructions), 256 threads per block, enough blocks for 8 waves per SIMD; reported: cycles per
// wave-instruction per SIMD (4 = the 16-lane SIMD's native rate for a 64-lane wave).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/microbench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

typedef float v2f __attribute__((ext_vector_type(2)));

#define KERNEL(name, DECL, BODY, SINK) \
__global__ void __launch_bounds__(256) name(float * out, float seed) { \
	DECL \
	for (int it = 0; it < ITER; it++) { REP64(BODY) } \
	float sink = 0; SINK \
	if (sink == 12345.678f) out[threadIdx.x] = sink; \
}

#define DECL_F float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7, b=seed*0.5f, c=seed*0.25f;
#define SINK_F sink = a0+a1+a2+a3+a4+a5+a6+a7;
#define B_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_fma, DECL_F, B_FMA, SINK_F)
#define B_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_max3, DECL_F, B_MAX3, SINK_F)
#define B_MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_min, DECL_F, B_MIN, SINK_F)
#define B_CVT(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a##i));
KERNEL(k_cvt_ubyte, DECL_F, B_CVT, SINK_F)
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
KERNEL(k_cndmask, DECL_F, B_CNDMASK, SINK_F)
#define B_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a##i), "v"(b) : "vcc");
KERNEL(k_cmp, DECL_F, B_CMP, SINK_F)
#define B_BFE(i) asm volatile("v_bfe_u32 %0, %0, 5, 3" : "+v"(a##i));
KERNEL(k_bfe, DECL_F, B_BFE, SINK_F)
#define B_LSHL(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a##i) : "v"(b));
KERNEL(k_lshl, DECL_F, B_LSHL, SINK_F)
#define B_SDWA(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(a##i) : "v"(b));
KERNEL(k_lshl_sdwa, DECL_F, B_SDWA, SINK_F)
#define B_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul_lo_u32, DECL_F, B_MULLO, SINK_F)
#define B_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul_u32_u24, DECL_F, B_MUL24, SINK_F)
#define B_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_or3, DECL_F, B_OR3, SINK_F)
#define B_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_perm, DECL_F, B_PERM, SINK_F)
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
KERNEL(k_rcp, DECL_F, B_RCP, SINK_F)
#define B_DIVSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
KERNEL(k_div_scale, DECL_F, B_DIVSCALE, SINK_F)
#define B_DIVFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
KERNEL(k_div_fmas, DECL_F, B_DIVFMAS, SINK_F)
#define B_DIVFIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_div_fixup, DECL_F, B_DIVFIXUP, SINK_F)
#define B_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_bcnt, DECL_F, B_BCNT, SINK_F)
#define B_FFBH(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a##i));
KERNEL(k_ffbh, DECL_F, B_FFBH, SINK_F)
#define B_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_bitop3, DECL_F, B_BITOP3, SINK_F)

#define DECL_P v2f a0={seed,seed},a1={seed+1,seed},a2={seed+2,seed},a3={seed+3,seed},a4={seed+4,seed},a5={seed+5,seed},a6={seed+6,seed},a7={seed+7,seed}, b={seed*0.5f,seed}, c={seed*0.25f,seed};
#define SINK_P sink = a0.x+a1.x+a2.x+a3.x+a4.x+a5.x+a6.x+a7.x+a0.y+a1.y+a2.y+a3.y+a4.y+a5.y+a6.y+a7.y;
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_pk_fma, DECL_P, B_PKFMA, SINK_P)
#define B_PKFMA_SEL(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_pk_fma_bcast, DECL_P, B_PKFMA_SEL, SINK_P)
#define B_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_pk_mul, DECL_P, B_PKMUL, SINK_P)
#define DECL_U unsigned long long a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; unsigned b=seed*3, c=seed*5;
#define SINK_U sink = float(a0+a1+a2+a3+a4+a5+a6+a7);
#define B_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
KERNEL(k_mad_u64_u32, DECL_U, B_MAD64, SINK_U)
#define B_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(a##i));
KERNEL(k_lshl_add_u64, DECL_U, B_LSHLADD64, SINK_U)

int main() {
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1000.0;
	float * out; hipMalloc(&out, 4096);
	struct Entry { const char * name; void (*kernel)(float *, float); };
	std::vector<Entry> entries = {
		{"v_fma_f32", k_fma}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_fma_f32 op_sel bcast", k_pk_fma_bcast}, {"v_pk_mul_f32", k_pk_mul}, {"v_max3_f32", k_max3}, {"v_min_f32", k_min},
		{"v_cvt_f32_ubyte1", k_cvt_ubyte}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32", k_cmp}, {"v_bfe_u32", k_bfe}, {"v_lshlrev_b32", k_lshl}, {"v_lshlrev_b32_sdwa", k_lshl_sdwa},
		{"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_u32_u24", k_mul_u32_u24}, {"v_or3_b32", k_or3}, {"v_perm_b32", k_perm}, {"v_bitop3_b32", k_bitop3}, {"v_bcnt_u32_b32", k_bcnt}, {"v_ffbh_u32", k_ffbh},
		{"v_rcp_f32", k_rcp}, {"v_div_scale_f32", k_div_scale}, {"v_div_fmas_f32", k_div_fmas}, {"v_div_fixup_f32", k_div_fixup}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64} };
	printf("%s: %d CUs, %.0f MHz (reported); cycles per wave-instruction per SIMD at 8 and at 1 wave(s) per SIMD\n", prop.name, cus, mhz);
	for (const Entry & e : entries) {
		double cycles[2];
		for (int pass = 0; pass < 2; pass++) {
			const int waves_per_simd = pass == 0 ? 8 : 1;
			const int blocks = cus * waves_per_simd;   // a block is 4 waves = one per SIMD
			hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
			hipLaunchKernelGGL(e.kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
			hipEventRecord(t0);
			hipLaunchKernelGGL(e.kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
			hipEventRecord(t1); hipEventSynchronize(t1);
			float ms = 0; hipEventElapsedTime(&ms, t0, t1);
			const double instr_per_simd = double(ITER) * 64 * waves_per_simd;
			cycles[pass] = ms * 1e-3 * mhz * 1e6 / instr_per_simd;
		}
		printf("  %-28s %6.2f  %6.2f\n", e.name, cycles[0], cycles[1]);
	}
	return 0;
}
