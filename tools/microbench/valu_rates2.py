#!/usr/bin/env python3
"""Issue cost of single vector instructions on the chip this runs on (development tool, not product).
Generates one kernel per instruction -- ONE asm block of 64 copies on 8 independent register sets inside a loop, so the compiler
adds nothing between them (valu_rates.hip's per-instruction asm statements made it insert s_nop behind every vcc user) --,
compiles with hipcc and prints cycles per wave-instruction per SIMD at 8 waves and at 1 wave per SIMD.
   python tools/microbench/valu_rates2.py [out.txt]"""
import os, subprocess, sys, tempfile

# name -> (template with {a} = accumulator register (index 0..7 via operand), extra), operands: %0..%7 accumulators (32-bit), %8 b, %9 c
F32 = {
    "v_fma_f32": "v_fma_f32 {a}, {a}, %8, %9",
    "v_mul_f32": "v_mul_f32 {a}, {a}, %8",
    "v_add_f32": "v_add_f32 {a}, {a}, %8",
    "v_sub_f32": "v_sub_f32 {a}, {a}, %8",
    "v_max_f32": "v_max_f32 {a}, {a}, %8",
    "v_min3_f32": "v_min3_f32 {a}, {a}, %8, %9",
    "v_med3_f32": "v_med3_f32 {a}, {a}, %8, %9",
    "v_fma_mix_f32 (f16 lo, f32, f32)": "v_fma_mix_f32 {a}, {a}, %8, %9 op_sel_hi:[1,0,0]",
    "v_fma_mix_f32 (f16 hi, f32, f32)": "v_fma_mix_f32 {a}, {a}, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]",
    "v_cvt_f32_ubyte0": "v_cvt_f32_ubyte0 {a}, {a}",
    "v_cvt_f32_ubyte3": "v_cvt_f32_ubyte3 {a}, {a}",
    "v_cvt_f32_f16": "v_cvt_f32_f16 {a}, {a}",
    "v_cvt_f32_u32": "v_cvt_f32_u32 {a}, {a}",
    "v_cndmask_b32 (sgpr mask)": "v_cndmask_b32_e64 {a}, {a}, %8, s[10:11]",
    "v_cndmask_b32 (vcc)": "v_cndmask_b32 {a}, {a}, %8, vcc",
    "v_cmp_lt_f32 -> vcc": "v_cmp_lt_f32 vcc, {a}, %8",
    "v_cmp_lt_f32 -> sgpr": "v_cmp_lt_f32_e64 s[12:13], {a}, %8",
    "v_mov_b32": "v_mov_b32 {a}, %8",
    "v_and_b32": "v_and_b32 {a}, {a}, %8",
    "v_or_b32": "v_or_b32 {a}, {a}, %8",
    "v_xor_b32": "v_xor_b32 {a}, {a}, %8",
    "v_and_or_b32": "v_and_or_b32 {a}, {a}, %8, %9",
    "v_bitop3_b32": "v_bitop3_b32 {a}, {a}, %8, %9 bitop3:0x6c",
    "v_add_u32": "v_add_u32 {a}, {a}, %8",
    "v_addc_co_u32 (vcc in/out)": "v_addc_co_u32 {a}, vcc, {a}, {a}, vcc",
    "v_lshlrev_b32": "v_lshlrev_b32 {a}, 3, {a}",
    "v_lshrrev_b32": "v_lshrrev_b32 {a}, 3, {a}",
    "v_lshl_or_b32": "v_lshl_or_b32 {a}, {a}, 3, %8",
    "v_lshl_add_u32": "v_lshl_add_u32 {a}, {a}, 3, %8",
    "v_bfe_u32": "v_bfe_u32 {a}, {a}, 5, 3",
    "v_perm_b32": "v_perm_b32 {a}, {a}, %8, %9",
    "v_bcnt_u32_b32": "v_bcnt_u32_b32 {a}, {a}, %8",
    "v_ffbh_u32": "v_ffbh_u32 {a}, {a}",
    "v_mul_lo_u32": "v_mul_lo_u32 {a}, {a}, %8",
    "v_mul_u32_u24": "v_mul_u32_u24 {a}, {a}, %8",
    "v_mad_u32_u24": "v_mad_u32_u24 {a}, {a}, %8, %9",
    "v_rcp_f32": "v_rcp_f32 {a}, {a}",
    "v_rsq_f32": "v_rsq_f32 {a}, {a}",
    "v_sqrt_f32": "v_sqrt_f32 {a}, {a}",
    "v_exp_f32": "v_exp_f32 {a}, {a}",
    "v_log_f32": "v_log_f32 {a}, {a}",
    "v_sin_f32": "v_sin_f32 {a}, {a}",
    "v_div_scale_f32": "v_div_scale_f32 {a}, vcc, {a}, %8, %9",
    "v_div_fmas_f32": "v_div_fmas_f32 {a}, {a}, %8, %9",
    "v_div_fixup_f32": "v_div_fixup_f32 {a}, {a}, %8, %9",
    "v_readfirstlane_b32": "v_readfirstlane_b32 s14, {a}",
    "v_mov_b32 dpp quad_perm": "v_mov_b32_dpp {a}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
    "ds_bpermute_b32": "ds_bpermute_b32 {a}, %8, {a}",
    "ds_read_b32 (lane-linear)": "ds_read_b32 {a}, %10",
    "ds_read_u8 (lane-linear)": "ds_read_u8 {a}, %10",
    "ds_read_b64 -> 2 regs (lane-linear)": None,
    "pair: v_cvt_f32_ubyte0 + v_fma_f32": "v_cvt_f32_ubyte0 {a}, {a}\\n v_fma_f32 {a}, {a}, %8, %9",
    "pair: v_max3_f32 + v_fma_f32": "v_max3_f32 {a}, {a}, %8, %9\\n v_fma_f32 {a}, {a}, %8, %9",
}

def source():
    out = ["#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <vector>", "#define ITER 1000",
           "__shared__ unsigned lds_words[4096];"]
    names = []
    for idx, (name, tmpl) in enumerate(F32.items()):
        if tmpl is None:
            continue
        per = 2 if name.startswith("pair") else 1
        body = "\\n ".join(tmpl.format(a="%%%d" % (k % 8)) for k in range(64 // per))
        needs_wait = "ds_" in tmpl
        if needs_wait:
            body += "\\n s_waitcnt lgkmcnt(0)"
        out.append("""__global__ void __launch_bounds__(256) k%d(float * out, float seed) {
    float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7, b=seed*0.5f, c=seed*0.25f; unsigned addr = threadIdx.x * 4u;
    lds_words[threadIdx.x] = threadIdx.x; __syncthreads();
    for (int it = 0; it < ITER; it++) {
        asm volatile("s_mov_b64 s[10:11], exec\\n %s" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b), "v"(c), "v"(addr) : "vcc", "s10", "s11", "s12", "s13", "s14", "memory");
    }
    float sink = a0+a1+a2+a3+a4+a5+a6+a7;
    if (sink == 12345.678f) out[threadIdx.x] = sink;
}""" % (idx, body))
        names.append((name, idx, 64))
    out.append("struct Entry { const char * name; void (*kernel)(float *, float); int count; };")
    out.append("int main() { hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1000.0; float * out; (void)hipMalloc(&out, 4096);")
    out.append("  std::vector<Entry> entries = {%s};" % ", ".join('{"%s", k%d, %d}' % (n.replace('"', ''), i, c) for n, i, c in names))
    out.append(r"""  printf("%s: %d CUs, %.0f MHz; cycles per wave-instruction per SIMD at 8 waves / 1 wave per SIMD (pairs: per instruction of the pair)\n", prop.name, cus, mhz);
  for (const Entry & e : entries) { double cycles[2];
    for (int pass = 0; pass < 2; pass++) { const int waves = pass == 0 ? 8 : 1; const int blocks = cus * waves;
      hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
      hipLaunchKernelGGL(e.kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f); (void)hipEventRecord(t0);
      hipLaunchKernelGGL(e.kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0f); (void)hipEventRecord(t1); (void)hipEventSynchronize(t1);
      float ms = 0; (void)hipEventElapsedTime(&ms, t0, t1);
      cycles[pass] = ms * 1e-3 * mhz * 1e6 / (double(ITER) * e.count * waves); }
    printf("  %-40s %6.2f  %6.2f\n", e.name, cycles[0], cycles[1]); }
  return 0; }""")
    return "\n".join(out)

def main():
    work = tempfile.mkdtemp(prefix="valu_rates_")
    src = os.path.join(work, "valu_rates2.hip"); exe = os.path.join(work, "valu_rates2")
    open(src, "w").write(source())
    if "--source-only" in sys.argv:
        print(src); return
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, src], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-4000:]); sys.exit(1)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    print(r.stdout)
    outs = [a for a in sys.argv[1:] if not a.startswith("--")]
    if outs:
        open(outs[0], "w").write(r.stdout)

if __name__ == "__main__":
    main()
