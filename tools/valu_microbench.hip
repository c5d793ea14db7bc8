// valu_microbench.hip — measures VALU issue rates on gfx950 to calibrate the roofline denominator:
// plain v_fma_f32 / v_add_f32 / v_mul_f32, packed v_pk_fma_f32, and the quarter-rate ops the
// hot loop uses (v_rsq_f32, v_sqrt_f32, v_ldexp_f32, v_rndne_f32, v_cvt_i32_f32).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITERS 16384

__device__ unsigned long long g_clk[2];
template <int OP> __global__ void __launch_bounds__(256) k(float* out, float seed) {
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = clock64(); r0 = wall_clock64(); }
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    double q0 = seed, q1 = seed + 1, q2 = seed + 2, q3 = seed + 3;   // register pairs of the mixed streams
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) {  // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 1) {  // v_add_f32
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            } else if (OP == 2) {  // v_pk_fma_f32 (2 floats per lane per instruction) on 4 register pairs
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(1.0), "v"(0.0));
            } else if (OP == 3) {  // v_rsq_f32
                asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                             "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 4) {  // v_sqrt_f32
                asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                             "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 5) {  // v_ldexp_f32
                asm volatile("v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n"
                             "v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0));
            } else if (OP == 6) {  // v_rndne_f32
                asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n"
                             "v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 7) {  // v_cndmask_b32 with vcc
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");
            } else if (OP == 8) {  // v_mul_f32 with an SGPR operand
                asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                             "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));
            } else if (OP == 10) {  // v_fmaak_f32 with a 32-bit literal
                asm volatile("v_fmaak_f32 %0, %0, %8, 0x3d2aaa8d\n v_fmaak_f32 %1, %1, %8, 0x3d2aaa8d\n v_fmaak_f32 %2, %2, %8, 0x3d2aaa8d\n v_fmaak_f32 %3, %3, %8, 0x3d2aaa8d\n"
                             "v_fmaak_f32 %4, %4, %8, 0x3d2aaa8d\n v_fmaak_f32 %5, %5, %8, 0x3d2aaa8d\n v_fmaak_f32 %6, %6, %8, 0x3d2aaa8d\n v_fmaak_f32 %7, %7, %8, 0x3d2aaa8d"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (OP == 11) {  // v_fma_f32 with inline constant 0.5
                asm volatile("v_fma_f32 %0, %0, %8, 0.5\n v_fma_f32 %1, %1, %8, 0.5\n v_fma_f32 %2, %2, %8, 0.5\n v_fma_f32 %3, %3, %8, 0.5\n"
                             "v_fma_f32 %4, %4, %8, 0.5\n v_fma_f32 %5, %5, %8, 0.5\n v_fma_f32 %6, %6, %8, 0.5\n v_fma_f32 %7, %7, %8, 0.5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (OP == 12) {  // v_lshl_add_u32
                asm volatile("v_lshl_add_u32 %0, %0, 23, %8\n v_lshl_add_u32 %1, %1, 23, %8\n v_lshl_add_u32 %2, %2, 23, %8\n v_lshl_add_u32 %3, %3, 23, %8\n"
                             "v_lshl_add_u32 %4, %4, 23, %8\n v_lshl_add_u32 %5, %5, 23, %8\n v_lshl_add_u32 %6, %6, 23, %8\n v_lshl_add_u32 %7, %7, 23, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (OP == 13) {  // v_cmp_nle_f32 -> vcc (VGPR operands)
                asm volatile("v_cmp_nle_f32 vcc, %0, %8\n v_cmp_nle_f32 vcc, %1, %8\n v_cmp_nle_f32 vcc, %2, %8\n v_cmp_nle_f32 vcc, %3, %8\n"
                             "v_cmp_nle_f32 vcc, %4, %8\n v_cmp_nle_f32 vcc, %5, %8\n v_cmp_nle_f32 vcc, %6, %8\n v_cmp_nle_f32 vcc, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            } else if (OP == 14) {  // v_min3_u32
                asm volatile("v_min3_u32 %0, %0, %8, %9\n v_min3_u32 %1, %1, %8, %9\n v_min3_u32 %2, %2, %8, %9\n v_min3_u32 %3, %3, %8, %9\n"
                             "v_min3_u32 %4, %4, %8, %9\n v_min3_u32 %5, %5, %8, %9\n v_min3_u32 %6, %6, %8, %9\n v_min3_u32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 15) {  // v_fmac_f32 (VOP2, dst = accumulator)
                asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                             "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 16) {  // v_fma_f32 with an SGPR operand
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed), "v"(c));
            } else if (OP == 17) {  // v_sub_f32 e32 with SGPR src0
                asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                             "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));
            } else if (OP == 18) {  // v_mul_f32 with literal
                asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0\n v_mul_f32 %1, 0x3fb8aa3b, %1\n v_mul_f32 %2, 0x3fb8aa3b, %2\n v_mul_f32 %3, 0x3fb8aa3b, %3\n"
                             "v_mul_f32 %4, 0x3fb8aa3b, %4\n v_mul_f32 %5, 0x3fb8aa3b, %5\n v_mul_f32 %6, 0x3fb8aa3b, %6\n v_mul_f32 %7, 0x3fb8aa3b, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 19) {  // v_max_f32 literal
                asm volatile("v_max_f32 %0, 0x0f800000, %0\n v_max_f32 %1, 0x0f800000, %1\n v_max_f32 %2, 0x0f800000, %2\n v_max_f32 %3, 0x0f800000, %3\n"
                             "v_max_f32 %4, 0x0f800000, %4\n v_max_f32 %5, 0x0f800000, %5\n v_max_f32 %6, 0x0f800000, %6\n v_max_f32 %7, 0x0f800000, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 20) {  // v_add_f32 literal
                asm volatile("v_add_f32 %0, 0x05800000, %0\n v_add_f32 %1, 0x05800000, %1\n v_add_f32 %2, 0x05800000, %2\n v_add_f32 %3, 0x05800000, %3\n"
                             "v_add_f32 %4, 0x05800000, %4\n v_add_f32 %5, 0x05800000, %5\n v_add_f32 %6, 0x05800000, %6\n v_add_f32 %7, 0x05800000, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 21) {  // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n"
                             "v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 22) {  // v_add_u32
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 23) {  // v_lshlrev_b32
                asm volatile("v_lshlrev_b32 %0, 23, %0\n v_lshlrev_b32 %1, 23, %1\n v_lshlrev_b32 %2, 23, %2\n v_lshlrev_b32 %3, 23, %3\n"
                             "v_lshlrev_b32 %4, 23, %4\n v_lshlrev_b32 %5, 23, %5\n v_lshlrev_b32 %6, 23, %6\n v_lshlrev_b32 %7, 23, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 24) {  // v_mul_u32_u24
                asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n"
                             "v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 25) {  // v_and_b32
                asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                             "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 26) {  // v_add_lshl_u32
                asm volatile("v_add_lshl_u32 %0, %0, %8, 23\n v_add_lshl_u32 %1, %1, %8, 23\n v_add_lshl_u32 %2, %2, %8, 23\n v_add_lshl_u32 %3, %3, %8, 23\n"
                             "v_add_lshl_u32 %4, %4, %8, 23\n v_add_lshl_u32 %5, %5, %8, 23\n v_add_lshl_u32 %6, %6, %8, 23\n v_add_lshl_u32 %7, %7, %8, 23"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 27) {  // v_med3_f32
                asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                             "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 28) {  // v_bfi_b32
                asm volatile("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n"
                             "v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 29) {  // v_mul_f32 omod
                asm volatile("v_mul_f32_e64 %0, %0, %8 mul:2\n v_mul_f32_e64 %1, %1, %8 mul:2\n v_mul_f32_e64 %2, %2, %8 mul:2\n v_mul_f32_e64 %3, %3, %8 mul:2\n"
                             "v_mul_f32_e64 %4, %4, %8 mul:2\n v_mul_f32_e64 %5, %5, %8 mul:2\n v_mul_f32_e64 %6, %6, %8 mul:2\n v_mul_f32_e64 %7, %7, %8 mul:2"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 30) {  // v_max_f32 vgpr
                asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 31) {  // v_mad_u64_u32?
                asm volatile("v_alignbit_b32 %0, %0, %8, 9\n v_alignbit_b32 %1, %1, %8, 9\n v_alignbit_b32 %2, %2, %8, 9\n v_alignbit_b32 %3, %3, %8, 9\n"
                             "v_alignbit_b32 %4, %4, %8, 9\n v_alignbit_b32 %5, %5, %8, 9\n v_alignbit_b32 %6, %6, %8, 9\n v_alignbit_b32 %7, %7, %8, 9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 32) {  // v_add3_u32
                asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n"
                             "v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 33) {  // v_lshl_or_b32
                asm volatile("v_lshl_or_b32 %0, %0, 23, %8\n v_lshl_or_b32 %1, %1, 23, %8\n v_lshl_or_b32 %2, %2, 23, %8\n v_lshl_or_b32 %3, %3, 23, %8\n"
                             "v_lshl_or_b32 %4, %4, 23, %8\n v_lshl_or_b32 %5, %5, 23, %8\n v_lshl_or_b32 %6, %6, 23, %8\n v_lshl_or_b32 %7, %7, 23, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 34) {  // v_sub_f32 vgpr
                asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                             "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 35) {  // v_mul_f32 vgpr
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 36) {  // v_fma_f32 neg
                asm volatile("v_fma_f32 %0, -%0, %8, %9\n v_fma_f32 %1, -%1, %8, %9\n v_fma_f32 %2, -%2, %8, %9\n v_fma_f32 %3, -%3, %8, %9\n"
                             "v_fma_f32 %4, -%4, %8, %9\n v_fma_f32 %5, -%5, %8, %9\n v_fma_f32 %6, -%6, %8, %9\n v_fma_f32 %7, -%7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 37) {  // v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 38) {  // v_exp_f32
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 39) {  // v_mov_b32
                asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                             "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 40) {  // v_xad_u32
                asm volatile("v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %8, %9\n v_xad_u32 %2, %2, %8, %9\n v_xad_u32 %3, %3, %8, %9\n"
                             "v_xad_u32 %4, %4, %8, %9\n v_xad_u32 %5, %5, %8, %9\n v_xad_u32 %6, %6, %8, %9\n v_xad_u32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 41) {  // v_pk_mad_u16
                asm volatile("v_pk_mad_u16 %0, %0, %8, %9 op_sel_hi:[0,1,1]\n v_pk_mad_u16 %1, %1, %8, %9 op_sel_hi:[0,1,1]\n v_pk_mad_u16 %2, %2, %8, %9 op_sel_hi:[0,1,1]\n v_pk_mad_u16 %3, %3, %8, %9 op_sel_hi:[0,1,1]\n"
                             "v_pk_mad_u16 %4, %4, %8, %9 op_sel_hi:[0,1,1]\n v_pk_mad_u16 %5, %5, %8, %9 op_sel_hi:[0,1,1]\n v_pk_mad_u16 %6, %6, %8, %9 op_sel_hi:[0,1,1]\n v_pk_mad_u16 %7, %7, %8, %9 op_sel_hi:[0,1,1]"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 42) {  // v_pk_add_u16
                asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                             "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 43) {  // v_mad_u32_u16
                asm volatile("v_mad_u32_u16 %0, %0, %8, %9\n v_mad_u32_u16 %1, %1, %8, %9\n v_mad_u32_u16 %2, %2, %8, %9\n v_mad_u32_u16 %3, %3, %8, %9\n"
                             "v_mad_u32_u16 %4, %4, %8, %9\n v_mad_u32_u16 %5, %5, %8, %9\n v_mad_u32_u16 %6, %6, %8, %9\n v_mad_u32_u16 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (OP == 50) {  // mix 6fma+2max
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_max_f32 %3, %3, %12\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_max_f32 %7, %7, %12"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 51) {  // mix 4fma+4max
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_max_f32 %1, %1, %12\n v_fma_f32 %2, %2, %12, %13\n v_max_f32 %3, %3, %12\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_max_f32 %5, %5, %12\n v_fma_f32 %6, %6, %12, %13\n v_max_f32 %7, %7, %12"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 52) {  // mix 6fma+2pkfma
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_pk_fma_f32 %11, %11, %14, %15\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_pk_fma_f32 %11, %11, %14, %15"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 53) {  // mix 4fma+4pkfma
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_pk_fma_f32 %9, %9, %14, %15\n v_fma_f32 %2, %2, %12, %13\n v_pk_fma_f32 %11, %11, %14, %15\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_pk_fma_f32 %9, %9, %14, %15\n v_fma_f32 %6, %6, %12, %13\n v_pk_fma_f32 %11, %11, %14, %15"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 54) {  // mix 6fma+2rsq
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_rsq_f32 %3, %3\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_rsq_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 55) {  // mix 7fma+1rsq
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_rsq_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 56) {  // mix 6fma+2lshl
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_lshl_add_u32 %3, %3, 23, %12\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_lshl_add_u32 %7, %7, 23, %12"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 57) {  // 8 pkfma (4 pairs)
                asm volatile("v_pk_fma_f32 %8, %8, %14, %15\n v_pk_fma_f32 %9, %9, %14, %15\n v_pk_fma_f32 %10, %10, %14, %15\n v_pk_fma_f32 %11, %11, %14, %15\n"
                             "v_pk_fma_f32 %8, %8, %14, %15\n v_pk_fma_f32 %9, %9, %14, %15\n v_pk_fma_f32 %10, %10, %14, %15\n v_pk_fma_f32 %11, %11, %14, %15"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 58) {  // mix 4fma+4pkmul
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_pk_mul_f32 %9, %9, %14\n v_fma_f32 %2, %2, %12, %13\n v_pk_mul_f32 %11, %11, %14\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_pk_mul_f32 %9, %9, %14\n v_fma_f32 %6, %6, %12, %13\n v_pk_mul_f32 %11, %11, %14"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 59) {  // mix 2fma+6pkfma
                asm volatile("v_fma_f32 %0, %0, %12, %13\n v_pk_fma_f32 %9, %9, %14, %15\n v_pk_fma_f32 %10, %10, %14, %15\n v_pk_fma_f32 %11, %11, %14, %15\n"
                             "v_fma_f32 %4, %4, %12, %13\n v_pk_fma_f32 %9, %9, %14, %15\n v_pk_fma_f32 %10, %10, %14, %15\n v_pk_fma_f32 %11, %11, %14, %15"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                             : "v"(m), "v"(c), "v"(1.0), "v"(0.0));
            } else if (OP == 9) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(1.0), "v"(0.0));
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = clock64() - c0; g_clk[1] = wall_clock64() - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(q0 + q1 + q2 + q3);
}

template <int OP> void run(const char* name, int blocksPerCU, int cus, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * blocksPerCU;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveInstr = (double)blocks * 4 * ITERS * REP;         // wave-level instructions
    const double perSimdPerSec = waveInstr / (cus * 4.0) / (ms * 1e-3);
    unsigned long long clk[2] = {0, 1};
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    const double mhz = 100.0 * (double)clk[0] / (double)clk[1];        // s_memrealtime ticks at 100 MHz
    printf("%-14s waves/SIMD %d: %.3f ms, %.3f G wave-instr/s/SIMD -> %.2f cycles/instr @2.4GHz, chip %.1f T lane-ops/s; shader clock %.0f MHz -> %.2f shader cycles/instr\n", name, blocksPerCU,
           ms, perSimdPerSec / 1e9, 2.4e9 / perSimdPerSec, waveInstr * 64 / (ms * 1e-3) / 1e12, mhz, mhz * 1e6 / perSimdPerSec);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    float* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, cus, d); run<1>("v_add_f32", w, cus, d); run<2>("v_pk_fma_f32", w, cus, d);
    }
    run<8>("v_mul_f32 sgpr", 8, cus, d);
    run<3>("v_rsq_f32", 8, cus, d); run<4>("v_sqrt_f32", 8, cus, d); run<5>("v_ldexp_f32", 8, cus, d);
    run<6>("v_rndne_f32", 8, cus, d); run<7>("v_cndmask_b32", 8, cus, d); run<9>("v_fma_f64", 8, cus, d);
    run<10>("v_fmaak lit", 8, cus, d); run<11>("v_fma inline.5", 8, cus, d); run<12>("v_lshl_add_u32", 8, cus, d);
    run<13>("v_cmp_nle_f32", 8, cus, d); run<14>("v_min3_u32", 8, cus, d); run<15>("v_fmac_f32", 8, cus, d);
    run<16>("v_fma sgpr", 8, cus, d); run<17>("v_sub sgpr", 8, cus, d); run<18>("v_mul literal", 8, cus, d);
    run<19>("v_max_f32 literal", 8, cus, d); run<20>("v_add_f32 literal", 8, cus, d); run<21>("v_mad_u32_u24", 8, cus, d); run<22>("v_add_u32", 8, cus, d); run<23>("v_lshlrev_b32", 8, cus, d); run<24>("v_mul_u32_u24", 8, cus, d); run<25>("v_and_b32", 8, cus, d); run<26>("v_add_lshl_u32", 8, cus, d); run<27>("v_med3_f32", 8, cus, d); run<28>("v_bfi_b32", 8, cus, d); run<29>("v_mul_f32 omod", 8, cus, d); run<30>("v_max_f32 vgpr", 8, cus, d); run<31>("v_mad_u64_u32?", 8, cus, d); run<32>("v_add3_u32", 8, cus, d); run<33>("v_lshl_or_b32", 8, cus, d); run<34>("v_sub_f32 vgpr", 8, cus, d); run<35>("v_mul_f32 vgpr", 8, cus, d); run<36>("v_fma_f32 neg", 8, cus, d); run<37>("v_rcp_f32", 8, cus, d); run<38>("v_exp_f32", 8, cus, d); run<39>("v_mov_b32", 8, cus, d); run<40>("v_xad_u32", 8, cus, d);
    run<41>("v_pk_mad_u16", 8, cus, d); run<42>("v_pk_add_u16", 8, cus, d); run<43>("v_mad_u32_u16", 8, cus, d);
    run<50>("mix 6fma+2max", 8, cus, d); run<51>("mix 4fma+4max", 8, cus, d); run<52>("mix 6fma+2pkfma", 8, cus, d); run<53>("mix 4fma+4pkfma", 8, cus, d); run<54>("mix 6fma+2rsq", 8, cus, d); run<55>("mix 7fma+1rsq", 8, cus, d); run<56>("mix 6fma+2lshl", 8, cus, d); run<57>("8 pkfma (4 pairs)", 8, cus, d); run<58>("mix 4fma+4pkmul", 8, cus, d); run<59>("mix 2fma+6pkfma", 8, cus, d);
    run<50>("mix 6fma+2max", 6, cus, d); run<51>("mix 4fma+4max", 6, cus, d); run<52>("mix 6fma+2pkfma", 6, cus, d); run<53>("mix 4fma+4pkfma", 6, cus, d); run<54>("mix 6fma+2rsq", 6, cus, d); run<55>("mix 7fma+1rsq", 6, cus, d); run<56>("mix 6fma+2lshl", 6, cus, d); run<57>("8 pkfma (4 pairs)", 6, cus, d); run<58>("mix 4fma+4pkmul", 6, cus, d); run<59>("mix 2fma+6pkfma", 6, cus, d);
    for (int w : {4, 5}) { run<0>("v_fma_f32", w, cus, d); }
    return 0;
}
