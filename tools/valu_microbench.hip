// valu_microbench.hip — measures VALU issue rates on gfx950 to calibrate the roofline denominator:
// plain v_fma_f32 / v_add_f32 / v_mul_f32, packed v_pk_fma_f32, and the quarter-rate ops the
// hot loop uses (v_rsq_f32, v_sqrt_f32, v_ldexp_f32, v_rndne_f32, v_cvt_i32_f32).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITERS 16384

template <int OP> __global__ void __launch_bounds__(256) k(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
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
            } else if (OP == 9) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(1.0), "v"(0.0));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
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
    printf("%-14s waves/SIMD %d: %.3f ms, %.3f G wave-instr/s/SIMD -> %.2f cycles/instr @2.4GHz, chip %.1f T lane-ops/s\n", name, blocksPerCU,
           ms, perSimdPerSec / 1e9, 2.4e9 / perSimdPerSec, waveInstr * 64 / (ms * 1e-3) / 1e12);
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
    for (int w : {4, 5}) { run<0>("v_fma_f32", w, cus, d); }
    return 0;
}
