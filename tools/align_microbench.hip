// align_microbench.hip — does the placement of 8-byte (64-bit encoded) VALU instructions relative to 8-byte
// boundaries change the VALU issue rate on gfx950?  (the C3 inner loop runs 6-8 % faster or slower
// depending on the 4-byte phase of its first instruction.)  Every test is one asm loop placed on a 64-byte boundary plus
// PAD s_nops (4 bytes each), 8 independent accumulator chains, all waves of the chip resident.
// Also reports the shader clock the chip sustains under this load (s_memtime ticks against s_memrealtime's 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 tools/align_microbench.hip -o tools/align_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define ITERS 4096

#define A4 "v_add_f32 %0, %9, %0\n v_add_f32 %1, %9, %1\n v_add_f32 %2, %9, %2\n v_add_f32 %3, %9, %3\n v_add_f32 %4, %9, %4\n v_add_f32 %5, %9, %5\n v_add_f32 %6, %9, %6\n v_add_f32 %7, %9, %7\n"
#define F8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define K8 "v_fmaak_f32 %0, %0, %8, 0x3d2aaa8d\n v_fmaak_f32 %1, %1, %8, 0x3d2aaa8d\n v_fmaak_f32 %2, %2, %8, 0x3d2aaa8d\n v_fmaak_f32 %3, %3, %8, 0x3d2aaa8d\n v_fmaak_f32 %4, %4, %8, 0x3d2aaa8d\n v_fmaak_f32 %5, %5, %8, 0x3d2aaa8d\n v_fmaak_f32 %6, %6, %8, 0x3d2aaa8d\n v_fmaak_f32 %7, %7, %8, 0x3d2aaa8d\n"
// VOP2 with a 32-bit literal as src0 (8 bytes): the forms the compiler picks for x += K * n and n = tm + K
#define M8 "v_fmac_f32 %0, 0xbf317200, %8\n v_fmac_f32 %1, 0xbf317200, %8\n v_fmac_f32 %2, 0xbf317200, %8\n v_fmac_f32 %3, 0xbf317200, %8\n v_fmac_f32 %4, 0xbf317200, %8\n v_fmac_f32 %5, 0xbf317200, %8\n v_fmac_f32 %6, 0xbf317200, %8\n v_fmac_f32 %7, 0xbf317200, %8\n"
#define D8 "v_add_f32 %0, 0xcb400000, %0\n v_add_f32 %1, 0xcb400000, %1\n v_add_f32 %2, 0xcb400000, %2\n v_add_f32 %3, 0xcb400000, %3\n v_add_f32 %4, 0xcb400000, %4\n v_add_f32 %5, 0xcb400000, %5\n v_add_f32 %6, 0xcb400000, %6\n v_add_f32 %7, 0xcb400000, %7\n"
#define X8 "v_max_f32 %0, 0xf800000, %0\n v_max_f32 %1, 0xf800000, %1\n v_max_f32 %2, 0xf800000, %2\n v_max_f32 %3, 0xf800000, %3\n v_max_f32 %4, 0xf800000, %4\n v_max_f32 %5, 0xf800000, %5\n v_max_f32 %6, 0xf800000, %6\n v_max_f32 %7, 0xf800000, %7\n"
// the same operations in the dedicated-literal forms: D = S0 * K + S1 (fmamk), D = S0 * S1 + K (fmaak)
#define MK8 "v_fmamk_f32 %0, %8, 0xbf317200, %0\n v_fmamk_f32 %1, %8, 0xbf317200, %1\n v_fmamk_f32 %2, %8, 0xbf317200, %2\n v_fmamk_f32 %3, %8, 0xbf317200, %3\n v_fmamk_f32 %4, %8, 0xbf317200, %4\n v_fmamk_f32 %5, %8, 0xbf317200, %5\n v_fmamk_f32 %6, %8, 0xbf317200, %6\n v_fmamk_f32 %7, %8, 0xbf317200, %7\n"
#define AK1 "v_fmaak_f32 %0, 1.0, %0, 0xcb400000\n v_fmaak_f32 %1, 1.0, %1, 0xcb400000\n v_fmaak_f32 %2, 1.0, %2, 0xcb400000\n v_fmaak_f32 %3, 1.0, %3, 0xcb400000\n v_fmaak_f32 %4, 1.0, %4, 0xcb400000\n v_fmaak_f32 %5, 1.0, %5, 0xcb400000\n v_fmaak_f32 %6, 1.0, %6, 0xcb400000\n v_fmaak_f32 %7, 1.0, %7, 0xcb400000\n"
// alternating 4-byte / 8-byte: every 8-byte instruction has the same 8-byte phase
#define AF "v_add_f32 %0, %9, %0\n v_fma_f32 %1, %1, %8, %9\n v_add_f32 %2, %9, %2\n v_fma_f32 %3, %3, %8, %9\n v_add_f32 %4, %9, %4\n v_fma_f32 %5, %5, %8, %9\n v_add_f32 %6, %9, %6\n v_fma_f32 %7, %7, %8, %9\n"
#define AK "v_add_f32 %0, %9, %0\n v_fmaak_f32 %1, %1, %8, 0x3d2aaa8d\n v_add_f32 %2, %9, %2\n v_fmaak_f32 %3, %3, %8, 0x3d2aaa8d\n v_add_f32 %4, %9, %4\n v_fmaak_f32 %5, %5, %8, 0x3d2aaa8d\n v_add_f32 %6, %9, %6\n v_fmaak_f32 %7, %7, %8, 0x3d2aaa8d\n"
// 4, 4, 8, 8: the 8-byte instructions keep the phase of the block start
#define AAFF "v_add_f32 %0, %9, %0\n v_add_f32 %1, %9, %1\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_add_f32 %4, %9, %4\n v_add_f32 %5, %9, %5\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"

// the C3 loop's shape: 72 four-byte instructions, then 56 eight-byte ones in one run (LONGRUN); the same instructions with
// the eight-byte ones in runs of 8 (SPLIT8) and of 16 (SPLIT16)
#define LONGRUN A4 A4 A4 A4 A4 A4 A4 A4 A4 K8 K8 K8 K8 K8 K8 K8
#define SPLIT8 A4 K8 A4 K8 A4 K8 A4 K8 A4 K8 A4 K8 A4 K8 A4 A4
#define SPLIT16 A4 A4 K8 K8 A4 A4 K8 K8 A4 A4 K8 K8 A4 A4 K8 A4
#define LONGRUN_F A4 A4 A4 A4 A4 A4 A4 A4 A4 F8 F8 F8 F8 F8 F8 F8
#define LOOP1(PADSTR, BODY)                                                                                         \
    asm volatile(".p2align 6\n\t" PADSTR "1:\n\t" BODY                                                             \
                 "s_sub_u32 %10, %10, 1\n\t s_cmp_lg_u32 %10, 0\n\t s_cbranch_scc1 1b\n\t"                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                    \
                 : "v"(m), "v"(c), "s"(cnt) : "scc")

#define LOOP(PADSTR, BODY)                                                                                          \
    asm volatile(".p2align 6\n\t" PADSTR "1:\n\t" BODY BODY BODY BODY BODY BODY BODY BODY                            \
                 "s_sub_u32 %10, %10, 1\n\t s_cmp_lg_u32 %10, 0\n\t s_cbranch_scc1 1b\n\t"                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                    \
                 : "v"(m), "v"(c), "s"(cnt) : "scc")

#define P0 ""
#define P1 "s_nop 0\n\t"
#define P2 "s_nop 0\n\t s_nop 0\n\t"
#define P3 "s_nop 0\n\t s_nop 0\n\t s_nop 0\n\t"

template <int BODY, int PAD> __global__ void __launch_bounds__(256) k(float* out, float seed, unsigned long long* clk) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    int cnt = ITERS;
    asm volatile("" : "+s"(cnt));
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
#define CASE(B, STR) \
    if (BODY == B) { if (PAD == 0) LOOP(P0, STR); else if (PAD == 1) LOOP(P1, STR); else if (PAD == 2) LOOP(P2, STR); else LOOP(P3, STR); }
    CASE(0, F8) CASE(1, A4) CASE(2, K8) CASE(3, AF) CASE(4, AK) CASE(5, AAFF)
#define CASE1(B, STR) \
    if (BODY == B) { if (PAD == 0) LOOP1(P0, STR); else if (PAD == 1) LOOP1(P1, STR); else if (PAD == 2) LOOP1(P2, STR); else LOOP1(P3, STR); }
    CASE1(6, LONGRUN) CASE1(7, SPLIT8) CASE1(8, SPLIT16) CASE1(9, LONGRUN_F)
    CASE(10, M8) CASE(11, D8) CASE(12, X8) CASE(13, MK8) CASE(14, AK1)
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int BODY, int PAD> void run(const char* name, int blocksPerCU, int cus, float* d, unsigned long long* dclk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * blocksPerCU;
    hipLaunchKernelGGL((k<BODY, PAD>), dim3(blocks), dim3(256), 0, 0, d, 1.0f, dclk);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<BODY, PAD>), dim3(blocks), dim3(256), 0, 0, d, 1.0f, dclk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long h[2]; hipMemcpy(h, dclk, 16, hipMemcpyDeviceToHost);
    const double waveInstr = (double)blocks * 4 * ITERS * (BODY >= 6 ? 128 : 64);
    const double perSimdPerSec = waveInstr / (cus * 4.0) / (best * 1e-3);
    printf("%-34s pad %d, %d waves/SIMD: %8.3f ms  %.3f ns/instr/SIMD  (%.2f cycles @2.4 GHz)   s_memtime/s_memrealtime = %.3f\n", name, PAD, blocksPerCU,
           best, 1e9 / perSimdPerSec, 2.4e9 / perSimdPerSec, h[1] ? (double)h[0] / (double)h[1] : 0.0);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    int wallRate = 0; hipDeviceGetAttribute(&wallRate, hipDeviceAttributeWallClockRate, 0);
    printf("%s, %d CUs, clockRate %d kHz, wall clock rate %d kHz\n", p.name, cus, p.clockRate, wallRate);
    float* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    unsigned long long* dclk; hipMalloc(&dclk, 16);
    for (int w : {7, 8}) {
        if (w == 7) {
#define TWO(B, NAME) run<B, 0>(NAME, 7, cus, d, dclk); run<B, 1>(NAME, 7, cus, d, dclk); run<B, 0>(NAME, 7, cus, d, dclk); run<B, 1>(NAME, 7, cus, d, dclk);
            TWO(10, "v_fmac_f32 v, literal, v  x64")
            TWO(11, "v_add_f32 v, literal, v   x64")
            TWO(12, "v_max_f32 v, literal, v   x64")
            TWO(13, "v_fmamk_f32 (same as fmac) x64")
            TWO(14, "v_fmaak_f32 v,1.0,v,K (=add) x64")
            TWO(2, "v_fmaak_f32 v, v, v, K     x64")
            TWO(0, "v_fma_f32 VOP3             x64")
            TWO(1, "v_add_f32 4 B              x64")
            TWO(6, "72 x 4 B, then 56 x 8 B (fmaak)")
            TWO(7, "same, 8 B instrs in runs of 8")
            TWO(8, "same, 8 B instrs in runs of 16")
            TWO(9, "72 x 4 B, then 56 x 8 B (v_fma VOP3)")
        } else if (w == 9) {
#define ALLPADS(B, NAME) run<B, 0>(NAME, 8, cus, d, dclk); run<B, 1>(NAME, 8, cus, d, dclk); run<B, 2>(NAME, 8, cus, d, dclk); run<B, 3>(NAME, 8, cus, d, dclk);
            ALLPADS(0, "v_fma_f32 (8 B) x64")
            ALLPADS(1, "v_add_f32 (4 B) x64")
            ALLPADS(2, "v_fmaak_f32 (8 B, literal) x64")
            ALLPADS(3, "v_add 4 B / v_fma 8 B alternating")
            ALLPADS(4, "v_add 4 B / v_fmaak 8 B alternating")
            ALLPADS(5, "4,4,8,8")
        }
    }
    return 0;
}
