// ta_rate.hip — how many cycles does one CU's texture addresser / vector L1 need per wave-level load instruction?
// (Question behind it: tools/profile_scene.sh counts 1.64e6 vector loads per CU in a 4000^2 Program.fs frame of 2.9e7 cycles — is the
// grid-union walk bound by the RATE of its record loads rather than by their latency?)  Stand-alone: not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ta_rate tools/experiments/ta_rate.hip && gpurun_out/ta_rate
//
// Every wave issues K independent loads per trip (addresses do not depend on loaded data) from a table of 32-byte records that stays in
// L1 (8 KB) or in L2 (4 MB); lanes are spread over `spread` consecutive records (1: the whole wave reads one record — the walk's common
// case, 64: every lane its own).  Reported: shader cycles per wave-level load instruction per CU (kernel cycles x 1 / loads per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
template <int W>   // dwords per lane and load: 1, 2, 4
__global__ void __launch_bounds__(256) ta_kernel(const uint4* __restrict__ tab, uint32_t mask, uint32_t spreadShift, uint32_t trips, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    const long long c0 = clock64();
    uint32_t rec = wave * 37u + (spreadShift >= 6 ? 0u : (lane >> spreadShift));   // spreadShift 6: one record per wave; 0: one per lane
    for (uint32_t t = 0; t < trips; ++t) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t r = (rec + (uint32_t)k * 5u) & mask;
            const char* p = reinterpret_cast<const char*>(tab) + (size_t)r * 32u + (k & 1) * 16u;
            // plain loads: the compiler places the waits (hand-written asm loads would let it reuse a destination register in flight)
            if (W == 4) { const u4v v = *reinterpret_cast<const u4v*>(p); acc ^= v.x ^ v.w; }
            else if (W == 2) { const u2v v = *reinterpret_cast<const u2v*>(p); acc ^= v.x ^ v.y; }
            else { acc ^= *reinterpret_cast<const uint32_t*>(p); }
        }
        rec += 41u;
    }
    const long long c1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (uint32_t)(c1 - c0); }
    if (acc == 0x12345u) out[0] = acc;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int mhz = 0; CHECK(hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0));   // kHz
    const size_t bytes = 4u << 20;
    uint4* tab; uint32_t* out;
    CHECK(hipMalloc(&tab, bytes)); CHECK(hipMemset(tab, 1, bytes)); CHECK(hipMalloc(&out, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"cus\": %d, \"clock_khz\": %d, \"rows\": [\n", cus, mhz);
    bool firstRow = true;
    const int widths[3] = {1, 2, 4};
    const uint32_t tables[2] = {8u << 10, 4u << 20};
    const uint32_t spreadShifts[4] = {6, 4, 2, 0};          // lanes per record 64, 16, 4, 1
    const int blocksPerCU[3] = {1, 3, 6};                   // 4, 12, 24 waves per CU
    for (int wi = 0; wi < 3; ++wi) for (int ti = 0; ti < 2; ++ti) for (int si = 0; si < 4; ++si) for (int bi = 0; bi < 3; ++bi) {
        const uint32_t mask = tables[ti] / 32u - 1u, trips = 2000;
        const int blocks = cus * blocksPerCU[bi];
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            if (widths[wi] == 4) hipLaunchKernelGGL(ta_kernel<4>, dim3(blocks), dim3(256), 0, 0, tab, mask, spreadShifts[si], trips, out);
            else if (widths[wi] == 2) hipLaunchKernelGGL(ta_kernel<2>, dim3(blocks), dim3(256), 0, 0, tab, mask, spreadShifts[si], trips, out);
            else hipLaunchKernelGGL(ta_kernel<1>, dim3(blocks), dim3(256), 0, 0, tab, mask, spreadShifts[si], trips, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        uint32_t h[2]; CHECK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
        const double loadsPerCU = (double)blocksPerCU[bi] * 4.0 * trips * 8.0;
        const double us = best * 1e3;
        printf("%s{\"dwords\": %d, \"table_kb\": %u, \"lanes_per_record\": %d, \"waves_per_cu\": %d, \"ms\": %.4f, \"ns_per_load_per_cu\": %.3f, \"cycles_per_load_per_cu\": %.2f}",
               firstRow ? "" : ",\n", widths[wi], tables[ti] >> 10, 1 << spreadShifts[si], blocksPerCU[bi] * 4, best, us * 1e3 / loadsPerCU, (double)h[1] / loadsPerCU);
        firstRow = false;
    }
    printf("\n]}\n");
    return 0;
}
