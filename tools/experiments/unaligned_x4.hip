// unaligned_x4.hip — does a global_load_dwordx4 from an address that is only 8-byte aligned return the right 16 bytes on gfx950, and at what rate?
// (The compact candidate lists of the grid-union walk hold 8-byte entries and a trip wants entries i and i+1 in one load.)  Stand-alone.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/unaligned_x4 tools/experiments/unaligned_x4.hip && /tmp/unaligned_x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u4v __attribute__((ext_vector_type(4), aligned(8)));

__global__ void __launch_bounds__(256) k(const uint32_t* __restrict__ tab, uint32_t maskEnt, uint32_t odd, uint32_t trips, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t bad = 0;
    uint32_t e = (wave * 37u + (lane >> 2)) * 2u + odd;      // entry index (8-byte entries); odd: the pair starts on an odd entry
    for (uint32_t t = 0; t < trips; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = (e + (uint32_t)j * 10u) & maskEnt;
            const u4v v = *reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(tab) + (size_t)i * 8u);
            bad += (v.x != 2u * i) + (v.y != 2u * i + 1u) + (v.z != 2u * i + 2u) + (v.w != 2u * i + 3u);
        }
        e += 82u;
    }
    if (bad) atomicAdd(out, bad);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t nEnt = 1u << 16;                               // 512 KB of entries (+ one spare entry behind the last)
    std::vector<uint32_t> h(2 * nEnt + 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;
    uint32_t *tab, *out; CHECK(hipMalloc(&tab, h.size() * 4)); CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (uint32_t odd = 0; odd < 2; ++odd) {
        CHECK(hipMemset(out, 0, 4));
        const uint32_t trips = 2000; const int blocks = cus * 6;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, nEnt - 2u, odd, trips, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        uint32_t bad; CHECK(hipMemcpy(&bad, out, 4, hipMemcpyDeviceToHost));
        printf("{\"pair_starts_on\": \"%s entry\", \"mismatching_dwords\": %u, \"ns_per_load_per_cu\": %.3f}\n", odd ? "an odd (8-byte aligned)" : "an even (16-byte aligned)", bad,
               best * 1e6 / (24.0 * trips * 8.0));
    }
    return 0;
}
