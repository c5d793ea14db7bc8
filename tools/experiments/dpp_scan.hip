// dpp_scan.hip — the wave64 inclusive / exclusive prefix sum and maximum the culling pass needs, built from DPP row shifts and v_readlane (no LDS
// traffic), checked against the host in double.  Stand-alone:  hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_scan tools/experiments/dpp_scan.hip && /tmp/dpp_scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ float dpp_or(float old, float v) {   // lanes without a source keep `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// inclusive prefix sum over the 64 lanes; total = lane 63's value (wave-uniform)
__device__ __forceinline__ float wave_scan_incl(float v, float& total) {
    const uint32_t lane = threadIdx.x & 63u;
    v += dpp_or<0x111>(0.0f, v);      // row_shr:1
    v += dpp_or<0x112>(0.0f, v);      // row_shr:2
    v += dpp_or<0x114>(0.0f, v);      // row_shr:4
    v += dpp_or<0x118>(0.0f, v);      // row_shr:8  -> inclusive within each row of 16
    const float r0 = readlane_f(v, 15), r1 = readlane_f(v, 31), r2 = readlane_f(v, 47), r3 = readlane_f(v, 63);
    const float p1 = r0, p2 = r0 + r1, p3 = (r0 + r1) + r2;
    v += lane >= 48u ? p3 : (lane >= 32u ? p2 : (lane >= 16u ? p1 : 0.0f));
    total = p3 + r3;
    return v;
}
__device__ __forceinline__ float wave_shift_right1(float v) { return dpp_or<0x138>(0.0f, v); }   // wave_shr:1, lane 0 gets 0
__device__ __forceinline__ float wave_max(float v) {                                             // v >= 0 everywhere
    v = fmaxf(v, dpp_or<0x111>(0.0f, v)); v = fmaxf(v, dpp_or<0x112>(0.0f, v)); v = fmaxf(v, dpp_or<0x114>(0.0f, v)); v = fmaxf(v, dpp_or<0x118>(0.0f, v));
    return fmaxf(fmaxf(readlane_f(v, 15), readlane_f(v, 31)), fmaxf(readlane_f(v, 47), readlane_f(v, 63)));
}

__global__ void k(const float* in, float* incl, float* excl, float* tot, float* mx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float t;
    const float s = wave_scan_incl(in[i], t);
    incl[i] = s; excl[i] = wave_shift_right1(s); tot[i] = t; mx[i] = wave_max(in[i]);
}

int main() {
    const int waves = 4096, n = waves * 64;
    std::vector<float> h(n);
    srand(7);
    for (int i = 0; i < n; ++i) { const double u = rand() / (double)RAND_MAX; h[i] = (float)std::exp2(-60.0 * u) * ((rand() & 7) ? 1.0f : 0.0f); }   // 18 decades, some zeros
    float *d, *a, *b, *c, *m;
    CHECK(hipMalloc(&d, n * 4)); CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4)); CHECK(hipMalloc(&c, n * 4)); CHECK(hipMalloc(&m, n * 4));
    CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, c, m);
    std::vector<float> A(n), B(n), C(n), M(n);
    CHECK(hipMemcpy(A.data(), a, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(B.data(), b, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(C.data(), c, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(M.data(), m, n * 4, hipMemcpyDeviceToHost));
    long bad = 0; double worst = 0;
    for (int w = 0; w < waves; ++w) {
        double run = 0, mxv = 0;
        for (int l = 0; l < 64; ++l) mxv = std::fmax(mxv, h[w * 64 + l]);
        for (int l = 0; l < 64; ++l) {
            const int i = w * 64 + l;
            const double before = run; run += h[i];
            const double ei = std::fabs(A[i] - run) / (run > 0 ? run : 1), ee = std::fabs(B[i] - before) / (before > 0 ? before : 1);
            worst = std::fmax(worst, std::fmax(ei, ee));
            if (ei > 1e-5 || ee > 1e-5 || (l > 0 && B[i] != A[i - 1]) || (l == 0 && B[i] != 0.0f) || M[i] != (float)mxv) ++bad;
        }
        for (int l = 0; l < 64; ++l) if (std::fabs(C[w * 64 + l] - run) > 1e-5 * run) ++bad;
    }
    printf("{\"waves\": %d, \"bad\": %ld, \"worst_relative_error\": %.3g}\n", waves, bad, worst);
    return bad != 0;
}
