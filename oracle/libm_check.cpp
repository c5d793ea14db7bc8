// libm_check.cpp — TEST INFRASTRUCTURE (tests/test_libm_restatement.py): the product's restatement of glibc's expf / logf / powf
// (fraytracer_amd/csrc/ft_libm.h, compiled here for the host exactly as kernels.hip compiles it for the device) against the C runtime
// of the machine this runs on, over whole ranges of float bit patterns.  Nothing in the product links or loads this file.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../fraytracer_amd/csrc/ft_libm.h"

static const ft_u64 TAB[FT_LIBM_TAB_DOUBLES] = FT_LIBM_TAB_INIT;

static inline uint32_t canon(float r) { uint32_t v; memcpy(&v, &r, 4); return r != r ? 0x7fc00000u : v; }

extern "C" {
// op 0 expf, 1 logf, 2 powf(x, y); variant 1 = FMA build, 2 = SSE2 build.  Counts the bit patterns u in [lo, lo + count) whose restated
// result differs from the C runtime's; *first_bad = the first such pattern.
uint64_t chk_compare(int op, int variant, float y, uint32_t lo, uint64_t count, int nthreads, uint32_t* first_bad) {
    std::atomic<uint64_t> bad{0};
    std::atomic<uint64_t> first{~0ull};
    std::atomic<uint64_t> next{0};
    const uint64_t step = 1ull << 20;
    auto work = [&]() {
        for (;;) {
            const uint64_t b = next.fetch_add(step);
            if (b >= count) return;
            const uint64_t e = b + step < count ? b + step : count;
            uint64_t nb = 0;
            for (uint64_t k = b; k < e; ++k) {
                const uint32_t u = lo + (uint32_t)k;
                float x; memcpy(&x, &u, 4);
                volatile float xv = x;
                float want, got;
                if (op == 0) { want = expf(xv); got = variant == 1 ? ft_glibc_expf<true>(x, TAB) : ft_glibc_expf<false>(x, TAB); }
                else if (op == 1) { want = logf(xv); got = variant == 1 ? ft_glibc_logf<true>(x, TAB) : ft_glibc_logf<false>(x, TAB); }
                else { want = powf(xv, y); got = variant == 1 ? ft_glibc_powf<true>(x, y, TAB) : ft_glibc_powf<false>(x, y, TAB); }
                if (canon(want) != canon(got)) {
                    ++nb;
                    uint64_t cur = first.load();
                    while ((uint64_t)u < cur && !first.compare_exchange_weak(cur, (uint64_t)u)) {}
                }
            }
            bad += nb;
        }
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < (nthreads > 0 ? nthreads : 1); ++t) ts.emplace_back(work);
    for (auto& t : ts) t.join();
    if (first_bad) *first_bad = first.load() == ~0ull ? 0u : (uint32_t)first.load();
    return bad.load();
}
// powf over n explicit (x, y) pairs: mismatches against the C runtime
uint64_t chk_compare_pow_pairs(int variant, const float* x, const float* y, int64_t n, int64_t* first_bad) {
    uint64_t bad = 0;
    if (first_bad) *first_bad = -1;
    for (int64_t i = 0; i < n; ++i) {
        volatile float xv = x[i], yv = y[i];
        const float want = powf(xv, yv), got = variant == 1 ? ft_glibc_powf<true>(x[i], y[i], TAB) : ft_glibc_powf<false>(x[i], y[i], TAB);
        if (canon(want) != canon(got)) { if (!bad && first_bad) *first_bad = i; ++bad; }
    }
    return bad;
}
float chk_eval(int op, int variant, float x, float y) {
    if (op == 0) return variant == 1 ? ft_glibc_expf<true>(x, TAB) : ft_glibc_expf<false>(x, TAB);
    if (op == 1) return variant == 1 ? ft_glibc_logf<true>(x, TAB) : ft_glibc_logf<false>(x, TAB);
    return variant == 1 ? ft_glibc_powf<true>(x, y, TAB) : ft_glibc_powf<false>(x, y, TAB);
}
}
