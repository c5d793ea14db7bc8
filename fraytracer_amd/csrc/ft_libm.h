// ft_libm.h — glibc's expf / logf / powf restated for host and gfx950 device code: the arithmetic MathF.Exp / MathF.Log /
// MathF.Pow perform under .NET on Linux x86-64, where they are the C runtime's functions (SdfForm.fs:80,82 unionSmooth;
// FColor.fs:50-55 gammaInverse).
//
// THIRD-PARTY ALGORITHM, absent from /root/reference: GNU libc 2.35 (the libm of this image and of the GPU boxes),
// sysdeps/ieee754/flt-32/{e_expf.c, e_logf.c, e_powf.c, math_config.h} with the tables of e_exp2f_data.c, e_logf_data.c,
// e_powf_log2_data.c (Szabolcs Nagy's "optimized routines": double-precision table + polynomial evaluation, one rounding to
// float at the end; configuration of the x86-64 build: TOINT_INTRINSICS 0, WANT_ROUNDING 1, WANT_ERRNO 1, WANT_ERRNO_UFLOW 1,
// POWF_SCALE_BITS 0, round-to-nearest).  Constants are the published ones (cross-checked against the .rodata of
// /lib/x86_64-linux-gnu/libm.so.6).
//
// x86-64 glibc builds every one of the three functions twice from that one source and picks at load time (ifunc): `__*_fma`
// (compiled -mfma -mavx2; taken when the CPU has FMA and AVX2, i.e. on every host an MI355X sits in) and `__*_sse2`.  The FMA
// build contracts a*b+c wherever the source has one, the SSE2 build rounds every operation — the results differ in the last bit
// for a fraction of the inputs.  Both contraction patterns are restated here (template parameter FMA; the fused operations are
// those of the machine code of the two builds, read from the disassembly: tools/libm_variants.md) and proved bit-identical to
// the running libm over all 2^32 inputs by tests/test_libm_restatement.py (host, this header) and on the GPU by
// ft_selftest_libm (device, checksums per 2^24 inputs against the box's libm).
//
// Tables: 80 doubles, FT_LIBM_TAB below — [0,32) 2^(i/32) bit patterns minus (i << 47), [32,48) invc, [48,64) log(c), [64,80) log2(c).
#pragma once
#include "ft_math.h"

#define FT_LIBM_TAB_DOUBLES 80
#define FT_LIBM_TAB_INIT {                                                                                                  \
    /* __exp2f_data.tab (bit patterns) */                                                                                   \
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,      \
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,      \
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,      \
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,      \
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,      \
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,      \
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,                                                                           \
    /* invc of __logf_data.tab = __powf_log2_data.tab (bit patterns of the doubles) */                                      \
    0x3ff661ec79f8f3beull, 0x3ff571ed4aaf883dull, 0x3ff49539f0f010b0ull, 0x3ff3c995b0b80385ull, 0x3ff30d190c8864a5ull,      \
    0x3ff25e227b0b8ea0ull, 0x3ff1bb4a4a1a343full, 0x3ff12358f08ae5baull, 0x3ff0953f419900a7ull, 0x3ff0000000000000ull,      \
    0x3fee608cfd9a47acull, 0x3feca4b31f026aa0ull, 0x3feb2036576afce6ull, 0x3fe9c2d163a1aa2dull, 0x3fe886e6037841edull,      \
    0x3fe767dcf5534862ull,                                                                                                  \
    /* logc of __logf_data.tab */                                                                                           \
    0xbfd57bf7808caadeull, 0xbfd2bef0a7c06ddbull, 0xbfd01eae7f513a67ull, 0xbfcb31d8a68224e9ull, 0xbfc6574f0ac07758ull,      \
    0xbfc1aa2bc79c8100ull, 0xbfba4e76ce8c0e5eull, 0xbfb1973c5a611cccull, 0xbfa252f438e10c1eull, 0x0000000000000000ull,      \
    0x3faaa5aa5df25984ull, 0x3fbc5e53aa362eb4ull, 0x3fc526e57720db08ull, 0x3fcbc2860d224770ull, 0x3fd1058bc8a07ee1ull,      \
    0x3fd4043057b6ee09ull,                                                                                                  \
    /* logc of __powf_log2_data.tab */                                                                                      \
    0xbfdefec65b963019ull, 0xbfdb0b6832d4fca4ull, 0xbfd7418b0a1fb77bull, 0xbfd39de91a6dcf7bull, 0xbfd01d9bf3f2b631ull,      \
    0xbfc97c1d1b3b7af0ull, 0xbfc2f9e393af3c9full, 0xbfb960cbbf788d5cull, 0xbfaa6f9db6475fceull, 0x0000000000000000ull,      \
    0x3fb338ca9f24f53dull, 0x3fc476a9543891baull, 0x3fce840b4ac4e4d2ull, 0x3fd40645f0c6651cull, 0x3fd88e9c2c1b9ff8ull,      \
    0x3fdce0a44eb17bccull }

typedef unsigned long long ft_u64;

FT_HD ft_u64 ft_d2u(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (ft_u64)__double_as_longlong(d);
#else
    ft_u64 u; memcpy(&u, &d, 8); return u;
#endif
}
FT_HD double ft_u2d(ft_u64 u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d; memcpy(&d, &u, 8); return d;
#endif
}
FT_HD float ft_u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
// a*b+c: one rounding in the FMA build, two in the SSE2 build
template <bool FMA> FT_HD double ft_mad(double a, double b, double c) { return FMA ? fma(a, b, c) : a * b + c; }

// ---- expf (e_expf.c) --------------------------------------------------------------------------------------------------
// tab: FT_LIBM_TAB (the 2^(i/32) entries are read as 64-bit patterns)
// ft_glibc_expf_main: the function's main path, for callers that have established |x| < 88 (no special case can apply)
template <bool FMA> FT_HD float ft_glibc_expf_main(float x, const ft_u64* tab) {
    const double InvLn2N = 0x1.71547652b82fep+5, SHIFT = 0x1.8p+52;                       // N/ln2 (N = 32); 1.5 * 2^52
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;   // poly_scaled
    const double xd = (double)x;
    // x*N/ln2 = k + r, r in [-1/2, 1/2]; the FMA build fuses the product into both the shifted sum and the remainder
    double kd, r;
    ft_u64 ki;
    if (FMA) {
        kd = fma(InvLn2N, xd, SHIFT);
        ki = ft_d2u(kd);
        kd = kd - SHIFT;
        r = fma(InvLn2N, xd, -kd);
    } else {
        const double z = InvLn2N * xd;
        kd = z + SHIFT;
        ki = ft_d2u(kd);
        kd = kd - SHIFT;
        r = z - kd;
    }
    // exp(x) = 2^(k/N) * 2^(r/N) ~= s * (C0 r^3 + C1 r^2 + C2 r + 1)
    const ft_u64 t = tab[ki & 31u] + (ki << 47);
    const double s = ft_u2d(t);
    const double z2 = ft_mad<FMA>(C0, r, C1);
    const double r2 = r * r;
    double y = ft_mad<FMA>(C2, r, 1.0);
    y = ft_mad<FMA>(z2, r2, y);
    y = y * s;
    return (float)y;
}
template <bool FMA> FT_HD float ft_glibc_expf(float x, const ft_u64* tab) {
    const uint32_t ux = ft_bits(x), abstop = (ux >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                                                                  // |x| >= 88 or x is NaN
        if (ux == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return ft_u2f(0x7f800000u);                                   // __math_oflowf: x > log(2^128)
        if (x < -0x1.9fe368p6f) return 0.0f;                                                 // __math_uflowf: x < log(2^-150)
        if (x < -0x1.9d1d9ep6f) return ft_u2f(1u);                                           // __math_may_uflowf: 0x1.4p-75f squared = 2^-149
    }
    return ft_glibc_expf_main<FMA>(x, tab);
}

// ---- logf (e_logf.c) ----------------------------------------------------------------------------------------------------
template <bool FMA> FT_HD float ft_glibc_logf(float x, const ft_u64* tab) {
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = ft_bits(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {                                     // x < 2^-126, inf or NaN
        if (ix * 2u == 0u) return ft_u2f(0xff800000u);                                       // __math_divzerof(1): -inf
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return ft_u2f(0x7fc00000u);        // __math_invalidf
        ix = ft_bits(x * 0x1p23f);                                                           // subnormal: normalise
        ix -= 23u << 23;
    }
    // x = 2^k z, z in [OFF, 2 OFF); 16 subintervals, c near the centre of each
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const int k = (int)tmp >> 23;                                                            // arithmetic shift
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = ft_u2d(tab[32u + i]), logc = ft_u2d(tab[48u + i]);
    const double z = (double)ft_u2f(iz);
    // log(x) = log1p(z/c - 1) + log(c) + k ln2
    const double r = ft_mad<FMA>(z, invc, -1.0);
    const double y0 = ft_mad<FMA>((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = ft_mad<FMA>(A1, r, A2);
    y = ft_mad<FMA>(A0, r2, y);
    y = ft_mad<FMA>(y, r2, y0 + r);
    return (float)y;
}

// ---- powf (e_powf.c) ----------------------------------------------------------------------------------------------------
FT_HD int ft_glibc_checkint(uint32_t iy) {                         // 0: not an integer, 1: odd, 2: even (iy finite, non-zero)
    const int e = (int)((iy >> 23) & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
FT_HD bool ft_glibc_zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }
FT_HD bool ft_glibc_issignaling(uint32_t ix) { return 2u * (ix ^ 0x00400000u) > 2u * 0x7fc00000u; }

template <bool FMA> FT_HD float ft_glibc_powf(float x, float y, const ft_u64* tab) {
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
                 A4 = 0x1.71547652ab82bp+0;                                                  // __powf_log2_data.poly (POWF_SCALE = 1)
    const double SHIFT = 0x1.8p+47;                                                          // __exp2f_data.shift_scaled = 1.5 * 2^52 / 32
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;     // __exp2f_data.poly
    uint32_t sign_bias = 0;
    uint32_t ix = ft_bits(x);
    const uint32_t iy = ft_bits(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || ft_glibc_zeroinfnan(iy)) {
        if (ft_glibc_zeroinfnan(iy)) {                                                       // y is 0, inf or NaN
            if (2u * iy == 0u) return ft_glibc_issignaling(ix) ? x + y : 1.0f;
            if (ix == 0x3f800000u) return ft_glibc_issignaling(iy) ? x + y : 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;            // |x| < 1 && y = inf, or |x| > 1 && y = -inf
            return y * y;
        }
        if (ft_glibc_zeroinfnan(ix)) {                                                       // x is 0, inf or NaN
            float x2 = x * x;
            if ((ix & 0x80000000u) && ft_glibc_checkint(iy) == 1) { x2 = -x2; sign_bias = 1; }
            if (2u * ix == 0u && (iy & 0x80000000u)) return ft_u2f(sign_bias ? 0xff800000u : 0x7f800000u);   // __math_divzerof
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {                                                              // finite x < 0
            const int yint = ft_glibc_checkint(iy);
            if (yint == 0) return ft_u2f(0x7fc00000u);                                       // __math_invalidf
            if (yint == 1) sign_bias = 1u << 16;                                             // SIGN_BIAS
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                                                              // subnormal x: normalise
            ix = ft_bits(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int)top >> 23;
    const double invc = ft_u2d(tab[32u + i]), logc = ft_u2d(tab[64u + i]);
    const double z = (double)ft_u2f(iz);
    const double r = ft_mad<FMA>(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = ft_mad<FMA>(A0, r, A1);
    const double p = ft_mad<FMA>(A2, r, A3);
    const double r4 = r2 * r2;
    double q = ft_mad<FMA>(A4, r, y0);
    q = ft_mad<FMA>(p, r2, q);
    yy = ft_mad<FMA>(yy, r4, q);
    const double ylogx = (double)y * yy;                                                     // cannot overflow: y is single precision
    if (((ft_d2u(ylogx) >> 47) & 0xffffu) >= 0x80bfu) {                                      // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return ft_u2f(sign_bias ? 0xff800000u : 0x7f800000u);   // __math_oflowf
        if (ylogx <= -150.0) return ft_u2f(sign_bias ? 0x80000000u : 0u);                    // __math_uflowf
        if (ylogx < -149.0) return ft_u2f(sign_bias ? 0x80000001u : 1u);                     // __math_may_uflowf
    }
    // exp2_inline: x = k/N + r, r in [-1/(2N), 1/(2N)]
    double kd = ylogx + SHIFT;
    const ft_u64 ki = ft_d2u(kd);
    kd = kd - SHIFT;
    const double rr = ylogx - kd;
    const ft_u64 t = tab[ki & 31u] + ((ki + sign_bias) << 47);
    const double s = ft_u2d(t);
    const double z2 = ft_mad<FMA>(C0, rr, C1);
    const double rr2 = rr * rr;
    double e = ft_mad<FMA>(C2, rr, 1.0);
    e = ft_mad<FMA>(z2, rr2, e);
    e = e * s;
    return (float)e;
}
