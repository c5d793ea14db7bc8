// ft_math.h — float32 arithmetic of the FrayTracer hot path, usable from host and gfx950 device code.
//
// Every function reproduces the rounding sequence of the System.Numerics / System.MathF call the
// reference makes (src/FrayTracer/Math.fs:26-83); the translation unit must be compiled with
// -ffp-contract=off so that no a*b+c is fused behind our back.  Explicit fmaf() is used only
// inside ft_exp (a fixed algorithm of ours, see DESIGN.md "exp/log").
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FT_HD __host__ __device__ __forceinline__
#else
#define FT_HD inline
#endif

struct f3 { float x, y, z; };

FT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
FT_HD f3 splat3(float s) { return mk3(s, s, s); }
FT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
FT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
FT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
FT_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
FT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
FT_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
FT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
FT_HD f3 neg3(f3 a) { return mk3(0.0f - a.x, 0.0f - a.y, 0.0f - a.z); }   // Vector3 unary minus = Zero - v

// Vector3.Dot: (x*x' + y*y') + z*z'
FT_HD float ft_dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
FT_HD float ft_length2(f3 v) { return ft_dot(v, v); }
FT_HD float ft_length(f3 v) { return sqrtf(ft_dot(v, v)); }
FT_HD float ft_distance(f3 a, f3 b) { f3 d = a - b; return sqrtf(ft_dot(d, d)); }
FT_HD float ft_distance2(f3 a, f3 b) { f3 d = a - b; return ft_dot(d, d); }
FT_HD f3 ft_normalize(f3 v) { return v / ft_length(v); }
FT_HD f3 ft_cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
FT_HD f3 ft_lerp(f3 a, f3 b, float t) { return (a * (1.0f - t)) + (b * t); }
FT_HD f3 ft_vmin(f3 a, f3 b) { return mk3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }
FT_HD f3 ft_vmax(f3 a, f3 b) { return mk3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }

FT_HD uint32_t ft_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
FT_HD bool ft_isnan(float f) { return f != f; }

// MathF.Max / MathF.Min (.NET Core 3.0+): IEEE 754-2019 maximum / minimum — NaN propagates, -0 < +0.
FT_HD float ft_max(float a, float b) {
    if (a != b) { if (!(a != a)) return b < a ? a : b; return a; }
    return (ft_bits(b) >> 31) ? a : b;
}
FT_HD float ft_min(float a, float b) {
    if (a != b) { if (!(a != a)) return a < b ? a : b; return a; }
    return (ft_bits(a) >> 31) ? a : b;
}
FT_HD float ft_clamp01(float x) { return ft_max(0.0f, ft_min(1.0f, x)); }     // Math.fs:51

// `MathF.Floor x |> int` (Math.fs:57): conv.i4 with x86 semantics (out of range / NaN -> INT_MIN)
FT_HD int ft_conv_i4(float f) {
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int)f;
}
FT_HD int ft_floor_i(float x) { return ft_conv_i4(floorf(x)); }
FT_HD int ft_ceiling_i(float x) { return ft_conv_i4(ceilf(x)); }
FT_HD int ft_clamp_i(int lo, int hi, int x) { int m = hi < x ? hi : x; return lo > m ? lo : m; }  // Math.fs:23

// MathF.Sign (Math.fs:40).  .NET throws on NaN; we return 0 (callers flag NaN distances anyway).
FT_HD int ft_sign_i(float x) { return x < 0.0f ? -1 : (x > 0.0f ? 1 : 0); }

// ---------------------------------------------------------------------------------------------
// exp / log for SdfForm.unionSmooth (SdfForm.fs:80,82).  .NET's MathF.Exp/Log are platform libm
// and not bit-reproducible; these are fixed algorithms built from IEEE +,-,*,/ and fma only
// (max error 0.93 ulp for exp, 0.51 ulp for log), so any IEEE machine gives the same bits.
// ---------------------------------------------------------------------------------------------
FT_HD float ft_exp(float x) {
    if (x != x) return x;
    x = x < -104.0f ? -104.0f : (x > 89.0f ? 89.0f : x);
    // n = round-half-even(x * log2e) with ONE rounding: fma onto 1.5*2^23 (ulp 1); the integer n
    // also sits in the low mantissa bits of tm, which the device fast path shifts into the exponent
    const float tm = fmaf(x, 0x1.715476p+0f, 12582912.0f);
    const float n = tm - 12582912.0f;
    float r = fmaf(n, -0x1.62e4p-1f, x);               // - n*ln2_hi (exact product)
    r = fmaf(n, -0x1.7f7d1cp-20f, r);                  // - n*ln2_lo
    float p = 0x1.6d7538p-10f;                                // Horner, degree 6: 1 + r(1 + r(c2 + ... + c6 r^4))
    p = fmaf(p, r, 0x1.120b72p-7f);
    p = fmaf(p, r, 0x1.5554b8p-5f);
    p = fmaf(p, r, 0x1.5554dcp-3f);
    p = fmaf(p, r, 0x1.0p-1f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return ldexpf(p, (int)(ft_bits(tm) - 0x4B400000u));
}

FT_HD double ft_log_f64(double x) {                    // fdlibm e_log.c structure; x finite, > 0, normal
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    uint64_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = (uint64_t)__double_as_longlong(x);
#else
    memcpy(&u, &x, 8);
#endif
    int k = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
#if defined(__HIP_DEVICE_COMPILE__)
    m = __longlong_as_double((long long)u);
#else
    memcpy(&m, &u, 8);
#endif
    if (m > 1.4142135623730951) { m = m * 0.5; k += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}
FT_HD float ft_log(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    return (float)ft_log_f64((double)x);
}
