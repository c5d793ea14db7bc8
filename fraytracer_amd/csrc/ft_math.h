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

// ---------------------------------------------------------------------------------------------
// pow for the tone map (FColor.gammaInverse, FColor.fs:50-55: MathF.Pow per channel).  MathF.Pow is
// platform libm in .NET; this is a fixed algorithm from IEEE double + - * / only: exp(g * log(x)) in
// double (fdlibm e_log.c / e_exp.c structure), rounded to float once.  Its double-precision error
// (~1e-14 relative) is far below half a float ulp, so it returns the correctly rounded powf except
// within ~1e-7 relative probability of a rounding boundary.
// ---------------------------------------------------------------------------------------------
FT_HD double ft_exp_f64(double x) {                    // |x| <= 150
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const int k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    const double hi = x - (double)k * ln2HI, lo = (double)k * ln2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    uint64_t u;                                        // y in [0.7, 1.5), |k| <= 217: scale by adding k to the exponent field
#if defined(__HIP_DEVICE_COMPILE__)
    u = (uint64_t)__double_as_longlong(y);
#else
    memcpy(&u, &y, 8);
#endif
    u += (uint64_t)((int64_t)k << 52);
    double out;
#if defined(__HIP_DEVICE_COMPILE__)
    out = __longlong_as_double((long long)u);
#else
    memcpy(&out, &u, 8);
#endif
    return out;
}

FT_HD float ft_pow(float x, float g) {
    if (g == 0.0f) return 1.0f;
    if (x != x || g != g) return NAN;
    if (g == 1.0f) return x;
    if (x < 0.0f) return NAN;                          // (MathF.Pow of a negative base and a non-integer exponent)
    if (x == 0.0f) return g > 0.0f ? 0.0f : INFINITY;
    if (x == INFINITY) return g > 0.0f ? INFINITY : 0.0f;
    if (x == 1.0f) return 1.0f;
    if (g == INFINITY || g == -INFINITY) return ((x < 1.0f) == (g > 0.0f)) ? 0.0f : INFINITY;
    double y = (double)g * ft_log_f64((double)x);
    y = y < -150.0 ? -150.0 : (y > 150.0 ? 150.0 : y);
    return (float)ft_exp_f64(y);
}

// counter-based dither for the tone map: lowbias32 of (pixel, channel, seed) -> [0, 1) with 24 bits.  The reference
// draws rng.range_01() from ONE System.Random shared by a parallel map (Image.fs:46-49): racy, not reproducible.
FT_HD float ft_dither_u(uint32_t x, uint32_t y, uint32_t channel, uint32_t seed) {
    uint32_t h = seed ^ (x * 0x9E3779B1u) ^ (y * 0x85EBCA77u) ^ (channel * 0xC2B2AE3Du);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// FColor.toColor (FColor.fs:43-48): c * 254.5f + u |> MathF.Round (half to even) |> int |> min 255.  A NaN or negative value,
// where the reference's Color.FromArgb would throw, gives 0.
FT_HD uint32_t ft_to_byte(float c, float u) {
    const float v = c * 254.5f + u;
    if (!(v >= 0.0f)) return 0u;
    const float r = rintf(v);
    return r >= 255.0f ? 255u : (uint32_t)r;
}
