// ft_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A deliberately literal float32 restatement, written from the F# source text,
// of the FrayTracer per-pixel hot path (Image.render -> SdfScene.trace ->
// SdfObject.tryTrace -> SdfForm.tryTrace / normal / lights).  Every function
// cites the reference file:line it follows (paths relative to the reference
// repository root, src/FrayTracer/...).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.  The product (fraytracer_amd/, libfraytracer_hip.so)
// never links, imports or calls anything in this directory.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures,
// and no .NET/F# toolchain exists in the build container, so this restatement
// cannot be checked against output of the real F# program.  It is pinned by
// hand-derived known answers (tests/test_oracle_kat.py) and encodes the
// System.Numerics / System.MathF semantics listed as assumptions in DESIGN.md
// (dot = (xx+yy)+zz, true division, correctly rounded sqrt, NaN-propagating
// Min/Max).  MathF.Exp / MathF.Log are platform libm in .NET and are not bit
// reproducible anywhere; they are replaced by the fixed algorithms orc_expf /
// orc_logf below (libm is selectable with orc_set_libm(1) to measure how many
// pixels that substitution can move).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fno-fast-math).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// System.Numerics.Vector3 as used by the reference (Math.fs:62-83 wrappers).
// Assumed .NET 6 semantics: component-wise ops, Dot = (x*x' + y*y') + z*z',
// Length = sqrt(Dot(v,v)), Normalize = v / Length, vector / scalar = true
// per-component division, unary minus = Zero - v.
// ---------------------------------------------------------------------------
struct V3 { float X, Y, Z; };
struct V2 { float X, Y; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3s(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.X + b.X, a.Y + b.Y, a.Z + b.Z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.X - b.X, a.Y - b.Y, a.Z - b.Z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.X * b.X, a.Y * b.Y, a.Z * b.Z); }
inline V3 operator/(V3 a, V3 b) { return v3(a.X / b.X, a.Y / b.Y, a.Z / b.Z); }
inline V3 operator*(V3 a, float s) { return a * v3s(s); }
inline V3 operator*(float s, V3 a) { return v3s(s) * a; }
inline V3 operator/(V3 a, float s) { return a / v3s(s); }
inline V3 operator-(V3 a) { return v3s(0.0f) - a; }

inline float Dot(V3 a, V3 b) { return (a.X * b.X + a.Y * b.Y) + a.Z * b.Z; }
inline float LengthSquared(V3 v) { return Dot(v, v); }
inline float Length(V3 v) { return sqrtf(Dot(v, v)); }
inline float Distance(V3 a, V3 b) { V3 d = a - b; return sqrtf(Dot(d, d)); }
inline float DistanceSquared(V3 a, V3 b) { V3 d = a - b; return Dot(d, d); }
inline V3 Normalize(V3 v) { return v / Length(v); }
inline V3 Cross(V3 a, V3 b) {
    return v3(a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X);
}
inline V3 Lerp(V3 a, V3 b, float t) { return (a * (1.0f - t)) + (b * t); }
inline V3 VMin(V3 a, V3 b) {
    return v3(a.X < b.X ? a.X : b.X, a.Y < b.Y ? a.Y : b.Y, a.Z < b.Z ? a.Z : b.Z);
}
inline V3 VMax(V3 a, V3 b) {
    return v3(a.X > b.X ? a.X : b.X, a.Y > b.Y ? a.Y : b.Y, a.Z > b.Z ? a.Z : b.Z);
}
inline float V2Length(V2 v) { return sqrtf(v.X * v.X + v.Y * v.Y); }

// ---------------------------------------------------------------------------
// System.MathF (Math.fs:26-59).  Min/Max are the .NET Core 3.0+ IEEE-754:2019
// minimum/maximum: NaN-propagating, -0 < +0.
// ---------------------------------------------------------------------------
inline bool IsNegative(float f) { uint32_t u; memcpy(&u, &f, 4); return (u >> 31) != 0; }
inline float MathF_Max(float a, float b) {
    if (a != b) { if (!(a != a)) return b < a ? a : b; return a; }
    return IsNegative(b) ? a : b;
}
inline float MathF_Min(float a, float b) {
    if (a != b) { if (!(a != a)) return a < b ? a : b; return a; }
    return IsNegative(a) ? a : b;
}
// Math.fs:46,48  `MathF.min m x = MathF.Min(m, x)`, `MathF.max m x = MathF.Max(m, x)`
inline float fs_min(float m, float x) { return MathF_Min(m, x); }
inline float fs_max(float m, float x) { return MathF_Max(m, x); }
// Math.fs:51
inline float clamp01(float x) { return MathF_Max(0.0f, MathF_Min(1.0f, x)); }

thread_local uint32_t tl_flags = 0;   // bit0: NaN distance in a march, bit1: MathF.Sign(NaN), bit2: step cap
// Math.fs:40  MathF.Sign -> int.  .NET throws ArithmeticException on NaN; the
// oracle records a flag and returns 0 instead of aborting the render.
inline int sign_i(float x) {
    if (x != x) { tl_flags |= 2u; return 0; }
    return x < 0.0f ? -1 : (x > 0.0f ? 1 : 0);
}
// Math.fs:57,59  MathF.Floor/Ceiling |> int  (conv.i4: x86 cvttss2si semantics,
// out-of-range and NaN give INT_MIN)
inline int conv_i4(float f) {
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int)f;
}
inline int floor_i(float x) { return conv_i4(floorf(x)); }
inline int ceiling_i(float x) { return conv_i4(ceilf(x)); }
inline int clamp_i(int lo, int hi, int x) { return std::max(lo, std::min(hi, x)); }  // Math.fs:23

// ---------------------------------------------------------------------------
// exp / log used by SdfForm.unionSmooth (SdfForm.fs:80,82).  The reference
// calls MathF.Exp / MathF.Log = platform libm.  These fixed algorithms use only
// IEEE +,-,*,/,fma, rint and exact scaling, so a GPU can reproduce them bit
// for bit; the product carries its own copy (fraytracer_amd/csrc/ft_math.h) and
// tests/test_math_parity.py compares the two.
// ---------------------------------------------------------------------------
bool g_use_libm = false;

__attribute__((target_clones("fma", "default")))
float orc_expf_impl(float x) {
    // exp(x) = 2^n * e^r, n = rint(x*log2e) (fused), r = x - n*ln2 (two-term), e^r by a degree-6 Horner
    // polynomial with leading coefficients 1, 1 (six fmas) fitted on |r| <= ln2/2.  Max error 0.93 ulp;
    // 94 % of results are the correctly rounded value (measured against double exp, DESIGN.md "exp/log").
    if (x != x) return x;
    x = x < -104.0f ? -104.0f : (x > 89.0f ? 89.0f : x);             // below: rounds to 0; above: overflows to +inf
    // n = round-half-even(x*log2e) in ONE rounding: fma onto 1.5*2^23, whose ulp is 1
    const float tm = __builtin_fmaf(x, 0x1.715476p+0f, 12582912.0f);
    const float n = tm - 12582912.0f;
    float r = __builtin_fmaf(n, -0x1.62e4p-1f, x);                   // ln2 hi (n*hi exact)
    r = __builtin_fmaf(n, -0x1.7f7d1cp-20f, r);                      // ln2 lo
    float p = 0x1.6d7538p-10f;                                            // Horner, degree 6: 1 + r(1 + r(c2 + ... + c6 r^4))
    p = __builtin_fmaf(p, r, 0x1.120b72p-7f);
    p = __builtin_fmaf(p, r, 0x1.5554b8p-5f);
    p = __builtin_fmaf(p, r, 0x1.5554dcp-3f);
    p = __builtin_fmaf(p, r, 0x1.0p-1f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return ldexpf(p, (int)n);                                        // exact scaling; one rounding if subnormal
}

// fdlibm-style log evaluated in double with +,-,*,/ only, then rounded once.
double orc_log_double(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    uint64_t u; memcpy(&u, &x, 8);
    int k = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;        // m in [1,2)
    double m; memcpy(&m, &u, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; k += 1; }            // m in (sqrt2/2, sqrt2]
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
float orc_logf_impl(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    return (float)orc_log_double((double)x);                         // float subnormals are double normals
}
inline float fs_exp(float x) { return g_use_libm ? expf(x) : orc_expf_impl(x); }
inline float fs_log(float x) { return g_use_libm ? logf(x) : orc_logf_impl(x); }

// ---------------------------------------------------------------------------
// Counters (not in the reference; used for the roofline's algorithmic flops).
// ---------------------------------------------------------------------------
enum { PRIM_SPHERE = 0, PRIM_CAPSULE, PRIM_TORUS, PRIM_TRIANGLE, PRIM_BOX, PRIM_NTYPES };
struct Counters {
    uint64_t prim[8];          // primitive Distance calls by type
    uint64_t root_evals;       // scene-object Distance calls issued by march / normal
    uint64_t march_steps;      // Ray.move calls
    uint64_t rays_primary, rays_shadow, rays_ext;
    uint64_t hits_primary, hits_shadow;
    uint64_t smooth_children;  // exp() calls in unionSmooth
    uint64_t union_candidates; // candidates visited (both tests evaluated) in union loops
    uint64_t flags;
    uint64_t union_tested;     // of those, the candidates up to and including the first one whose :30 test fails — all that a walk needs to
                               // look at which knows that the list is sorted by LowerBound (measurement only: what the GPU walk executes)
};
thread_local Counters tl_cnt;

// ---------------------------------------------------------------------------
// Types.fs:9-79
// ---------------------------------------------------------------------------
struct Ray { V3 Origin; V3 Direction; float Length; float Epsilon; };       // Types.fs:9-17
struct SdfBoundary { V3 Center; float Radius; };                            // Types.fs:19-24
struct SdfFormTraceResult { Ray ray; float Distance; };                     // Types.fs:32-37
struct FColor { V3 c; };                                                    // FColor.fs:8-10
inline FColor operator+(FColor a, FColor b) { return FColor{a.c + b.c}; }   // FColor.fs:15-17
inline FColor operator*(FColor a, FColor b) { return FColor{a.c * b.c}; }   // FColor.fs:19-21
inline FColor operator*(FColor a, float s) { return FColor{a.c * s}; }      // FColor.fs:23-25
inline FColor operator/(FColor a, float s) { return FColor{a.c / s}; }      // FColor.fs:27-29

struct GridDump;
struct SdfForm {                                                            // Types.fs:40-44
    std::function<float(V3)> Distance;
    SdfBoundary Boundary;
    std::shared_ptr<GridDump> grid;   // oracle-only: lets tests inspect the union's lookup
};
struct GlassParams { float Ior, Dispersion; V3 Tint; };                     // EXTENSION (not in the reference)
struct SdfMaterial {                                                        // Types.fs:46-49
    std::function<FColor(V3, V3)> Color;
    std::function<const GlassParams*(V3)> Glass;   // EXTENSION: empty / nullptr = an ordinary (solid) material
};
struct SdfObject { SdfForm Form; SdfMaterial Material; };                   // Types.fs:51-55
struct SdfObjectTraceResult { Ray ray; V3 Normal; FColor Color; };          // Types.fs:57-65
struct SdfLight {                                                           // Types.fs:67-72
    std::function<V3(V3)> Direction;
    std::function<bool(const SdfObject&, const Ray&, FColor&)> Intensity;   // voption -> bool + out
};
struct SdfScene { SdfObject Object; FColor BackgroundColor; std::vector<SdfLight> Lights; };  // Types.fs:74-79

// Ray.fs:6-13
inline V3 Ray_get(float length, const Ray& ray) { return ray.Origin + ray.Direction * length; }
inline Ray Ray_move(float length, const Ray& ray) {
    Ray r = ray;
    r.Origin = Ray_get(length, ray);
    r.Length = ray.Length - length;
    return r;
}

// ---------------------------------------------------------------------------
// SdfBoundary.fs:7-67
// ---------------------------------------------------------------------------
SdfBoundary Boundary_union(SdfBoundary a, SdfBoundary b) {                  // SdfBoundary.fs:7-22
    V3 diff = b.Center - a.Center;
    float distance = Length(diff);
    if (distance + b.Radius <= a.Radius) return a;
    else if (distance + a.Radius <= b.Radius) return b;
    else {
        V3 dir = diff / distance;
        V3 a_ = a.Center - dir * a.Radius;
        V3 b_ = b.Center + dir * b.Radius;
        return SdfBoundary{(a_ + b_) * 0.5f, Distance(a_, b_) * 0.5f};
    }
}
SdfBoundary Boundary_intersection(SdfBoundary a, SdfBoundary b) {           // SdfBoundary.fs:29-49
    V3 diff = b.Center - a.Center;
    float distance = Length(diff);
    if (distance + b.Radius <= a.Radius) return b;
    else if (distance + a.Radius <= b.Radius) return a;
    else {
        V3 dir = diff / distance;
        V3 a_ = a.Center + dir * a.Radius;
        V3 b_ = b.Center - dir * b.Radius;
        float d2 = distance * distance;
        float aR2 = a.Radius * a.Radius;
        float bR2 = b.Radius * b.Radius;
        // SdfBoundary.fs:48 — the missing square on (d2 - bR2 + aR2) is the reference's, kept.
        float radius = sqrtf(4.0f * d2 * aR2 - (d2 - bR2 + aR2)) / (2.0f * distance);
        return SdfBoundary{(a_ + b_) * 0.5f, radius};
    }
}
template <class F> SdfBoundary reduceBoundaries(const std::vector<SdfBoundary>& bs, F f) {  // Seq.reduce
    SdfBoundary acc = bs[0];
    for (size_t i = 1; i < bs.size(); ++i) acc = f(acc, bs[i]);
    return acc;
}
inline float getMinDistance(SdfBoundary x, V3 p) { return Distance(x.Center, p) - x.Radius; }  // :62
inline float getMaxDistance(SdfBoundary x, V3 p) { return Distance(x.Center, p) + x.Radius; }  // :63
inline V3 AABB_getMin(SdfBoundary b) { return b.Center - v3s(b.Radius); }                      // :66
inline V3 AABB_getMax(SdfBoundary b) { return b.Center + v3s(b.Radius); }                      // :67

// ---------------------------------------------------------------------------
// SdfBoundary.buildSpatialLookup  (SdfBoundary.fs:211-282)
// ---------------------------------------------------------------------------
struct LookupItem { float LowerBound; int Item; };                           // SdfBoundary.fs:211-216
struct LookupCell { V3 Center; std::vector<LookupItem> Items; };             // SdfBoundary.fs:219-223
struct GridDump {
    V3 aabbMin, cellSize, cellSizeInv;
    int countX, countY, countZ;
    std::vector<LookupCell> cells;   // [x,y,z] -> (x*countY + y)*countZ + z
    const LookupCell& lookup(V3 position) const {                            // SdfBoundary.fs:276-282
        V3 c = (position - aabbMin) * cellSizeInv;
        int ix = clamp_i(0, countX - 1, floor_i(c.X));
        int iy = clamp_i(0, countY - 1, floor_i(c.Y));
        int iz = clamp_i(0, countZ - 1, floor_i(c.Z));
        return cells[((size_t)ix * countY + iy) * countZ + iz];
    }
};

std::shared_ptr<GridDump> buildSpatialLookup(const std::vector<SdfBoundary>& bounds) {
    auto g = std::make_shared<GridDump>();
    const size_t n = bounds.size();
    V3 aabbMin = AABB_getMin(bounds[0]);                                     // :229 Seq.reduce Vector3.min
    V3 aabbMax = AABB_getMax(bounds[0]);                                     // :230
    float sum = bounds[0].Radius;                                            // :232 Seq.reduce (+)
    for (size_t i = 1; i < n; ++i) {
        aabbMin = VMin(aabbMin, AABB_getMin(bounds[i]));
        aabbMax = VMax(aabbMax, AABB_getMax(bounds[i]));
        sum = sum + bounds[i].Radius;
    }
    float countSize = 1.5f * (sum / (float)(int)n);                          // :233
    V3 aabbSize = aabbMax - aabbMin;                                         // :235
    int countX = std::max(1, ceiling_i(aabbSize.X / countSize));             // :237
    int countY = std::max(1, ceiling_i(aabbSize.X / countSize));             // :238 (X, as in the reference)
    int countZ = std::max(1, ceiling_i(aabbSize.X / countSize));             // :239 (X, as in the reference)
    V3 cellSize = aabbSize / v3((float)countX, (float)countY, (float)countZ);// :240
    V3 cellSizeInv = v3s(1.0f) / cellSize;                                   // :241
    g->aabbMin = aabbMin; g->cellSize = cellSize; g->cellSizeInv = cellSizeInv;
    g->countX = countX; g->countY = countY; g->countZ = countZ;
    g->cells.resize((size_t)countX * countY * countZ);
    for (int x = 0; x < countX; ++x)
    for (int y = 0; y < countY; ++y)
    for (int z = 0; z < countZ; ++z) {
        V3 center = aabbMin + cellSize * 0.5f + cellSize * v3((float)x, (float)y, (float)z);   // :246
        float m = getMaxDistance(bounds[0], center);                         // :249-252 Seq.min
        for (size_t i = 1; i < n; ++i) { float v = getMaxDistance(bounds[i], center); if (v < m) m = v; }
        float upperBound = m + Length(cellSize * 0.5f);                      // :253
        LookupCell& cell = g->cells[((size_t)x * countY + y) * countZ + z];
        cell.Center = center;
        for (size_t i = 0; i < n; ++i) {                                     // :255-264
            float minDistance = getMinDistance(bounds[i], center);
            if (minDistance < upperBound) cell.Items.push_back(LookupItem{minDistance, (int)i});
        }
        // :267-268 Array.sortInPlaceBy is unstable in .NET; ties are broken here by input order.
        std::stable_sort(cell.Items.begin(), cell.Items.end(),
                         [](const LookupItem& a, const LookupItem& b) { return a.LowerBound < b.LowerBound; });
    }
    return g;
}

// ---------------------------------------------------------------------------
// SdfForm.fs:14-91  combinators
// ---------------------------------------------------------------------------
SdfForm Form_union(const std::vector<SdfForm>& forms, std::string& err) {    // SdfForm.fs:14-40
    if (forms.empty()) { err = "No SdfObjects given."; return SdfForm{}; }
    if (forms.size() == 1) return forms[0];
    std::vector<SdfBoundary> bs; for (auto& f : forms) bs.push_back(f.Boundary);
    auto grid = buildSpatialLookup(bs);
    for (auto& c : grid->cells) if (c.Items.empty()) { err = "union: empty lookup cell (reference would throw)"; return SdfForm{}; }
    auto items = std::make_shared<std::vector<SdfForm>>(forms);
    SdfForm out;
    out.grid = grid;
    out.Distance = [grid, items](V3 position) -> float {
        const LookupCell& cell = grid->lookup(position);                     // :23
        float distanceToCenter = Distance(cell.Center, position);            // :25
        float min = (*items)[cell.Items[0].Item].Distance(position);         // :26
        bool sortedWalkDone = false;                                         // (counter only)
        for (size_t i = 1; i < cell.Items.size(); ++i) {                     // :27
            const LookupItem& sdf = cell.Items[i];
            const SdfForm& item = (*items)[sdf.Item];
            tl_cnt.union_candidates++;
            if (!sortedWalkDone) { tl_cnt.union_tested++; if (!(min > sdf.LowerBound - distanceToCenter)) sortedWalkDone = true; }
            if (min > sdf.LowerBound - distanceToCenter                      // :30
                && min > getMinDistance(item.Boundary, position)) {          // :31
                min = fs_min(min, item.Distance(position));                  // :33
            }
        }
        return min;
    };
    out.Boundary = reduceBoundaries(bs, Boundary_union);                     // :36-39
    return out;
}

SdfForm Form_subtract(const SdfForm& a, const SdfForm& b) {                  // SdfForm.fs:42-49
    SdfForm out;
    out.Distance = [a, b](V3 position) -> float {
        float da = a.Distance(position);
        return fs_max(-(b.Distance(position)), da);                          // :46-47
    };
    out.Boundary = a.Boundary;
    return out;
}

SdfForm Form_intersect(const std::vector<SdfForm>& forms, std::string& err) {// SdfForm.fs:51-67
    if (forms.empty()) { err = "No SdfObjects given."; return SdfForm{}; }
    if (forms.size() == 1) return forms[0];
    auto items = std::make_shared<std::vector<SdfForm>>(forms);
    SdfForm out;
    out.Distance = [items](V3 position) -> float {
        float max = (*items)[0].Distance(position);                          // :59
        for (size_t i = 1; i < items->size(); ++i) {                         // :60
            const SdfForm& obj = (*items)[i];
            if (max < getMaxDistance(obj.Boundary, position))                // :62
                max = fs_max(max, obj.Distance(position));                   // :63
        }
        return max;
    };
    std::vector<SdfBoundary> bs; for (auto& f : forms) bs.push_back(f.Boundary);
    out.Boundary = reduceBoundaries(bs, Boundary_intersection);              // :66
    return out;
}

SdfForm Form_unionSmooth(float strength, const std::vector<SdfForm>& forms, std::string& err) {  // SdfForm.fs:69-91
    if (forms.empty()) { err = "blub"; return SdfForm{}; }
    if (forms.size() == 1) return forms[0];
    auto sdfs = std::make_shared<std::vector<SdfForm>>(forms);
    float strengthInverse = -1.0f / strength;                                // :75
    SdfForm out;
    out.Distance = [sdfs, strengthInverse, strength](V3 position) -> float {
        float sum = 0.0f;                                                    // :77
        for (size_t i = 0; i < sdfs->size(); ++i) {                          // :78
            float distance = (*sdfs)[i].Distance(position);                  // :79
            tl_cnt.smooth_children++;
            sum = sum + fs_exp(strengthInverse * distance);                  // :80
        }
        return -fs_log(sum) * strength;                                      // :82
    };
    std::vector<SdfBoundary> bs; for (auto& f : forms) bs.push_back(f.Boundary);
    out.Boundary = reduceBoundaries(bs, Boundary_union);                     // :87-90
    return out;
}

// ---------------------------------------------------------------------------
// SdfForm.fs:93-115  march loop and normal
// ---------------------------------------------------------------------------
const uint64_t STEP_CAP = 1u << 20;   // not in the reference (it has no cap); see DESIGN.md "NaN / step cap"

bool Form_tryTrace(const SdfForm& sdf, Ray ray, SdfFormTraceResult& out) {   // SdfForm.fs:93-104
    uint64_t steps = 0;
    for (;;) {                                                               // tail call -> loop (fsproj:7)
        if (ray.Length <= 0.0f) return false;                                // :94
        float distance = sdf.Distance(ray.Origin);                           // :97
        tl_cnt.root_evals++;
        if (distance != distance) { tl_flags |= 1u; return false; }          // reference would spin forever
        if (distance < ray.Epsilon) { out.ray = ray; out.Distance = distance; return true; }  // :98-102
        ray = Ray_move(distance, ray);                                       // :104
        tl_cnt.march_steps++;
        if (++steps >= STEP_CAP) { tl_flags |= 4u; return false; }
    }
}

V3 Form_normal(const SdfForm& sdf, float epsilon, V3 position) {             // SdfForm.fs:106-112
    float dx = sdf.Distance(v3(position.X + epsilon, position.Y, position.Z));
    float dy = sdf.Distance(v3(position.X, position.Y + epsilon, position.Z));
    float dz = sdf.Distance(v3(position.X, position.Y, position.Z + epsilon));
    float dc = sdf.Distance(position);
    tl_cnt.root_evals += 4;
    return Normalize(v3(dx, dy, dz) - v3s(dc));
}
inline V3 Form_normalFromRay(const SdfForm& sdf, const Ray& ray) {           // SdfForm.fs:114-115
    return Form_normal(sdf, ray.Epsilon * 0.125f, Ray_get(-ray.Epsilon, ray));
}

// ---------------------------------------------------------------------------
// SdfForm.fs:117-268  primitives
// ---------------------------------------------------------------------------
SdfForm Prim_sphere(V3 Center, float Radius) {                               // SdfForm.fs:125-135
    SdfForm out;
    out.Distance = [Center, Radius](V3 position) -> float {
        tl_cnt.prim[PRIM_SPHERE]++;
        return Distance(Center, position) - Radius;                          // :129
    };
    out.Boundary = SdfBoundary{Center, Radius};
    return out;
}

SdfForm Prim_capsule(V3 From, V3 To, float Radius) {                         // SdfForm.fs:145-170
    V3 dir = To - From;                                                      // :148
    V3 dirInv = dir / LengthSquared(dir);                                    // :149 (Math.fs:69)
    SdfForm out;
    out.Distance = [From, Radius, dir, dirInv](V3 position) -> float {
        tl_cnt.prim[PRIM_CAPSULE]++;
        V3 diff = position - From;                                           // :153
        float t = Dot(diff, dirInv);                                         // :154
        float distance;
        if (t <= 0.0f) distance = Length(diff);                              // :155-156
        else if (t >= 1.0f) distance = Distance(diff, dir);                  // :157-158
        else distance = Distance(diff, dir * t);                             // :160
        return distance - Radius;                                            // :164
    };
    out.Boundary = SdfBoundary{Lerp(From, To, 0.5f), Radius + Distance(From, To) * 0.5f};  // :166-169
    return out;
}

SdfForm Prim_torus(V3 Center, V3 NormalIn, float MajorRadius, float MinorRadius) {  // SdfForm.fs:181-203
    V3 Normal = Normalize(NormalIn);                                         // :182-185
    float planeD = -(Dot(Center, Normal));                                   // :188
    SdfForm out;
    out.Distance = [Center, Normal, MajorRadius, MinorRadius, planeD](V3 position) -> float {
        tl_cnt.prim[PRIM_TORUS]++;
        float distanceToPlane = Dot(position, Normal) + planeD;              // :190
        float distanceToCenter = Distance(Center, position - (distanceToPlane * Normal));  // :191
        float distanceToCircle = distanceToCenter - MajorRadius;             // :192
        return V2Length(V2{distanceToPlane, distanceToCircle}) - MinorRadius;// :194
    };
    out.Boundary = SdfBoundary{Center, MajorRadius + MinorRadius};           // :199-202
    return out;
}

SdfForm Prim_triangle(V3 V1, V3 V2_, V3 V3_, float Radius) {                 // SdfForm.fs:214-268
    V3 v21 = V2_ - V1;               V3 v21i = v21 / LengthSquared(v21);     // :216-217
    V3 v32 = V3_ - V2_;              V3 v32i = v32 / LengthSquared(v32);     // :218-219
    V3 v13 = V1 - V3_;               V3 v13i = v13 / LengthSquared(v13);     // :220-221
    V3 nor = Normalize(Cross(v21, v13));                                     // :222
    V3 n21 = Normalize(Cross(v21, nor));                                     // :223
    V3 n32 = Normalize(Cross(v32, nor));                                     // :224
    V3 n13 = Normalize(Cross(v13, nor));                                     // :225
    SdfForm out;
    out.Distance = [=](V3 position) -> float {
        tl_cnt.prim[PRIM_TRIANGLE]++;
        V3 p1 = position - V1;                                               // :228
        V3 p2 = position - V2_;
        V3 p3 = position - V3_;
        float distance;
        if ((sign_i(Dot(n21, p1)) + sign_i(Dot(n32, p2)) + sign_i(Dot(n13, p3))) < 2) {  // :235-237
            float d21 = DistanceSquared(p1, v21 * clamp01(Dot(v21i, p1)));  // :240
            float d32 = DistanceSquared(p2, v32 * clamp01(Dot(v32i, p2)));  // :241
            float d13 = DistanceSquared(p3, v13 * clamp01(Dot(v13i, p3)));  // :242
            distance = sqrtf(fs_min(d13, fs_min(d32, d21)));                 // :243-244
        } else {
            distance = fabsf(Dot(nor, p1));                                  // :247-248
        }
        return distance - Radius;                                            // :250
    };
    {                                                                        // :252-263
        float areaInv = 0.5f / LengthSquared(Cross(V1 - V2_, V2_ - V3_));
        float w1 = LengthSquared(V2_ - V3_) * Dot(V1 - V2_, V1 - V3_) * areaInv;
        float w2 = LengthSquared(V1 - V3_) * Dot(V2_ - V1, V2_ - V3_) * areaInv;
        float w3 = 1.0f - w1 - w2;
        V3 center = w1 * V1 + w2 * V2_ + w3 * V3_;
        float radius = Length(v21) * Length(v32) * Length(v13) / 2.0f / Length(Cross(v21, v32)) + Radius;
        out.Boundary = SdfBoundary{center, radius};
    }
    return out;
}

// EXT (not in the reference): axis-aligned box, exact SDF, bounding sphere = half diagonal.
// Defined here so the product's extension has a checker; carries no parity claim.
SdfForm Prim_box(V3 Center, V3 Half) {
    SdfForm out;
    out.Distance = [Center, Half](V3 position) -> float {
        tl_cnt.prim[PRIM_BOX]++;
        V3 d = position - Center;
        V3 q = v3(fabsf(d.X) - Half.X, fabsf(d.Y) - Half.Y, fabsf(d.Z) - Half.Z);
        V3 qp = v3(MathF_Max(q.X, 0.0f), MathF_Max(q.Y, 0.0f), MathF_Max(q.Z, 0.0f));
        float outside = Length(qp);
        float inside = MathF_Min(MathF_Max(q.X, MathF_Max(q.Y, q.Z)), 0.0f);
        return outside + inside;
    };
    out.Boundary = SdfBoundary{Center, Length(Half)};
    return out;
}

// ---------------------------------------------------------------------------
// SdfMaterial.fs:4-10, SdfObject.fs:6-78
// ---------------------------------------------------------------------------
SdfMaterial Material_createSolid(FColor color) {                             // SdfMaterial.fs:4-7
    return SdfMaterial{[color](V3, V3) { return color; }, nullptr};
}
// EXTENSION: glass.  Shades as createSolid(tint) wherever bounces are off (max_bounces = 0).
SdfMaterial Material_createGlass(FColor tint, float ior, float dispersion) {
    auto g = std::make_shared<GlassParams>(GlassParams{ior, dispersion, tint.c});
    return SdfMaterial{[tint](V3, V3) { return tint; }, [g](V3) -> const GlassParams* { return g.get(); }};
}
SdfObject Object_create(const SdfMaterial& material, const SdfForm& form) {  // SdfObject.fs:6-10
    return SdfObject{form, material};
}
SdfObject Object_union(const std::vector<SdfObject>& objects, std::string& err) {  // SdfObject.fs:12-48
    if (objects.empty()) { err = "No SdfObjects given."; return SdfObject{}; }
    if (objects.size() == 1) return objects[0];
    std::vector<SdfForm> forms; for (auto& o : objects) forms.push_back(o.Form);
    SdfObject out;
    out.Form = Form_union(forms, err);                                       // :16-19
    if (!err.empty()) return out;
    std::vector<SdfBoundary> bs; for (auto& o : objects) bs.push_back(o.Form.Boundary);
    auto grid = buildSpatialLookup(bs);                                      // :26 (second, identical grid)
    auto objs = std::make_shared<std::vector<SdfObject>>(objects);
    // the argmin of SdfObject.fs:27-46, shared by Color and by the EXTENSION hook Glass
    auto pick = [grid, objs](V3 position) -> const SdfObject* {
        const LookupCell& cell = grid->lookup(position);                     // :28
        const SdfObject* material = &(*objs)[cell.Items[0].Item];            // :29
        float min = (*objs)[cell.Items[0].Item].Form.Distance(position);     // :30
        float distanceToCenter = Distance(cell.Center, position);            // :32
        for (size_t i = 0; i < cell.Items.size(); ++i) {                     // :34 (from 0)
            const LookupItem& sdf = cell.Items[i];
            const SdfObject& item = (*objs)[sdf.Item];
            if (min > sdf.LowerBound - distanceToCenter                      // :37
                && min > getMinDistance(item.Form.Boundary, position)) {     // :38
                float distance = item.Form.Distance(position);               // :40
                if (distance < min) { min = distance; material = &item; }    // :41-43
            }
        }
        return material;
    };
    out.Material.Color = [pick](V3 position, V3 normal) -> FColor {
        return pick(position)->Material.Color(position, normal);             // :45-46
    };
    out.Material.Glass = [pick](V3 position) -> const GlassParams* {         // EXTENSION
        const SdfObject* o = pick(position);
        return o->Material.Glass ? o->Material.Glass(position) : nullptr;
    };
    return out;
}
SdfObject Object_subtract(const SdfObject& object, const SdfForm& form) {    // SdfObject.fs:50-54
    return SdfObject{Form_subtract(object.Form, form), object.Material};
}
SdfObject Object_intersect(const SdfObject& object, const std::vector<SdfForm>& forms, std::string& err) {  // :56-64
    std::vector<SdfForm> all; all.push_back(object.Form);
    for (auto& f : forms) all.push_back(f);
    return SdfObject{Form_intersect(all, err), object.Material};
}
bool Object_tryTrace(const SdfObject& object, const Ray& ray, SdfObjectTraceResult& out) {  // SdfObject.fs:66-78
    SdfFormTraceResult result;
    if (!Form_tryTrace(object.Form, ray, result)) return false;              // :67-68
    V3 normal = Form_normalFromRay(object.Form, result.ray);                 // :70
    out.ray = Ray_move(-ray.Epsilon, result.ray);                            // :73
    out.Normal = normal;                                                     // :74
    out.Color = object.Material.Color(result.ray.Origin, normal);            // :75-77
    return true;
}

// ---------------------------------------------------------------------------
// SdfLight.fs:6-42
// ---------------------------------------------------------------------------
SdfLight Light_directional(V3 directionIn, FColor color) {                   // SdfLight.fs:6-21
    V3 direction = Normalize(-directionIn);                                  // :7
    SdfLight l;
    l.Direction = [direction](V3) { return direction; };                     // :9
    l.Intensity = [direction, color](const SdfObject& o, const Ray& r, FColor& out) -> bool {
        Ray s{r.Origin, direction, 1000.0f, r.Epsilon};                      // :11-16
        tl_cnt.rays_shadow++;
        SdfObjectTraceResult tr;
        if (!Object_tryTrace(o, s, tr)) { out = color; return true; }        // :17-19
        tl_cnt.hits_shadow++;
        return false;                                                        // :20
    };
    return l;
}
SdfLight Light_point(V3 position, FColor color) {                            // SdfLight.fs:23-42
    SdfLight l;
    l.Direction = [position](V3 p) { return Normalize(position - p); };      // :25
    l.Intensity = [position, color](const SdfObject& o, const Ray& r, FColor& out) -> bool {
        V3 diff = position - r.Origin;                                       // :27
        float distance2 = LengthSquared(diff);                               // :28
        float distance = sqrtf(distance2);                                   // :29
        V3 direction = diff / distance2;                                     // :30 (not unit — reference quirk)
        Ray s{r.Origin, direction, distance, r.Epsilon};                     // :32-37
        tl_cnt.rays_shadow++;
        SdfObjectTraceResult tr;
        if (!Object_tryTrace(o, s, tr)) { out = color / distance2; return true; }  // :38-40
        tl_cnt.hits_shadow++;
        return false;
    };
    return l;
}

// ---------------------------------------------------------------------------
// SdfScene.fs:7-28
// ---------------------------------------------------------------------------
const float piInv = 1.0f / 3.14159274101257324f;                             // Math.fs:28-30

FColor Scene_trace(const SdfScene& scene, const Ray& ray) {                  // SdfScene.fs:7-28
    SdfObjectTraceResult result;
    tl_cnt.rays_primary++;
    if (!Object_tryTrace(scene.Object, ray, result)) return scene.BackgroundColor;   // :9-10
    tl_cnt.hits_primary++;
    FColor lightColor = scene.BackgroundColor;                               // :12
    for (const SdfLight& light : scene.Lights) {                             // :13
        V3 lightDirection = light.Direction(result.ray.Origin);              // :14 (result.Position)
        float lightCos = Dot(result.Normal, lightDirection);                 // :15
        if (lightCos > 0.0f) {                                               // :17
            FColor intensity;
            if (light.Intensity(scene.Object, result.ray, intensity))        // :18-20
                lightColor = lightColor + intensity * lightCos;              // :23
        }
    }
    return result.Color * (lightColor * piInv);                              // :28
}

// ---------------------------------------------------------------------------
// EXTENSIONS (not in the reference; BASELINE.json configs ask for them): ambient occlusion and
// several samples per pixel.  Defined here only so that the product's extension has a checker;
// no parity claim is attached to them.
//   AO: after a primary hit, `aoSamples` rays (<= 16) leave the pulled-back hit point in the directions
//   normalize(Normal + AO_DIRS[k]) with Length = aoRadius and the ray's Epsilon, marched with
//   SdfForm.tryTrace; lightColor starts as BackgroundColor * (unoccluded / aoSamples) instead of
//   BackgroundColor.  A NaN direction (zero gradient) counts as unoccluded without a march.
//   spp = n*n: sample k uses the pixel corner offset ((k % n) / n, (k / n) / n); the pixel is the sum
//   of the samples in order divided by spp.  spp = 1, aoSamples = 0 is exactly the reference.
// ---------------------------------------------------------------------------
const float AO_DIRS[16][3] = {
    {0x1.02414ep-3f, -0x1.4c1e18p-2f, 0x1.e00000p-1f}, {-0x1.0bab14p-1f, 0x1.082252p-2f, 0x1.a00000p-1f},
    {0x1.64fce8p-1f, 0x1.9faf66p-3f, 0x1.600000p-1f}, {-0x1.b78ec2p-2f, -0x1.69cc1ap-1f, 0x1.200000p-1f},
    {-0x1.662d3ep-3f, 0x1.c39baap-1f, 0x1.c00000p-2f}, {0x1.88019ap-1f, -0x1.1fe15ep-1f, 0x1.400000p-2f},
    {-0x1.f3fa74p-1f, -0x1.b27cbep-4f, 0x1.800000p-3f}, {0x1.5150bap-1f, 0x1.7fd8c8p-1f, 0x1.000000p-4f},
    {0x1.51e2d4p-6f, -0x1.fee3d2p-1f, -0x1.000000p-4f}, {-0x1.5b4eb4p-1f, 0x1.6bbd02p-1f, -0x1.800000p-3f},
    {0x1.e54554p-1f, -0x1.03ffa2p-4f, -0x1.400000p-2f}, {-0x1.6781e0p-1f, -0x1.1f9d7cp-1f, -0x1.c00000p-2f},
    {0x1.046c0ap-3f, 0x1.a248a2p-1f, -0x1.200000p-1f}, {0x1.9bff54p-2f, -0x1.3585eap-1f, -0x1.600000p-1f},
    {-0x1.21c850p-1f, 0x1.1e0d66p-3f, -0x1.a00000p-1f}, {0x1.38c4f8p-2f, 0x1.55799ap-3f, -0x1.e00000p-1f}};

// ---------------------------------------------------------------------------
// EXTENSION (BASELINE.json config 5; nothing like it runs in the reference): glass with refraction,
// reflection and per-wavelength dispersion.  The only related reference text is the dead `fresnel` in
// Light.fs:30-59; its reflectance expressions (:41-53) and reflect / transmit vectors (:56, :58) are kept,
// with two repairs: cos(theta_i) is taken against the incoming direction (cosi = -dot(N, D), N facing the
// ray), and cos(theta_t) is Snell's sqrt(1 - eta^2 (1 - cosi^2)) (the text's `1 + eta cosi^2 - eta` is not).
// Definition (one path per sample, no ray tree):
//   * f = sign * Distance, sign = +1 outside / -1 inside glass; SdfForm.tryTrace and normalFromRay run on f.
//   * hit on a non-glass leaf: outside -> the reference's shading (incl. AO extension); inside -> absorbed (black).
//   * hit on a glass leaf (material picked at the hit origin like SdfObject.fs:75): after max_bounces glass
//     interactions the path ends black.  n = Ior + Dispersion * cauchy(bin) (Cauchy, relative to 550 nm);
//     eta = n1 / n2; total internal reflection when 1 - eta^2 (1 - cosi^2) < 0; otherwise reflect iff
//     u < Reflectance with u = top 24 bits of lowbias32(seed + (bounce + 1) * 0x27D4EB2F) * 2^-24,
//     seed = x * 0x9E3779B1 + y * 0x85EBCA77 + sample * 0xC2B2AE3D (global pixel coordinates).
//     reflect:  Direction = normalize(D + N * (2 cosi)),                   Origin = hit + N * (2 eps)
//     transmit: Direction = normalize(D * eta + N * (eta cosi - cost)),   Origin = hit - N * (4 eps), sign flips,
//               throughput *= Tint when entering.   Length = the primary ray's Length again.
//   * sample value = (background | shaded colour) * throughput; throughput starts as the sample's wavelength
//     weight (1 when spectral = 0).  Sample k uses wavelength bin k % spectral.
//   * wavelength bins: lambda_j = 400 + (j + 1/2) 300 / spectral nm; RGB response = tents (610+-120, 540+-110,
//     460+-120) normalised so that each channel's weights average 1; cauchy_j = 1/lambda_um^2 - 1/0.55^2.
//     Only + - * / in double, then one rounding to float: identical on every host.
// max_bounces = 0 renders glass as createSolid(Tint), i.e. the reference's path.
// ---------------------------------------------------------------------------
struct SpectralBin { V3 weight; float cauchy; };

void spectral_table(int nw, SpectralBin out[16]) {
    const double centre[3] = {610.0, 540.0, 460.0}, width[3] = {120.0, 110.0, 120.0};
    double resp[16][3], sum[3] = {0.0, 0.0, 0.0};
    for (int j = 0; j < nw; ++j) {
        const double lambda = 400.0 + ((double)j + 0.5) * 300.0 / (double)nw;
        for (int c = 0; c < 3; ++c) {
            double d = lambda - centre[c]; if (d < 0.0) d = -d;
            double t = 1.0 - d / width[c]; if (t < 0.0) t = 0.0;
            resp[j][c] = t; sum[c] += t;
        }
        const double um = lambda / 1000.0;
        out[j].cauchy = (float)(1.0 / (um * um) - 1.0 / (0.55 * 0.55));
    }
    for (int j = 0; j < nw; ++j) {
        float w[3];
        for (int c = 0; c < 3; ++c) w[c] = (float)(resp[j][c] * (double)nw / sum[c]);
        out[j].weight = v3(w[0], w[1], w[2]);
    }
}

inline uint32_t glass_hash(uint32_t seed, uint32_t bounce) {
    uint32_t h = seed + (bounce + 1u) * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

// Light.fs:28-59 with the two repairs named above; N faces the incoming direction D.  `total` = total internal
// reflection (Reflectance 1, no Transmit).
struct Fresnel { float Reflectance; bool total; V3 Reflect, Transmit; };
Fresnel fresnel_ext(float n1, float n2, V3 N, V3 D) {
    Fresnel out;
    float cosi = -Dot(N, D);
    if (!(cosi > 0.0f)) cosi = 0.0f;
    const float eta = n1 / n2;                                               // Light.fs:36
    const float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
    out.Reflect = Normalize(D + N * (2.0f * cosi));                          // Light.fs:56
    out.total = !(k >= 0.0f);
    if (out.total) { out.Reflectance = 1.0f; out.Transmit = v3s(0.0f); return out; }
    const float cost = sqrtf(k);
    float rs, rp;
    { const float a = n2 * cosi, b = n1 * cost, x = (a - b) / (a + b); rs = x * x; }   // Light.fs:41-45
    { const float a = n1 * cosi, b = n2 * cost, x = (a - b) / (a + b); rp = x * x; }   // Light.fs:47-51
    out.Reflectance = 0.5f * (rs + rp);                                      // Light.fs:53
    out.Transmit = Normalize(D * eta + N * (eta * cosi - cost));             // Light.fs:58
    return out;
}

// SdfScene.fs:12-28 on an already traced hit, with the ambient-occlusion extension in front
FColor Scene_shade_ext(const SdfScene& scene, const Ray& hit, V3 Normal, FColor Color, int aoSamples, float aoRadius) {
    FColor lightColor = scene.BackgroundColor;                               // :12
    if (aoSamples > 0) {
        int open = 0;
        for (int k = 0; k < aoSamples; ++k) {
            V3 dir = Normalize(Normal + v3(AO_DIRS[k][0], AO_DIRS[k][1], AO_DIRS[k][2]));
            tl_cnt.rays_ext++;
            if (dir.X != dir.X || dir.Y != dir.Y || dir.Z != dir.Z) { open++; continue; }
            Ray r{hit.Origin, dir, aoRadius, hit.Epsilon};
            SdfFormTraceResult tr;
            if (!Form_tryTrace(scene.Object.Form, r, tr)) open++;
        }
        lightColor = scene.BackgroundColor * ((float)open / (float)aoSamples);
    }
    for (const SdfLight& light : scene.Lights) {                             // :13-26
        V3 lightDirection = light.Direction(hit.Origin);
        float lightCos = Dot(Normal, lightDirection);
        if (lightCos > 0.0f) {
            FColor intensity;
            if (light.Intensity(scene.Object, hit, intensity)) lightColor = lightColor + intensity * lightCos;
        }
    }
    return Color * (lightColor * piInv);                                     // :28
}

FColor Scene_trace_ext(const SdfScene& scene, const Ray& ray, int aoSamples, float aoRadius) {
    if (aoSamples <= 0) return Scene_trace(scene, ray);
    SdfObjectTraceResult result;
    tl_cnt.rays_primary++;
    if (!Object_tryTrace(scene.Object, ray, result)) return scene.BackgroundColor;
    tl_cnt.hits_primary++;
    return Scene_shade_ext(scene, result.ray, result.Normal, result.Color, aoSamples, aoRadius);
}

FColor Scene_trace_path(const SdfScene& scene, Ray ray, int aoSamples, float aoRadius, int maxBounces,
                        uint32_t seed, const SpectralBin* bin) {
    const FColor black{v3s(0.0f)};
    const float length0 = ray.Length;
    V3 thr = bin ? bin->weight : v3s(1.0f);
    float sign = 1.0f;
    uint32_t bounce = 0;
    tl_cnt.rays_primary++;
    const SdfForm& outer = scene.Object.Form;
    SdfForm inner;
    inner.Distance = [&outer](V3 p) { return -outer.Distance(p); };
    inner.Boundary = outer.Boundary;
    for (;;) {
        const SdfForm& form = sign < 0.0f ? inner : outer;
        SdfFormTraceResult res;
        if (!Form_tryTrace(form, ray, res)) return FColor{scene.BackgroundColor.c * thr};
        tl_cnt.hits_primary++;
        const V3 N = Form_normalFromRay(form, res.ray);
        const Ray hit = Ray_move(-ray.Epsilon, res.ray);
        const GlassParams* g = (maxBounces > 0 && scene.Object.Material.Glass) ? scene.Object.Material.Glass(res.ray.Origin) : nullptr;
        if (!g) {
            if (sign < 0.0f) return black;                                   // a diffuse surface seen from inside glass: absorbed
            const FColor color = scene.Object.Material.Color(res.ray.Origin, N);
            return FColor{Scene_shade_ext(scene, hit, N, color, aoSamples, aoRadius).c * thr};
        }
        if (bounce >= (uint32_t)maxBounces) return black;
        if (N.X != N.X || N.Y != N.Y || N.Z != N.Z) return black;
        const V3 D = ray.Direction;
        const float n = bin ? g->Ior + g->Dispersion * bin->cauchy : g->Ior;
        const Fresnel f = fresnel_ext(sign > 0.0f ? 1.0f : n, sign > 0.0f ? n : 1.0f, N, D);
        bool reflect = true;
        if (!f.total) {
            const float u = (float)(glass_hash(seed, bounce) >> 8) * (1.0f / 16777216.0f);
            reflect = u < f.Reflectance;
        }
        tl_cnt.rays_ext++;
        Ray next;
        next.Epsilon = ray.Epsilon; next.Length = length0;
        if (reflect) {
            next.Direction = f.Reflect;
            next.Origin = hit.Origin + N * (2.0f * ray.Epsilon);
        } else {
            next.Direction = f.Transmit;
            next.Origin = hit.Origin - N * (4.0f * ray.Epsilon);
            sign = -sign;
            if (sign < 0.0f) thr = thr * g->Tint;
        }
        bounce++;
        ray = next;
    }
}

// ---------------------------------------------------------------------------
// Camera.fs:11-54, Image.fs:17-35
// ---------------------------------------------------------------------------
struct Camera { V3 Position, Forward, UpScaled, RightScaled; };              // Camera.fs:16-22

float Lens_create(float fieldOfView) {                                       // Camera.fs:11-14
    // F# `sin` on float32: evaluated in double, rounded to float32 (assumption).
    return (float)sin((double)(fieldOfView * 0.5f));
}
Camera Camera_lookAt(V3 Position, V3 LookAt, V3 Up, float NearPlaneSize) {   // Camera.fs:33-42
    V3 forward = Normalize(LookAt - Position);
    V3 right = Normalize(Cross(Up, forward));
    return Camera{Position, forward, Cross(forward, right) * NearPlaneSize, right * NearPlaneSize};
}
Ray Camera_uniformPixelToRay(float epsilon, float length, const Camera& camera, V2 position) {  // Camera.fs:44-54
    Ray r;
    r.Origin = camera.Position;
    r.Direction = Normalize(camera.Forward
                            + (position.X - 0.5f) * camera.RightScaled
                            + (position.Y - 0.5f) * camera.UpScaled);
    r.Epsilon = epsilon;
    r.Length = length;
    return r;
}

// ---------------------------------------------------------------------------
// Arena behind the C API
// ---------------------------------------------------------------------------
std::vector<SdfForm> g_forms;
std::vector<SdfMaterial> g_materials;
std::vector<SdfObject> g_objects;
std::vector<SdfLight> g_lights;
std::vector<SdfScene> g_scenes;
std::string g_err;

inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
int fail(const std::string& e) { g_err = e; return -1; }
bool okf(int h) { return h >= 0 && (size_t)h < g_forms.size(); }
bool oko(int h) { return h >= 0 && (size_t)h < g_objects.size(); }

void addCounters(Counters& a, const Counters& b) {
    for (int i = 0; i < 8; ++i) a.prim[i] += b.prim[i];
    a.root_evals += b.root_evals; a.march_steps += b.march_steps;
    a.rays_primary += b.rays_primary; a.rays_shadow += b.rays_shadow; a.rays_ext += b.rays_ext;
    a.hits_primary += b.hits_primary; a.hits_shadow += b.hits_shadow;
    a.smooth_children += b.smooth_children; a.union_candidates += b.union_candidates;
    a.flags |= b.flags; a.union_tested += b.union_tested;
}

}  // namespace

// ===========================================================================
// C API (ctypes).  Handles are indices into the arenas above; -1 = error,
// message from orc_last_error().
// ===========================================================================
extern "C" {

typedef struct orc_counters {
    uint64_t prim[8];
    uint64_t root_evals, march_steps, rays_primary, rays_shadow, rays_ext;
    uint64_t hits_primary, hits_shadow, smooth_children, union_candidates, flags, union_tested;
} orc_counters;

const char* orc_last_error(void) { return g_err.c_str(); }
void orc_reset(void) {
    g_forms.clear(); g_materials.clear(); g_objects.clear(); g_lights.clear(); g_scenes.clear(); g_err.clear();
}
void orc_set_libm(int on) { g_use_libm = on != 0; }

// The C runtime's expf (op 0) / logf (1) / powf(x, y) (2) of THIS machine over chunks of 2^24 consecutive float bit patterns from lo:
// sums[c] = sum of splitmix64((input bits << 32) | result bits) mod 2^64, every NaN result taken as 0x7fc00000.  The product's device
// restatement of glibc forms the same sums on the GPU (ft_selftest_libm); equal sums over all 256 chunks = equal on every float.
static uint64_t orc_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void orc_libm_checksums(int op, float y, uint32_t lo, int n_chunks, uint64_t* sums, int nthreads) {
    std::atomic<int> next{0};
    auto work = [&]() {
        for (;;) {
            const int c = next.fetch_add(1);
            if (c >= n_chunks) return;
            uint64_t acc = 0;
            const uint32_t base = lo + ((uint32_t)c << 24);
            for (uint32_t k = 0; k < (1u << 24); ++k) {
                const uint32_t u = base + k;
                float x; memcpy(&x, &u, 4);
                volatile float xv = x;                               // keep the call: no constant folding, no builtin expansion
                const float r = op == 0 ? expf(xv) : (op == 1 ? logf(xv) : powf(xv, y));
                uint32_t v; memcpy(&v, &r, 4);
                if (r != r) v = 0x7fc00000u;
                acc += orc_splitmix64(((uint64_t)u << 32) | v);
            }
            sums[c] = acc;
        }
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < std::max(1, nthreads); ++t) ts.emplace_back(work);
    for (auto& t : ts) t.join();
}
void orc_libm_array(int op, const float* x, const float* y, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { volatile float xv = x[i]; out[i] = op == 0 ? expf(xv) : (op == 1 ? logf(xv) : powf(xv, y[i])); }
}

float orc_expf(float x) { return orc_expf_impl(x); }
float orc_logf(float x) { return orc_logf_impl(x); }
float orc_sqrtf(float x) { return sqrtf(x); }
float orc_divf(float a, float b) { return a / b; }
float orc_mathf_min(float a, float b) { return MathF_Min(a, b); }
float orc_mathf_max(float a, float b) { return MathF_Max(a, b); }
void orc_expf_array(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = orc_expf_impl(x[i]); }
void orc_logf_array(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = orc_logf_impl(x[i]); }
void orc_sqrtf_array(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = sqrtf(x[i]); }

int orc_form_sphere(const float c[3], float r) { g_forms.push_back(Prim_sphere(ld3(c), r)); return (int)g_forms.size() - 1; }
int orc_form_capsule(const float from[3], const float to[3], float r) {
    g_forms.push_back(Prim_capsule(ld3(from), ld3(to), r)); return (int)g_forms.size() - 1;
}
int orc_form_torus(const float c[3], const float n[3], float R, float r) {
    g_forms.push_back(Prim_torus(ld3(c), ld3(n), R, r)); return (int)g_forms.size() - 1;
}
int orc_form_triangle(const float v1[3], const float v2[3], const float v3_[3], float r) {
    g_forms.push_back(Prim_triangle(ld3(v1), ld3(v2), ld3(v3_), r)); return (int)g_forms.size() - 1;
}
int orc_form_box(const float c[3], const float half[3]) {
    g_forms.push_back(Prim_box(ld3(c), ld3(half))); return (int)g_forms.size() - 1;
}
static bool gather(const int* hs, int n, std::vector<SdfForm>& out) {
    for (int i = 0; i < n; ++i) { if (!okf(hs[i])) return false; out.push_back(g_forms[hs[i]]); }
    return true;
}
int orc_form_union(const int* forms, int n) {
    std::vector<SdfForm> fs; if (!gather(forms, n, fs)) return fail("bad form handle");
    std::string e; SdfForm f = Form_union(fs, e); if (!e.empty()) return fail(e);
    g_forms.push_back(f); return (int)g_forms.size() - 1;
}
int orc_form_subtract(int a, int b) {
    if (!okf(a) || !okf(b)) return fail("bad form handle");
    g_forms.push_back(Form_subtract(g_forms[a], g_forms[b])); return (int)g_forms.size() - 1;
}
int orc_form_intersect(const int* forms, int n) {
    std::vector<SdfForm> fs; if (!gather(forms, n, fs)) return fail("bad form handle");
    std::string e; SdfForm f = Form_intersect(fs, e); if (!e.empty()) return fail(e);
    g_forms.push_back(f); return (int)g_forms.size() - 1;
}
int orc_form_union_smooth(float strength, const int* forms, int n) {
    std::vector<SdfForm> fs; if (!gather(forms, n, fs)) return fail("bad form handle");
    std::string e; SdfForm f = Form_unionSmooth(strength, fs, e); if (!e.empty()) return fail(e);
    g_forms.push_back(f); return (int)g_forms.size() - 1;
}
float orc_form_distance(int form, const float p[3]) { return okf(form) ? g_forms[form].Distance(ld3(p)) : NAN; }
int orc_form_boundary(int form, float out[4]) {
    if (!okf(form)) return fail("bad form handle");
    const SdfBoundary& b = g_forms[form].Boundary;
    out[0] = b.Center.X; out[1] = b.Center.Y; out[2] = b.Center.Z; out[3] = b.Radius; return 0;
}
// grid inspection: info = aabbMin[3], cellSize[3], cellSizeInv[3]; counts[3]; returns total item count
int64_t orc_form_grid_info(int form, float info[9], int counts[3]) {
    if (!okf(form) || !g_forms[form].grid) return -1;
    const GridDump& g = *g_forms[form].grid;
    const V3* vs[3] = {&g.aabbMin, &g.cellSize, &g.cellSizeInv};
    for (int i = 0; i < 3; ++i) { info[3 * i] = vs[i]->X; info[3 * i + 1] = vs[i]->Y; info[3 * i + 2] = vs[i]->Z; }
    counts[0] = g.countX; counts[1] = g.countY; counts[2] = g.countZ;
    int64_t total = 0; for (auto& c : g.cells) total += (int64_t)c.Items.size();
    return total;
}
// cell_start[ncells+1], centers[3*ncells], lower[total], item[total]
int orc_form_grid_dump(int form, uint32_t* cell_start, float* centers, float* lower, int32_t* item) {
    if (!okf(form) || !g_forms[form].grid) return fail("form has no grid");
    const GridDump& g = *g_forms[form].grid;
    uint32_t pos = 0;
    for (size_t c = 0; c < g.cells.size(); ++c) {
        cell_start[c] = pos;
        centers[3 * c] = g.cells[c].Center.X; centers[3 * c + 1] = g.cells[c].Center.Y; centers[3 * c + 2] = g.cells[c].Center.Z;
        for (auto& it : g.cells[c].Items) { lower[pos] = it.LowerBound; item[pos] = it.Item; ++pos; }
    }
    cell_start[g.cells.size()] = pos;
    return 0;
}

int orc_material_solid(const float rgb[3]) {
    g_materials.push_back(Material_createSolid(FColor{ld3(rgb)})); return (int)g_materials.size() - 1;
}
int orc_material_glass(const float tint[3], float ior, float dispersion) {       // EXTENSION
    g_materials.push_back(Material_createGlass(FColor{ld3(tint)}, ior, dispersion)); return (int)g_materials.size() - 1;
}
// EXTENSION: the wavelength table (nw x 4 floats: weight rgb, cauchy term)
int orc_spectral_table(int nw, float* out) {
    if (nw < 1 || nw > 16) return fail("1 <= nw <= 16");
    SpectralBin b[16]; spectral_table(nw, b);
    for (int j = 0; j < nw; ++j) { out[4 * j] = b[j].weight.X; out[4 * j + 1] = b[j].weight.Y; out[4 * j + 2] = b[j].weight.Z; out[4 * j + 3] = b[j].cauchy; }
    return 0;
}
uint32_t orc_glass_hash(uint32_t seed, uint32_t bounce) { return glass_hash(seed, bounce); }
// EXTENSION: out = Reflectance, total (0/1), Reflect xyz, Transmit xyz
void orc_fresnel(float n1, float n2, const float N[3], const float D[3], float out[8]) {
    const Fresnel f = fresnel_ext(n1, n2, ld3(N), ld3(D));
    out[0] = f.Reflectance; out[1] = f.total ? 1.0f : 0.0f;
    out[2] = f.Reflect.X; out[3] = f.Reflect.Y; out[4] = f.Reflect.Z;
    out[5] = f.Transmit.X; out[6] = f.Transmit.Y; out[7] = f.Transmit.Z;
}
int orc_object_create(int material, int form) {
    if (material < 0 || (size_t)material >= g_materials.size() || !okf(form)) return fail("bad handle");
    g_objects.push_back(Object_create(g_materials[material], g_forms[form])); return (int)g_objects.size() - 1;
}
int orc_object_union(const int* objs, int n) {
    std::vector<SdfObject> os;
    for (int i = 0; i < n; ++i) { if (!oko(objs[i])) return fail("bad object handle"); os.push_back(g_objects[objs[i]]); }
    std::string e; SdfObject o = Object_union(os, e); if (!e.empty()) return fail(e);
    g_objects.push_back(o); return (int)g_objects.size() - 1;
}
int orc_object_subtract(int obj, int form) {
    if (!oko(obj) || !okf(form)) return fail("bad handle");
    g_objects.push_back(Object_subtract(g_objects[obj], g_forms[form])); return (int)g_objects.size() - 1;
}
int orc_object_intersect(int obj, const int* forms, int n) {
    if (!oko(obj)) return fail("bad object handle");
    std::vector<SdfForm> fs; if (!gather(forms, n, fs)) return fail("bad form handle");
    std::string e; SdfObject o = Object_intersect(g_objects[obj], fs, e); if (!e.empty()) return fail(e);
    g_objects.push_back(o); return (int)g_objects.size() - 1;
}
int orc_object_form(int obj) {   // expose an object's Form as a form handle (for distance probes)
    if (!oko(obj)) return fail("bad object handle");
    g_forms.push_back(g_objects[obj].Form); return (int)g_forms.size() - 1;
}
int orc_object_color(int obj, const float p[3], const float n[3], float out[3]) {
    if (!oko(obj)) return fail("bad object handle");
    FColor c = g_objects[obj].Material.Color(ld3(p), ld3(n));
    out[0] = c.c.X; out[1] = c.c.Y; out[2] = c.c.Z; return 0;
}
int orc_light_directional(const float dir[3], const float rgb[3]) {
    g_lights.push_back(Light_directional(ld3(dir), FColor{ld3(rgb)})); return (int)g_lights.size() - 1;
}
int orc_light_point(const float pos[3], const float rgb[3]) {
    g_lights.push_back(Light_point(ld3(pos), FColor{ld3(rgb)})); return (int)g_lights.size() - 1;
}
int orc_scene_create(int obj, const float bg[3], const int* lights, int n) {
    if (!oko(obj)) return fail("bad object handle");
    SdfScene s; s.Object = g_objects[obj]; s.BackgroundColor = FColor{ld3(bg)};
    for (int i = 0; i < n; ++i) {
        if (lights[i] < 0 || (size_t)lights[i] >= g_lights.size()) return fail("bad light handle");
        s.Lights.push_back(g_lights[lights[i]]);
    }
    g_scenes.push_back(s); return (int)g_scenes.size() - 1;
}

float orc_lens_create(float fov) { return Lens_create(fov); }
// cam out: Position, Forward, UpScaled, RightScaled (12 floats, Camera.fs:16-22 order)
void orc_camera_lookat(const float pos[3], const float look[3], const float up[3], float nearPlaneSize, float cam[12]) {
    Camera c = Camera_lookAt(ld3(pos), ld3(look), ld3(up), nearPlaneSize);
    const V3* vs[4] = {&c.Position, &c.Forward, &c.UpScaled, &c.RightScaled};
    for (int i = 0; i < 4; ++i) { cam[3 * i] = vs[i]->X; cam[3 * i + 1] = vs[i]->Y; cam[3 * i + 2] = vs[i]->Z; }
}
static Camera ldcam(const float cam[12]) { return Camera{ld3(cam), ld3(cam + 3), ld3(cam + 6), ld3(cam + 9)}; }

// rays: n x 8 floats (Origin, Direction, Length, Epsilon — Types.fs:9-17 layout); out n x 3
int orc_trace_rays(int scene, const float* rays, int64_t n, float* out, orc_counters* cnt) {
    if (scene < 0 || (size_t)scene >= g_scenes.size()) return fail("bad scene handle");
    const SdfScene& sc = g_scenes[scene];
    tl_cnt = Counters{}; tl_flags = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float* r = rays + 8 * i;
        Ray ray{ld3(r), ld3(r + 3), r[6], r[7]};
        FColor c = Scene_trace(sc, ray);
        out[3 * i] = c.c.X; out[3 * i + 1] = c.c.Y; out[3 * i + 2] = c.c.Z;
    }
    tl_cnt.flags = tl_flags;
    if (cnt) memcpy(cnt, &tl_cnt, sizeof(Counters));
    return 0;
}

// SdfForm.tryTrace scene.Object.Form ray (SdfForm.fs:93-104): out = n x 10 dwords (Ray 8, Distance, hit as int32);
// a miss (ValueNone) is all zeros
int orc_form_try_trace(int scene, const float* rays, int64_t n, float* out, orc_counters* cnt) {
    if (scene < 0 || (size_t)scene >= g_scenes.size()) return fail("bad scene handle");
    const SdfScene& sc = g_scenes[scene];
    tl_cnt = Counters{}; tl_flags = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float* r = rays + 8 * i;
        float* o = out + 10 * i;
        SdfFormTraceResult res;
        if (!Form_tryTrace(sc.Object.Form, Ray{ld3(r), ld3(r + 3), r[6], r[7]}, res)) { memset(o, 0, 40); continue; }
        o[0] = res.ray.Origin.X; o[1] = res.ray.Origin.Y; o[2] = res.ray.Origin.Z;
        o[3] = res.ray.Direction.X; o[4] = res.ray.Direction.Y; o[5] = res.ray.Direction.Z;
        o[6] = res.ray.Length; o[7] = res.ray.Epsilon; o[8] = res.Distance;
        const int32_t one = 1; memcpy(o + 9, &one, 4);
    }
    tl_cnt.flags = tl_flags;
    if (cnt) memcpy(cnt, &tl_cnt, sizeof(Counters));
    return 0;
}
// SdfObject.tryTrace scene.Object ray (SdfObject.fs:66-78): out = n x 16 dwords (Ray 8, Normal 3, Color 3, hit, 0)
int orc_object_try_trace(int scene, const float* rays, int64_t n, float* out, orc_counters* cnt) {
    if (scene < 0 || (size_t)scene >= g_scenes.size()) return fail("bad scene handle");
    const SdfScene& sc = g_scenes[scene];
    tl_cnt = Counters{}; tl_flags = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float* r = rays + 8 * i;
        float* o = out + 16 * i;
        SdfObjectTraceResult res;
        if (!Object_tryTrace(sc.Object, Ray{ld3(r), ld3(r + 3), r[6], r[7]}, res)) { memset(o, 0, 64); continue; }
        o[0] = res.ray.Origin.X; o[1] = res.ray.Origin.Y; o[2] = res.ray.Origin.Z;
        o[3] = res.ray.Direction.X; o[4] = res.ray.Direction.Y; o[5] = res.ray.Direction.Z;
        o[6] = res.ray.Length; o[7] = res.ray.Epsilon;
        o[8] = res.Normal.X; o[9] = res.Normal.Y; o[10] = res.Normal.Z;
        o[11] = res.Color.c.X; o[12] = res.Color.c.Y; o[13] = res.Color.c.Z;
        const int32_t one = 1; memcpy(o + 14, &one, 4); o[15] = 0.0f;
    }
    tl_cnt.flags = tl_flags;
    if (cnt) memcpy(cnt, &tl_cnt, sizeof(Counters));
    return 0;
}

// Image.render (Image.fs:26-35) over columns [x0, x1) of a W x H image; out is
// (x1-x0) x H x 3 floats, x-major / y contiguous like FColor[X,Y] (Array2D.fs:30-38).
// Threading mirrors Array2D.fs:32: workers pull whole x-columns.
int orc_render_ext(int scene, const float cam[12], int W, int H, int x0, int x1, int xstep,
               float epsilon, float length, int spp, int aoSamples, float aoRadius, int maxBounces, int spectral,
               float* out, int nthreads, orc_counters* cnt);

// `xstep` > 1 renders only columns x0, x0+xstep, ... (< x1): the bounded sample bench.py times.
int orc_render_strided(int scene, const float cam[12], int W, int H, int x0, int x1, int xstep,
               float epsilon, float length, float* out, int nthreads, orc_counters* cnt) {
    return orc_render_ext(scene, cam, W, H, x0, x1, xstep, epsilon, length, 1, 0, 0.0f, 0, 0, out, nthreads, cnt);
}

// spp / aoSamples / maxBounces / spectral: EXTENSIONS (see Scene_trace_ext, Scene_trace_path); all at their
// neutral values (1, 0, 0, 0) is the reference's Image.render
int orc_render_ext(int scene, const float cam[12], int W, int H, int x0, int x1, int xstep,
               float epsilon, float length, int spp, int aoSamples, float aoRadius, int maxBounces, int spectral,
               float* out, int nthreads, orc_counters* cnt) {
    int sn = 1; while (sn * sn < spp) ++sn;
    if (spp < 1 || sn * sn != spp || aoSamples < 0 || aoSamples > 16) return fail("spp must be a square, ao_samples <= 16");
    if (maxBounces < 0 || maxBounces > 64 || spectral < 0 || spectral > 16 || (spectral > 0 && spp % spectral != 0))
        return fail("max_bounces <= 64, spectral <= 16 and a divisor of spp");
    SpectralBin bins[16];
    if (spectral > 0) spectral_table(spectral, bins);
    const bool path = maxBounces > 0 || spectral > 0;
    if (scene < 0 || (size_t)scene >= g_scenes.size()) return fail("bad scene handle");
    if (W <= 0 || H <= 0 || x0 < 0 || x1 > W || x0 > x1 || xstep < 1) return fail("bad image range");
    const SdfScene& sc = g_scenes[scene];
    const Camera camera = ldcam(cam);
    const float maxSize = (float)std::max(W, H);                             // Image.fs:18
    if (nthreads < 1) nthreads = 1;
    std::atomic<int> next(0);
    std::vector<Counters> per(nthreads);
    auto worker = [&](int tid) {
        tl_cnt = Counters{}; tl_flags = 0;
        for (;;) {
            const int col = next.fetch_add(1);
            const int x = x0 + col * xstep;
            if (x >= x1) break;
            for (int y = 0; y < H; ++y) {                                    // Array2D.fs:33
                FColor c{v3s(0.0f)};
                if (spp == 1 && aoSamples == 0 && !path) {
                    V2 pos{(float)x / maxSize, (float)y / maxSize};          // Image.fs:20-23,30
                    Ray ray = Camera_uniformPixelToRay(epsilon, length, camera, pos);  // Image.fs:32
                    c = Scene_trace(sc, ray);                                // Image.fs:34
                } else {                                                     // EXTENSION
                    for (int k = 0; k < spp; ++k) {
                        V2 pos{((float)x + (float)(k % sn) / (float)sn) / maxSize, ((float)y + (float)(k / sn) / (float)sn) / maxSize};
                        Ray ray = Camera_uniformPixelToRay(epsilon, length, camera, pos);
                        const uint32_t seed = (uint32_t)x * 0x9E3779B1u + (uint32_t)y * 0x85EBCA77u + (uint32_t)k * 0xC2B2AE3Du;
                        FColor sc_ = path ? Scene_trace_path(sc, ray, aoSamples, aoRadius, maxBounces, seed, spectral > 0 ? &bins[k % spectral] : nullptr)
                                          : Scene_trace_ext(sc, ray, aoSamples, aoRadius);
                        c = k == 0 ? sc_ : c + sc_;
                    }
                    if (spp > 1) c = c / (float)spp;
                }
                float* o = out + ((size_t)col * H + y) * 3;
                o[0] = c.c.X; o[1] = c.c.Y; o[2] = c.c.Z;
            }
        }
        tl_cnt.flags = tl_flags;
        per[tid] = tl_cnt;
    };
    if (nthreads == 1) worker(0);
    else {
        std::vector<std::thread> ts;
        for (int t = 0; t < nthreads; ++t) ts.emplace_back(worker, t);
        for (auto& t : ts) t.join();
    }
    if (cnt) { Counters tot{}; for (auto& c : per) addCounters(tot, c); memcpy(cnt, &tot, sizeof(Counters)); }
    return 0;
}

int orc_render(int scene, const float cam[12], int W, int H, int x0, int x1,
               float epsilon, float length, float* out, int nthreads, orc_counters* cnt) {
    return orc_render_strided(scene, cam, W, H, x0, x1, 1, epsilon, length, out, nthreads, cnt);
}

// primary ray for one pixel (Image.fs:30-32) — lets tests feed identical rays to both sides
void orc_pixel_ray(const float cam[12], int W, int H, int x, int y, float epsilon, float length, float ray[8]) {
    const float maxSize = (float)std::max(W, H);
    Ray r = Camera_uniformPixelToRay(epsilon, length, ldcam(cam), V2{(float)x / maxSize, (float)y / maxSize});
    ray[0] = r.Origin.X; ray[1] = r.Origin.Y; ray[2] = r.Origin.Z;
    ray[3] = r.Direction.X; ray[4] = r.Direction.Y; ray[5] = r.Direction.Z;
    ray[6] = r.Length; ray[7] = r.Epsilon;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Post-processing: Image.toColors (Image.fs:37-50), FColor.gammaInverse / toColor (FColor.fs:43-55),
// Image.toBitmap's buffer order (Image.fs:61-86).  MathF.Pow is platform libm in .NET; like exp / log it is replaced
// by one fixed algorithm (exp(g * log x) in double from + - * / only, rounded to float once) that the product carries
// its own copy of.  The reference draws the noise from one System.Random shared by a parallel map (racy); the noise
// here is 0.5 or a counter-based hash of (x, y, channel, seed) — the reference is comparable to +-1 LSB only.
// ---------------------------------------------------------------------------
static double orc_exp_double(double x) {                             // fdlibm e_exp.c structure, |x| <= 150
    static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const int k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    const double hi = x - (double)k * ln2HI, lo = (double)k * ln2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    uint64_t u; memcpy(&u, &y, 8);
    u += (uint64_t)((int64_t)k << 52);
    double out; memcpy(&out, &u, 8);
    return out;
}
static float orc_powf_impl(float x, float g) {                       // MathF.Pow(x, g), FColor.fs:52-54
    if (g_use_libm) return powf(x, g);                               // orc_set_libm(1): the C runtime's powf, as under .NET on this machine
    if (g == 0.0f) return 1.0f;
    if (x != x || g != g) return NAN;
    if (g == 1.0f) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return g > 0.0f ? 0.0f : INFINITY;
    if (x == INFINITY) return g > 0.0f ? INFINITY : 0.0f;
    if (x == 1.0f) return 1.0f;
    if (g == INFINITY || g == -INFINITY) return ((x < 1.0f) == (g > 0.0f)) ? 0.0f : INFINITY;
    double y = (double)g * orc_log_double((double)x);
    y = y < -150.0 ? -150.0 : (y > 150.0 ? 150.0 : y);
    return (float)orc_exp_double(y);
}
static float orc_dither(uint32_t x, uint32_t y, uint32_t channel, uint32_t seed) {
    uint32_t h = seed ^ (x * 0x9E3779B1u) ^ (y * 0x85EBCA77u) ^ (channel * 0xC2B2AE3Du);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// FColor.toColor, one channel (FColor.fs:45): c * 254.5f + rng.range_01() |> MathF.round_i |> min 255.  MathF.Round is
// round-half-to-even (Math.fs:54,58).  Where Color.FromArgb would throw (NaN -> conv.i4 = INT_MIN, or a negative value)
// the byte is 0.
static uint8_t orc_to_byte(float c, float u) {
    const float v = c * 254.5f + u;
    if (!(v >= 0.0f)) return 0;
    const float r = rintf(v);                                        // default rounding mode: half to even
    return r >= 255.0f ? 255 : (uint8_t)(int)r;
}

extern "C" {
float orc_powf(float x, float g) { return orc_powf_impl(x, g); }

// image: FColor[X,Y] as X x Y x 3 floats.  bmp_order = 0: out = Color[X,Y] as R,G,B bytes (Image.toColors' value);
// 1: the buffer of Image.toBitmap after Array.rev (Image.fs:65-74): entry k = index' -> x = index' % X, y = Y-1 - index'/X
// with index' = X*Y-1-k, stored B,G,R.  Returns the normalisation `max` (Image.fs:40-43).
float orc_tone_map(const float* image, int X, int Y, float gamma, int dither, uint32_t seed, int bmp_order, uint8_t* out) {
    const float gammaInv = 1.0f / gamma;                             // Image.fs:38
    float mx = -INFINITY;                                            // Array2D.Parallel.maxWith FColor.getMaxColor (Image.fs:41-42)
    for (int x = 0; x < X; ++x)
        for (int y = 0; y < Y; ++y) {
            const float* c = image + 3 * ((size_t)x * Y + y);
            const float m = fs_max(c[2], fs_max(c[1], c[0]));        // Math.fs:83: v.X |> MathF.max v.Y |> MathF.max v.Z
            if (m > mx) mx = m;
        }
    mx = fs_max(0.01f, mx);                                          // Image.fs:43
    std::vector<uint8_t> colors((size_t)X * Y * 3);
    for (int x = 0; x < X; ++x)
        for (int y = 0; y < Y; ++y) {
            const float* c = image + 3 * ((size_t)x * Y + y);
            uint8_t* o = colors.data() + 3 * ((size_t)x * Y + y);
            for (int k = 0; k < 3; ++k) {                            // fcolor / max |> gammaInverse gammaInv |> toColor rng (Image.fs:47-49); R, G, B
                const float v = orc_powf_impl(c[k] / mx, gammaInv);
                o[k] = orc_to_byte(v, dither ? orc_dither((uint32_t)x, (uint32_t)y, (uint32_t)k, seed) : 0.5f);
            }
        }
    if (!bmp_order) { memcpy(out, colors.data(), colors.size()); return mx; }
    const size_t n = (size_t)X * Y;
    for (size_t k = 0; k < n; ++k) {                                 // Image.fs:65-74
        const size_t index = n - 1 - k;                              // Array.rev
        const int x = (int)(index % (size_t)X), y = Y - 1 - (int)(index / (size_t)X);
        const uint8_t* c = colors.data() + 3 * ((size_t)x * Y + y);
        out[3 * k] = c[2]; out[3 * k + 1] = c[1]; out[3 * k + 2] = c[0];   // {B; G; R}
    }
    return mx;
}
}  // extern "C"
