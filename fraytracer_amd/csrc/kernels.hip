// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the FrayTracer hot path.
//
// One lane = one ray at a time.  A workgroup is four independent persistent waves; every wave
// runs a per-lane state machine whose only expensive step is "evaluate the scene SDF at my
// query point":
//     MARCH (SdfForm.tryTrace, SdfForm.fs:93-104)  ->  NX,NY,NZ,NC (SdfForm.normal, :106-115)
//     -> LIGHTS (SdfScene.trace, SdfScene.fs:12-26) -> SHADOW march per light (SdfLight.fs:10-20,
//     26-41) -> write FColor (SdfScene.fs:28) -> IDLE -> refill.
// Lanes whose ray finished are refilled from a wave-local chunk of jobs (one global atomic per
// chunk), so secondary (shadow) rays never touch HBM and all 64 lanes keep evaluating — this is
// the "persistent-threads compaction" of the design (DESIGN.md "Kernel").
//
// The arithmetic is IEEE float32 in the reference's operation order; build with
// -ffp-contract=off (no implicit FMA) and hipcc's default correctly rounded sqrt/divide.
#include <hip/hip_runtime.h>

#include "../../include/fraytracer_hip.h"
#include "ft_device.h"
#include "ft_kernels.h"
#include "ft_math.h"
#include "ft_libm.h"

// ------------------------------------------------------------------------------------------------
// primitives (SdfForm.fs:125-268).  `c` points at the primitive's constant-pool record; when the
// index is wave-uniform the compiler turns these into scalar loads.
// ------------------------------------------------------------------------------------------------
// Scene data is immutable for the whole launch.  Reading it through the constant address space lets
// the compiler use scalar loads (s_load_dwordx4/x8) whenever the address is wave-uniform — one SMEM
// fetch per primitive per wave instead of a 64-lane vector load — and plain global loads otherwise.
#define FT_CONST __attribute__((address_space(4)))
typedef const float FT_CONST* cfp;
template <class T> __device__ __forceinline__ const T FT_CONST* as_const(const T* p) {
    return (const T FT_CONST*)(p);
}

__device__ __forceinline__ f3 ld3(cfp c) { return mk3(c[0], c[1], c[2]); }
// structs cannot be copy-constructed across address spaces: field-wise loaders
__device__ __forceinline__ FtInstr ld_instr(const FtInstr FT_CONST* q) {
    FtInstr r; r.op = q->op; r.dst = q->dst; r.src = q->src; r.type = q->type; r.count = q->count; r.data = q->data;
    r.aux = q->aux; r.flags = q->flags; r.f0 = q->f0; r.f1 = q->f1; r.pad0 = 0; r.pad1 = 0; return r;
}
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
struct ItemRegs { v4f a; v4u b; };   // a = lowerBound, bc.xyz ; b = br, typeData, mat, child
__device__ __forceinline__ ItemRegs ld_item(const FtItemRec FT_CONST* q) {
    ItemRegs r;
    r.a = *reinterpret_cast<const v4f FT_CONST*>(q);
    r.b = *reinterpret_cast<const v4u FT_CONST*>(reinterpret_cast<const float FT_CONST*>(q) + 4);
    return r;
}
// record `idx` of a list that starts at the uniform pointer `base`: a 32-bit byte offset (lists stay far below 2^27 records: capi.cpp checks)
// lets the load take the scalar base + 32-bit vector offset form instead of a 64-bit address built per record
__device__ __forceinline__ ItemRegs ld_item_at(const FtItemRec FT_CONST* base, uint32_t idx) {
    return ld_item(reinterpret_cast<const FtItemRec FT_CONST*>(reinterpret_cast<const char FT_CONST*>(base) + (size_t)(uint32_t)(idx << 5)));
}
// float `idx` of the constant pool, addressed the same way (the pool stays below 2^30 floats)
__device__ __forceinline__ const float FT_CONST* pool_at(const float FT_CONST* base, uint32_t idx) {
    return reinterpret_cast<const float FT_CONST*>(reinterpret_cast<const char FT_CONST*>(base) + (size_t)(uint32_t)(idx << 2));
}
__device__ __forceinline__ FtLight ld_light(const FtLight FT_CONST* q) {
    FtLight r; r.type = q->type; r.v[0] = q->v[0]; r.v[1] = q->v[1]; r.v[2] = q->v[2];
    r.color[0] = q->color[0]; r.color[1] = q->color[1]; r.color[2] = q->color[2]; r.pad = 0.0f; return r;
}

// ------------------------------------------------------------------------------------------------
// Dynamic LDS of a workgroup: [ FT_C_COUNT x FT_BLOCK per-lane statistics words | nSlots x FT_BLOCK value
// slots (distance) | nSlots x FT_BLOCK slots (material) | staged prefix of the constant pool ].
// Per-lane statistics live in LDS (word k * FT_BLOCK + tid), not in registers: seven counters would otherwise
// stay live across the whole SDF evaluation.  One ds_add per event.  They sit first so that their address
// does not depend on the scene: primitives can raise a flag without being handed a pointer.
// ------------------------------------------------------------------------------------------------
extern __shared__ float ft_lds[];
enum : uint32_t { FT_C_SHADOW = 0, FT_C_HITP, FT_C_HITS, FT_C_PRIMARY, FT_C_FLAGS, FT_C_EXT, FT_C_COUNT };
static_assert(FT_C_COUNT * (FT_BLOCK / 64) <= 32, "ft_kernels.h: the statistics words of the LDS header");
// Statistics are per WAVE: word k * 4 + wave of the header (round 2 kept one word per lane and counter: 7 KB per workgroup, which now
// holds the lanes' shading state instead).  The lanes that reach a counting site together add their number with ONE ds_add by their first
// lane; scene evaluations are not counted here at all (a wave-uniform register in ft_trace_body).  Behind the 32 words: the start clocks
// of the workgroup's first wave.  The header's address does not depend on the scene: a primitive can raise a flag without a pointer.
__device__ __forceinline__ void ft_count(uint32_t k) {
    const unsigned long long m = __ballot(1);                          // the lanes executing this call
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1))
        __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(ft_lds) + k * (FT_BLOCK / 64) + (threadIdx.x >> 6), (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ft_flag(uint32_t bits) {              // `bits` is a constant of the call site
    const unsigned long long m = __ballot(1);
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1))
        __hip_atomic_fetch_or(reinterpret_cast<uint32_t*>(ft_lds) + FT_C_FLAGS * (FT_BLOCK / 64) + (threadIdx.x >> 6), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// A lane's shading state — hit position, normal (and the three probes before it), accumulated light, the current light's contribution —
// is touched only between marches, so it lives in per-lane LDS rows (row r of thread t at FT_LDS_SH_BASE + r * FT_BLOCK + t: conflict-free),
// not in 13 registers that would stay live across every scene evaluation: that is what lets the general kernels keep more waves resident.
// The intensity and cosine of the light a shadow ray belongs to are not kept (round 3 did: four more rows, 4 KB per workgroup — the lean kernel's fifth
// workgroup per CU): when the ray has missed they are formed again from the light's record, the hit position and the normal, by the operations that
// formed them when the ray was started — same operands, same roundings, same bits.
// FT_SH_D0 (round 4): the scene's distance at the hit position — the value the fourth normal probe has just computed there (SdfForm.fs:112: D(p) at
// p = Ray.get(-eps), which is SdfObject.fs:73's pulled-back origin, the same expression on the same operands).  Every shadow ray (SdfLight.fs:11-16, 31-36)
// and every EXTENSION ambient-occlusion ray starts at exactly that point, so the first evaluation of its march is this number again: it is taken from here
// instead of being computed a second, third ... time (FT_OPT_REUSE; the reference evaluates it once per light).
enum : uint32_t { FT_SH_HP = 0, FT_SH_NRM = 3, FT_SH_LACC = 6, FT_SH_D0 = 9 };
static_assert(FT_SH_D0 + 1 <= FT_SH_ROWS, "ft_kernels.h: FT_SH_ROWS");
__device__ __forceinline__ float* ft_sh(uint32_t row) { return ft_lds + FT_LDS_SH_BASE + row * FT_BLOCK + threadIdx.x; }
__device__ __forceinline__ f3 sh_get3(uint32_t row) { const float* q = ft_sh(row); return mk3(q[0], q[FT_BLOCK], q[2 * FT_BLOCK]); }
__device__ __forceinline__ void sh_set3(uint32_t row, f3 v) { float* q = ft_sh(row); q[0] = v.x; q[FT_BLOCK] = v.y; q[2 * FT_BLOCK] = v.z; }

// ---- arithmetic of MathF.Exp / MathF.Log (FT_OPT_MATH) ------------------------------------------------------------------------
// MATH = 0: the fixed algorithms of ft_math.h (same bits on every machine; the default).  MATH = 1: glibc's expf / logf restated
// (ft_libm.h) — what the reference's MathF.Exp / Log (SdfForm.fs:80,82) return under .NET on Linux x86-64.  The MATH = 1 kernels are
// separate instantiations (the default kernels carry no double-precision code); which of glibc's two builds — FMA or SSE2 — is the
// wave-uniform FtSceneDev.mathFma.  The 640-byte table block sits in LDS behind the staged constants (8-byte aligned).
__device__ const ft_u64 ft_libm_tab_g[FT_LIBM_TAB_DOUBLES] = FT_LIBM_TAB_INIT;
__device__ __forceinline__ uint32_t ft_libm_lds_offset(const FtSceneDev& S) { return (FT_LDS_HDR_FLOATS + 2u * S.nSlots * FT_BLOCK + S.nStage + 1u) & ~1u; }
__device__ __forceinline__ const ft_u64* ft_libm_tab(const FtSceneDev& S) { return reinterpret_cast<const ft_u64*>(ft_lds + ft_libm_lds_offset(S)); }
template <int MATH> __device__ __forceinline__ float ft_exp_m(float x, const FtSceneDev& S) {
    if (MATH == 0) return ft_exp(x);
    return S.mathFma ? ft_glibc_expf<true>(x, ft_libm_tab(S)) : ft_glibc_expf<false>(x, ft_libm_tab(S));
}
template <int MATH> __device__ __forceinline__ float ft_log_m(float x, const FtSceneDev& S) {
    if (MATH == 0) return ft_log(x);
    return S.mathFma ? ft_glibc_logf<true>(x, ft_libm_tab(S)) : ft_glibc_logf<false>(x, ft_libm_tab(S));
}

// ---- square roots -----------------------------------------------------------------------------------------
// ft_sqrt_fast: bit-identical to sqrtf for every float in [2^-96, 2^100] (proved by exhaustion,
// ft_selftest_fastmath).  ft_sq<FQ>: FQ = false is the IEEE sqrtf.  FQ = true is used only where the
// caller has established (a) the operand is finite and < 2^100 (p finite, |p|inf < 20000, all scene
// coordinates <= 1e4: fast_point_ok + flatten-time checks) and (b) the root is next reduced by a radius
// >= 2^-20: then clamping the operand at 2^-96 cannot change the difference (both sqrt(q) and 2^-48 are
// far below half an ulp of the radius), so the 6-instruction clamped form gives the reference's value.
#define FT_FAST_Q_MIN 0x1p-96f
__device__ __forceinline__ float ft_sqrt_fast(float x) {
    const float r = __builtin_amdgcn_rsqf(x);      // v_rsq_f32
    const float s = x * r;                          // ~1 ulp estimate of sqrt(x)
    const float h = 0.5f * r;                       // ~1/(2 sqrt(x))
    const float d = fmaf(-s, s, x);                 // residual x - s^2 (one rounding)
    return fmaf(d, h, s);                           // s + d/(2 sqrt(x)): correctly rounded on the whole proved range
}

// The same root in FOUR instructions, for code that runs with output modifiers enabled (FT_OMOD_ON / FT_OMOD_OFF below): the halving of the
// seed and the doubling of q*h ride on the instructions' output modifiers.  h = rsq(x)/2 and s = 2 * rn(x*h) = rn(x * rsq(x)) are the values
// ft_sqrt_fast computes (scaling by 2 commutes with rounding away from the subnormal range, and x in [2^-96, 2^100] keeps both far from it),
// so the result is bit-identical — and proved so by exhaustion under the same mode (ft_selftest_fastmath).
__device__ __forceinline__ float ft_sqrt_fast_omod(float x) {
    float h, s;
    asm("v_rsq_f32_e64 %0, %1 div:2" : "=v"(h) : "v"(x));
    asm("v_mul_f32_e64 %0, %1, %2 mul:2" : "=v"(s) : "v"(x), "v"(h));
    const float d = fmaf(-s, s, x);
    return fmaf(d, h, s);
}
// The hardware honours output modifiers only with MODE.IEEE = 0 AND f32 denormals flushed (MODE.FP_DENORM[1:0] = 0).  FT_OMOD_ON saves
// MODE[9:4] and clears those three bits, FT_OMOD_OFF restores the saved value.  Only the NEAR sphere loop runs in between: its operands are
// finite (fast_point_ok), every exponential is a normal number (near_point_ok), and the values flushing could touch cannot reach a result —
// a subnormal square or residual is below half an ulp of what it is added to or sits under the 2^-96 clamp — which the exhaustive checks
// of the root and of the exponent-add exp confirm under this very mode.  The operands tie the switch into the data flow: everything computed
// from p comes after FT_OMOD_ON, everything that uses the sum after FT_OMOD_OFF.
__device__ __forceinline__ uint32_t ft_omod_on(f3& p) {
    uint32_t saved, tmp;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE, 4, 6)\n\ts_andn2_b32 %1, %0, 0x23\n\ts_setreg_b32 hwreg(HW_REG_MODE, 4, 6), %1\n\ts_nop 1"
                 : "=&s"(saved), "=&s"(tmp), "+v"(p.x), "+v"(p.y), "+v"(p.z));
    return saved;
}
__device__ __forceinline__ void ft_omod_off(uint32_t saved, float& v) {
    asm volatile("s_setreg_b32 hwreg(HW_REG_MODE, 4, 6), %1\n\ts_nop 1" : "+v"(v) : "s"(saved));
}

template <bool FQ> __device__ __forceinline__ float ft_sq(float q) {
    return FQ ? ft_sqrt_fast(__builtin_fmaxf(q, FT_FAST_Q_MIN)) : sqrtf(q);
}
template <bool FQ> __device__ __forceinline__ float ft_dist(f3 a, f3 b) { const f3 d = a - b; return ft_sq<FQ>(ft_dot(d, d)); }

template <bool FQ> __device__ __forceinline__ float prim_sphere(cfp c, f3 p) {
    return ft_dist<FQ>(ld3(c), p) - c[3];                              // SdfForm.fs:129
}

template <bool FQ> __device__ __forceinline__ float prim_capsule(cfp c, f3 p) {
    const f3 diff = p - ld3(c);                                        // :153
    const f3 dir = ld3(c + 4);
    const float t = ft_dot(diff, ld3(c + 8));                          // :154
    f3 w = dir * t;                                                    // :160
    if (t >= 1.0f) w = dir;                                            // :157-158
    if (t <= 0.0f) w = mk3(0.0f, 0.0f, 0.0f);                          // :155-156  (diff - 0 == diff)
    return ft_dist<FQ>(diff, w) - c[3];                                // :164
}

template <bool FQ> __device__ __forceinline__ float prim_torus(cfp c, f3 p) {
    const f3 n = ld3(c + 4);
    const float distanceToPlane = ft_dot(p, n) + c[8];                 // :190
    const float distanceToCenter = ft_dist<FQ>(ld3(c), p - (distanceToPlane * n));   // :191
    const float distanceToCircle = distanceToCenter - c[3];            // :192
    return ft_sq<FQ>(distanceToPlane * distanceToPlane + distanceToCircle * distanceToCircle) - c[7];   // :194
}

template <bool FQ> __device__ __forceinline__ float prim_triangle(cfp c, f3 p) {
    const f3 p1 = p - ld3(c), p2 = p - ld3(c + 4), p3 = p - ld3(c + 8);   // :228-230
    float distance;
    const float e1 = ft_dot(ld3(c + 40), p1), e2 = ft_dot(ld3(c + 44), p2), e3 = ft_dot(ld3(c + 48), p3);
    if (e1 != e1 || e2 != e2 || e3 != e3) ft_flag(2u);                 // MathF.Sign(NaN) throws in .NET (Math.fs:40): flag bit 1, sign taken as 0
    const int s = ft_sign_i(e1) + ft_sign_i(e2) + ft_sign_i(e3);
    if (s < 2) {                                                       // :235-237
        const f3 v21 = ld3(c + 12), v32 = ld3(c + 16), v13 = ld3(c + 20);
        const float d21 = ft_distance2(p1, v21 * ft_clamp01(ft_dot(ld3(c + 24), p1)));   // :240
        const float d32 = ft_distance2(p2, v32 * ft_clamp01(ft_dot(ld3(c + 28), p2)));   // :241
        const float d13 = ft_distance2(p3, v13 * ft_clamp01(ft_dot(ld3(c + 32), p3)));   // :242
        distance = ft_sq<FQ>(ft_min(d13, ft_min(d32, d21)));           // :243-244
    } else {
        distance = fabsf(ft_dot(ld3(c + 36), p1));                     // :247-248
    }
    return distance - c[3];                                            // :250
}

__device__ __forceinline__ float prim_box(cfp c, f3 p) {   // EXTENSION (always the IEEE sqrt: its root is not reduced by a radius)
    const f3 d = p - ld3(c);
    const f3 q = mk3(fabsf(d.x) - c[4], fabsf(d.y) - c[5], fabsf(d.z) - c[6]);
    const f3 qp = mk3(ft_max(q.x, 0.0f), ft_max(q.y, 0.0f), ft_max(q.z, 0.0f));
    return ft_length(qp) + ft_min(ft_max(q.x, ft_max(q.y, q.z)), 0.0f);
}

template <bool FQ> __device__ __forceinline__ float prim_eval_t(uint32_t type, cfp c, f3 p) {
    switch (type) {
        case FT_PR_SPHERE: return prim_sphere<FQ>(c, p);
        case FT_PR_CAPSULE: return prim_capsule<FQ>(c, p);
        case FT_PR_TORUS: return prim_torus<FQ>(c, p);
        case FT_PR_TRIANGLE: return prim_triangle<FQ>(c, p);
        default: return prim_box(c, p);
    }
}
__device__ __forceinline__ float prim_eval(uint32_t type, cfp c, f3 p) { return prim_eval_t<false>(type, c, p); }

__device__ __forceinline__ uint32_t prim_stride(uint32_t type) {
    switch (type) {
        case FT_PR_SPHERE: return FT_STRIDE_SPHERE;
        case FT_PR_CAPSULE: return FT_STRIDE_CAPSULE;
        case FT_PR_TORUS: return FT_STRIDE_TORUS;
        case FT_PR_TRIANGLE: return FT_STRIDE_TRIANGLE;
        default: return FT_STRIDE_BOX;
    }
}

// ------------------------------------------------------------------------------------------------
// Guarded fast forms of sqrt and exp for the uniform smooth-union loop.  Both give bit-identical
// results to sqrtf / ft_exp on their stated ranges (proved by exhaustion over every float in the
// range: ft_selftest_fastmath, tests/test_gpu_parity.py); callers fall back outside.
//   ft_sqrt_fast: q in [2^-96, 2^100].  v_rsq_f32 seed, s = q*r, one residual correction s + (q - s*s)*(r/2):
//                 1 quarter-rate + 4 full-rate ops, no compares/selects (hipcc's IEEE sqrtf is ~17
//                 instructions with denormal scaling and fix-ups; the classic coupled-Newton form needs 8).
//                 x*rsq(x) alone is wrong for 29 % of the range, raw v_sqrt_f32 for 15 %; this form for none.
//   ft_exp_fast:  t in [-2.9e6, 88]: no NaN test and no clamps; n is taken from the mantissa of the
//                 magic-number sum (one integer subtract) instead of v_rndne + v_cvt (both half rate).
// ------------------------------------------------------------------------------------------------
template <bool NEAR>
__device__ __forceinline__ float ft_exp_fast(float x) {
    const float tm = fmaf(x, 0x1.715476p+0f, 12582912.0f);
    const float n = tm - 12582912.0f;
    float r = fmaf(n, -0x1.62e4p-1f, x);
    r = fmaf(n, -0x1.7f7d1cp-20f, r);
    float pz = 0x1.6d7538p-10f;
    pz = fmaf(pz, r, 0x1.120b72p-7f);
    pz = fmaf(pz, r, 0x1.5554b8p-5f);
    pz = fmaf(pz, r, 0x1.5554dcp-3f);
    pz = fmaf(pz, r, 0x1.0p-1f);
    pz = fmaf(pz, r, 1.0f);
    pz = fmaf(pz, r, 1.0f);
    // NEAR (x in [-87, 88], result a normal number): 2^n is applied by adding n, which sits in the low
    // mantissa bits of tm, to the exponent field — one v_lshl_add_u32 instead of v_sub_u32 + v_ldexp_f32.
    if (NEAR) return __uint_as_float((__float_as_uint(tm) << 23) + __float_as_uint(pz));
    // otherwise v_ldexp_f32: rounds subnormal results correctly and flushes to 0 / inf beyond, so no clamps
    return __builtin_amdgcn_ldexpf(pz, (int)(__float_as_uint(tm) - 0x4B400000u));
}

#define FT_FAST_P_MAX 20000.0f             // |p|inf bound under which every |c - p|^2 < 2^32 (|c|inf <= 10000, scene.cpp)
#define FT_FAST_T_LO (-2900000.0f)         // |t * log2e| < 2^22: the magic-number rounding stays exact
#define FT_FAST_T_HI 88.0f

// sum += exp(si * (|c_i - p| - r_i)) for `count` spheres whose (c, r) records sit in LDS at ldsC
// (SdfForm.fs:77-80 with sphere children, :129).  Four children per step: their parameter reads are
// LDS broadcasts (ds_read_b128; no SGPR operands, which halve the VALU rate on gfx950) and their four
// dependency chains interleave.  The additions into `sum` stay in child order.
//
// Preconditions, checked by the caller once per evaluation: p finite with |p|inf < FT_FAST_P_MAX.
// With the flatten-time bounds (scene.cpp fastSphereRun: |c|inf <= 1e4, 2^-20 <= r <= 1e4, strength
// range) every q = |c - p|^2 is finite and < 2^32, and t lies in the range ft_exp_fast is proved on.
// Below 2^-96 (p within 1e-14 of a centre) q is clamped: sqrt(q) and sqrt(2^-96) = 2^-48 are both
// far below half an ulp of r >= 2^-20, so d = s - r rounds to -r either way — the clamp cannot change
// the result and ft_sqrt_fast never sees an operand outside [2^-96, 2^32).
#ifndef FT_UNROLL
#define FT_UNROLL 4
#endif
// Code placement of the hot loop.  The loop is ~72 four-byte instructions followed by one run of ~38 64-bit encoded VALU
// instructions (the exp part); on gfx950 a frame takes 7 % more shader cycles when that run starts on an 8-byte boundary
// than when it starts at 4 mod 8 (measured: profiles/r02_pad_sweep_*.jsonl, r02_asm_*.jsonl).  Which
// one a plain compile produces depends on the dword count of the code in front.  FT_LOOP_PHASE marks the spot: the build's
// layout pass (csrc/loop_layout.py, run by the Makefile on the device assembly) sets the number of s_nops behind the
// 64-byte boundary to 0 or 1 per loop so that every loop lands in the fast phase, and verifies it in the disassembly.
// FT_LOOP_PAD: the s_nop count of a build WITHOUT the pass (tools/pad_sweep_build.sh sweeps it; the pass overwrites it).
#ifndef FT_LOOP_PAD
#define FT_LOOP_PAD 0
#endif
#define FT_STR2(x) #x
#define FT_STR(x) FT_STR2(x)
#define FT_LOOP_PHASE() asm volatile(".p2align 6\n\t.rept " FT_STR(FT_LOOP_PAD) "\n\ts_nop 0\n\t.endr" ::: "memory")

#ifndef FT_SQRT_5
#define FT_LOOP_SQRT(q) (NEAR ? ft_sqrt_fast_omod(q) : ft_sqrt_fast(q))
#else
#define FT_LOOP_SQRT(q) ft_sqrt_fast(q)
#endif
// t = si * (s - w) for the strengths -2, -4 and -1/2 (unionSmooth 0.5, 0.25 — BASELINE.json's configs 3 and 4 — and 2): a product with a power of
// two is exact, so t = (w - s) * 2^k, and the scaling rides on the subtraction's output modifier (PW = 1: mul:2, 2: mul:4, 3: div:2; only inside the
// near loop's mode region).  |w - s| is 0 or at least 2^-44 and below 2^17, so nothing under- or overflows; a zero may come out as +0 where the product
// gives -0, which ft_exp_fast maps to the same 1.0f.  PW = 0 is the general product.
template <int PW> __device__ __forceinline__ float ft_strength_times_diff(float si, float s, float w) {
    float t;
    if (PW == 1) asm("v_sub_f32_e64 %0, %1, %2 mul:2" : "=v"(t) : "v"(w), "v"(s));
    else if (PW == 2) asm("v_sub_f32_e64 %0, %1, %2 mul:4" : "=v"(t) : "v"(w), "v"(s));
    else if (PW == 3) asm("v_sub_f32_e64 %0, %1, %2 div:2" : "=v"(t) : "v"(w), "v"(s));
    else t = si * (s - w);
    return t;
}
__device__ __forceinline__ int ft_strength_pw(float si) { return si == -2.0f ? 1 : (si == -4.0f ? 2 : (si == -0.5f ? 3 : 0)); }

// CLAMP = false (round 4): the culling pass of this wave and round has established that no ray's point lies within 1e-6 of any child's centre
// (ft_cull_children: |c - q0| - rho >= 1e-5 |c - q0| + 1e-6 for every child looked at), so every q is >= 1e-12 and the clamp at 2^-96 is a no-op: 25 instructions per child
template <bool NEAR, int PW = 0, bool CLAMP = true>
__device__ __forceinline__ float smooth_run_spheres_fast(const float* __restrict__ ldsC, uint32_t count, float si_, f3 p, float sum) {
    static_assert(NEAR || PW == 0, "output modifiers exist only in the near loop's mode region");
    static_assert(CLAMP || NEAR, "the clamp-free form exists for the near loop only");
    float si = si_;
    asm volatile("" : "+v"(si));                                       // keep the strength in a VGPR (SGPR operands issue at half rate)
    uint32_t i = 0;
#ifndef FT_SQRT_5
    uint32_t mode = 0;
    if (NEAR) mode = ft_omod_on(p);
#endif
    FT_LOOP_PHASE();
    for (; i + FT_UNROLL <= count; i += FT_UNROLL) {
        float4 prm[FT_UNROLL];
        float q[FT_UNROLL];
#pragma unroll
        for (int j = 0; j < FT_UNROLL; ++j) prm[j] = *reinterpret_cast<const float4*>(ldsC + 4 * (i + j));
#pragma unroll
        for (int j = 0; j < FT_UNROLL; ++j) {
            const float dx = prm[j].x - p.x, dy = prm[j].y - p.y, dz = prm[j].z - p.z;
            q[j] = (dx * dx + dy * dy) + dz * dz;
            if (CLAMP) q[j] = __builtin_fmaxf(q[j], FT_FAST_Q_MIN);
        }
#pragma unroll
        for (int j = 0; j < FT_UNROLL; ++j) sum = sum + ft_exp_fast<NEAR>(ft_strength_times_diff<PW>(si, FT_LOOP_SQRT(q[j]), prm[j].w));
    }
    for (; i < count; ++i) {
        const float4 prm = *reinterpret_cast<const float4*>(ldsC + 4 * i);
        const float dx = prm.x - p.x, dy = prm.y - p.y, dz = prm.z - p.z;
        const float q1 = (dx * dx + dy * dy) + dz * dz;
        sum = sum + ft_exp_fast<NEAR>(ft_strength_times_diff<PW>(si, FT_LOOP_SQRT(CLAMP ? __builtin_fmaxf(q1, FT_FAST_Q_MIN) : q1), prm.w));
    }
#ifndef FT_SQRT_5
    if (NEAR) ft_omod_off(mode, sum);
#endif
    return sum;
}

// The same run with glibc's expf (FT_OPT_MATH, MATH = 1 kernels): sum += expf(si * (|c_i - p| - r_i)).  FQ: the evaluation passed fast_point_ok and
// the run the flatten-time bounds, so the clamped five-instruction root equals sqrtf (as in the loop above); the exponential is the full
// restatement, special cases included, so no range precondition is needed.  Four children per trip: their double-precision chains interleave.
// NEAR: every t of the evaluation lies in [-87, 80] (near_point_ok + the flatten-time bounds, as for the fixed exponential's near form): none of expf's
// special cases can apply and the main path is called directly.
template <bool FMA, bool FQ, bool NEAR = false>
__device__ __forceinline__ float smooth_run_spheres_libm(const float* __restrict__ ldsC, uint32_t count, float si, f3 p, float sum, const ft_u64* __restrict__ tab) {
    uint32_t i = 0;
    for (; i + 4u <= count; i += 4u) {
        float4 prm[4];
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) prm[j] = *reinterpret_cast<const float4*>(ldsC + 4 * (i + j));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dx = prm[j].x - p.x, dy = prm[j].y - p.y, dz = prm[j].z - p.z;
            const float q = (dx * dx + dy * dy) + dz * dz;
            const float t = si * (ft_sq<FQ>(q) - prm[j].w);
            e[j] = NEAR ? ft_glibc_expf_main<FMA>(t, tab) : ft_glibc_expf<FMA>(t, tab);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sum = sum + e[j];
    }
    for (; i < count; ++i) {
        const float4 prm = *reinterpret_cast<const float4*>(ldsC + 4 * i);
        const float dx = prm.x - p.x, dy = prm.y - p.y, dz = prm.z - p.z;
        const float t = si * (ft_sq<FQ>((dx * dx + dy * dy) + dz * dz) - prm.w);
        sum = sum + (NEAR ? ft_glibc_expf_main<FMA>(t, tab) : ft_glibc_expf<FMA>(t, tab));
    }
    return sum;
}

// wave-uniform: are all active lanes inside the radius where every exponential of the fast runs is normal?
__device__ __forceinline__ bool near_point_ok(f3 p, float nearR2) {
    const float pp = p.x * p.x + p.y * p.y + p.z * p.z;               // any rounding is covered by the margin in nearR2
    return __ballot(!(pp <= nearR2)) == 0ull;
}

// wave-uniform precondition of the fast sphere runs for this evaluation
__device__ __forceinline__ bool fast_point_ok(f3 p) {
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p.x), __builtin_fabsf(p.y)), __builtin_fabsf(p.z));
    const bool bad = !(m < FT_FAST_P_MAX) || p.x != p.x || p.y != p.y || p.z != p.z;   // fmax drops NaN operands: test them
    return __ballot(bad) == 0ull;
}

// ------------------------------------------------------------------------------------------------
// union through the uniform grid: SdfBoundary.fs:276-282 lookup, SdfForm.fs:22-34 fold, with the
// argmin of SdfObject.fs:27-46 tracked in the same sweep (same grid, same tests, strict '<').
// Per-lane candidate list; lanes of a wave are neighbouring pixels and mostly share the cell.
// ------------------------------------------------------------------------------------------------
#ifdef FT_UNION_PROFILE
// diagnostic build only (`make profile`, tools/union_divergence.py).  Counters are per-lane words in LDS (rows behind the
// statistics rows, one ds_add per event) and are summed into ft_union_dbg when the kernel ends, so that counting does not slow
// the loop it measures: [0] loop trips summed over lanes, [1] loop trips per wave x 64, [2] candidate evaluations summed over
// lanes, [3] candidate-evaluation blocks per wave x 64, [4] shader cycles inside the union walk (per wave), [5] inside the whole
// scene evaluation, [6] wave-level evaluations x 64, [7] shader cycles of whole rounds (evaluation + state machine + refill)
__device__ unsigned long long ft_union_dbg[12];   // [8] shader cycles a wave waits for the candidate records of a trip, [9] cycles in candidate evaluations,
                                                  // [10] wave-level evaluations x 64 whose active lanes share ONE lookup cell, [11] ... the same cell as the evaluation before
__device__ __forceinline__ void ft_dbg_add(uint32_t k, uint32_t v) {
    __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(ft_lds) + FT_LDS_CNT_WORDS + threadIdx.x + k * FT_BLOCK, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool ft_dbg_leader() { return (threadIdx.x & 63u) == (unsigned)(__ffsll((long long)__ballot(1)) - 1); }
__device__ __forceinline__ uint32_t ft_dbg_now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return (uint32_t)t; }
#define FT_UDBG(k, v) ft_dbg_add(k, v)
#define FT_UDBG_WAVE(k) do { if (ft_dbg_leader()) ft_dbg_add(k, 64u); } while (0)
#define FT_UDBG_T0(v) const uint32_t v = ft_dbg_now()
#define FT_UDBG_T1(k, v) do { const uint32_t _t = ft_dbg_now(); if (ft_dbg_leader()) ft_dbg_add(k, _t - (v)); } while (0)
#else
#define FT_UDBG(k, v) do {} while (0)
#define FT_UDBG_WAVE(k) do {} while (0)
#define FT_UDBG_T0(v) do {} while (0)
#define FT_UDBG_T1(k, v) do {} while (0)
#endif

// cap / capBound ("lazy union", FT_FLAG_LAZY): the union is child 0 of an intersect whose child 1 evaluates to `cap` at p, with pruning bound
// `capBound` (SdfForm.fs:60-63: max = u; if max < bound then max = Max(max, cap)).  The fold below starts at Items.[0] and only falls, so once
// Items.[0]'s distance is <= cap (and < capBound) the intersect's result is `cap` whatever the rest of the walk would find: the lane stops there.
// (The material of an evaluation is only asked for at a hit, and the caller sets cap only where cap >= the ray's epsilon, i.e. no hit; -inf = off.)
template <bool FQ>
__device__ __forceinline__ void eval_union_prims(const FtSceneDev& S, const FtGrid FT_CONST& g, const f3 p,
                                           const float* __restrict__ sd, const uint32_t* __restrict__ sl,
                                           float& outD, uint32_t& outLeaf, float cap, float capBound) {   // unions without FT_PR_CALL children
    FT_UDBG_T0(tWalk);
    const f3 cc = (p - mk3(g.aabbMin[0], g.aabbMin[1], g.aabbMin[2])) * mk3(g.cellSizeInv[0], g.cellSizeInv[1], g.cellSizeInv[2]);
    const int ix = ft_clamp_i(0, g.count[0] - 1, ft_floor_i(cc.x));
    const int iy = ft_clamp_i(0, g.count[1] - 1, ft_floor_i(cc.y));
    const int iz = ft_clamp_i(0, g.count[2] - 1, ft_floor_i(cc.z));
    const uint32_t cell = g.cellBase + (uint32_t)((ix * g.count[1] + iy) * g.count[2] + iz);
    cfp ctr = as_const(S.cellCenters) + 3u * cell;
    const float distanceToCenter = ft_distance(mk3(ctr[0], ctr[1], ctr[2]), p);          // SdfForm.fs:25
    const uint32_t FT_CONST* cellStart = as_const(S.cellStart);
    const FtItemRec FT_CONST* items = as_const(S.items);
    cfp consts = as_const(S.consts);
    uint32_t i = cellStart[cell];
    const uint32_t end = cellStart[cell + 1];
#ifdef FT_UNION_PROFILE
    {
        const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)cell);
        const bool uni = __ballot(cell != u) == 0ull;
        uint32_t* prev = reinterpret_cast<uint32_t*>(ft_lds) + FT_LDS_CNT_WORDS + 12u * FT_BLOCK + (threadIdx.x & ~63u);
        if (uni) { FT_UDBG_WAVE(10); if (*prev == u) FT_UDBG_WAVE(11); }
        if (ft_dbg_leader()) *prev = uni ? u : 0xffffffffu;
    }
#endif

    // The reference scans the whole list (SdfForm.fs:27), testing  min > LowerBound - distanceToCenter (:30)  and
    // min > getMinDistance (:31)  before it evaluates a candidate.  The list is sorted by LowerBound (SdfBoundary.fs:267-268;
    // verified NaN-free when the grid is built) and `mn` never grows, so once :30 fails for one candidate it fails for every
    // later one (float subtraction is monotonic): leaving the loop there gives the identical result.
    //
    // The walk is bound by the latency of its dependent steps (it speeds up in proportion to the resident waves), so every
    // trip handles TWO candidates: both 32-byte records are requested together (they share a 64-byte line half of the time)
    // and both candidates' right-hand sides of :30 / :31 — which do not depend on `mn` — are computed side by side; the
    // decisions are then taken in list order.  An evaluation (8 % of the candidates) is done in one place for either of the
    // two, after which the walk resumes behind the evaluated candidate.  Items.[0] is evaluated unconditionally (:26): it
    // enters that same place with both tests forced true.
    float mn = 0.0f; uint32_t leaf = 0;
    bool first = true;
    while (i < end) {
        const uint32_t j = i + 1u < end ? i + 1u : i;
        FT_UDBG_T0(tLoad);
        ItemRegs ra = ld_item_at(items, i), rb = ld_item_at(items, j);
        // Both records are requested before anything waits for either: left alone, the compiler sinks B's loads behind A's tests (two memory
        // round trips per trip).  Neutral where waves share a SIMD six-fold (Program.fs scene at 4000^2: 12.00 -> 12.03 ms), -5 % where the frame
        // is small and the walk runs at its own latency (the reference's 1000^2: 2.28 -> 2.17 ms); a max-norm pre-test that skips both roots
        // where no lane can pass :31 was measured at the same time and is slower (+3 %: profiles/r03_walk_variants.txt).  So is a walk that always
        // consumes both candidates of a trip (B's right-hand sides kept across A's evaluation, two evaluation sites): bit-exact, 4 spills at 80
        // registers, +1 % at 4000^2 and +4 % at 1000^2 (same file); and a wave-uniform shortcut for the trips that decide nothing (one ballot
        // instead of the per-lane decision tree): +3 % at 4000^2, +6 % on C2 (same file); and loading only the 24 bytes of a record the tests and
        // the primitive switch need, the material index on evaluation only (to relieve the texture addresser): +6 % at 4000^2 (same file).
        asm volatile("" : "+v"(ra.a), "+v"(ra.b), "+v"(rb.a), "+v"(rb.b));
#ifdef FT_UNION_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        FT_UDBG_T1(8, tLoad);
        FT_UDBG(0, 1); FT_UDBG_WAVE(1);
        const float lbA = ra.a.x - distanceToCenter, lbB = rb.a.x - distanceToCenter;
        const float mdA = ft_dist<FQ>(mk3(ra.a.y, ra.a.z, ra.a.w), p) - __uint_as_float(ra.b.x);    // :31 getMinDistance
        const float mdB = ft_dist<FQ>(mk3(rb.a.y, rb.a.z, rb.a.w), p) - __uint_as_float(rb.b.x);
        uint32_t typeData, mat;
        if (first || (mn > lbA && mn > mdA)) { typeData = ra.b.y; mat = ra.b.z; i += 1u; }          // evaluate A, resume at B
        else {
            if (!(mn > lbA)) break;                                    // :30 false for A and every later candidate
            if (j == i || !(mn > lbB)) break;                          // list ends with A, or :30 false for B
            i += 2u;
            if (!(mn > mdB)) continue;                                 // neither candidate is evaluated
            typeData = rb.b.y; mat = rb.b.z;                           // evaluate B
        }
        FT_UDBG(2, 1); FT_UDBG_WAVE(3);
        FT_UDBG_T0(tPrim);
        const uint32_t type = typeData & 15u, data = typeData >> 4;
        float d; uint32_t l;
        if (type == FT_PR_SLOT) { d = sd[data * FT_BLOCK]; l = sl[data * FT_BLOCK]; }
        else { d = prim_eval_t<FQ>(type, pool_at(consts, data), p); l = mat; }
        FT_UDBG_T1(9, tPrim);
        if (first) { mn = d; leaf = l; first = false; if (mn <= cap && mn < capBound) break; }   // lazy union: the rest cannot matter
        else {
            if (d < mn) leaf = l;                                      // SdfObject.fs:41-43
            mn = ft_min(mn, d);                                        // SdfForm.fs:33
        }
    }
    FT_UDBG_T1(4, tWalk);
    outD = mn; outLeaf = leaf;
}

#define FT_CULL_NONE 0xffffffffu
#define FT_CULL_NOCLAMP 0x10000u               // flag on ft_cull_children's survivor count (<= FT_CULL_MAX = 256): no ray of the wave is within 1e-6 of a child's centre this round
// the interpreter (below); WITH_UNION = false is the instance the candidate loop uses for FT_PR_CALL children,
// which contain no union by construction (scene.cpp emitUnion) — that keeps the two mutually non-recursive
template <bool WITH_UNION, bool CALLS, int MATH, bool COOP = false>
__device__ __forceinline__ void ft_exec(const FtSceneDev& S, uint32_t pc, uint32_t pcEnd, const f3 p, float* __restrict__ sd,
                                        uint32_t* __restrict__ sl, const float* __restrict__ ldsC, bool fastOk, bool nearOk, float epsHit = __builtin_inff(),
                                        const float* __restrict__ cullRow = nullptr, uint32_t cullN = FT_CULL_NONE);

template <bool FQ, int MATH>
__device__ __forceinline__ void eval_union(const FtSceneDev& S, const FtGrid FT_CONST& g, const f3 p,
                                           float* __restrict__ sd, uint32_t* __restrict__ sl, const float* __restrict__ ldsC,
                                           bool fastOk, bool nearOk, float& outD, uint32_t& outLeaf, float cap, float capBound) {
    const f3 cc = (p - mk3(g.aabbMin[0], g.aabbMin[1], g.aabbMin[2])) * mk3(g.cellSizeInv[0], g.cellSizeInv[1], g.cellSizeInv[2]);
    const int ix = ft_clamp_i(0, g.count[0] - 1, ft_floor_i(cc.x));
    const int iy = ft_clamp_i(0, g.count[1] - 1, ft_floor_i(cc.y));
    const int iz = ft_clamp_i(0, g.count[2] - 1, ft_floor_i(cc.z));
    const uint32_t cell = g.cellBase + (uint32_t)((ix * g.count[1] + iy) * g.count[2] + iz);
    cfp ctr = as_const(S.cellCenters) + 3u * cell;
    const float distanceToCenter = ft_distance(mk3(ctr[0], ctr[1], ctr[2]), p);          // SdfForm.fs:25
    const uint32_t FT_CONST* cellStart = as_const(S.cellStart);
    const FtItemRec FT_CONST* items = as_const(S.items);
    cfp consts = as_const(S.consts);
    uint32_t i = cellStart[cell];
    const uint32_t end = cellStart[cell + 1];

    // Items.[0] is evaluated unconditionally (SdfForm.fs:26); it is the first trip of the same loop so that the
    // candidate evaluation — primitive switch, slot read, sub-program call — exists once in the code.
    // The reference scans the whole list (SdfForm.fs:27).  The list is sorted by LowerBound
    // (SdfBoundary.fs:267-268; verified NaN-free when the grid is built) and `mn` never grows, so once
    // `mn > LowerBound - distanceToCenter` (:30) fails for one candidate it fails for every later one
    // (float subtraction is monotonic): leaving the loop there gives the identical result.
    float mn = 0.0f; uint32_t leaf = 0;
    bool first = true;
    for (; i < end; ++i) {
        const ItemRegs cur = ld_item_at(items, i);
        if (!first) {
            FT_UDBG(0, 1); FT_UDBG_WAVE(1);
            if (!(mn > cur.a.x - distanceToCenter)) break;             // :30 false for this and all later candidates
            if (!(mn > ft_dist<FQ>(mk3(cur.a.y, cur.a.z, cur.a.w), p) - __uint_as_float(cur.b.x))) continue;   // :31 getMinDistance
            FT_UDBG(2, 1); FT_UDBG_WAVE(3);
        }
        const uint32_t type = cur.b.y & 15u, data = cur.b.y >> 4;
        float d; uint32_t l;
        if (type == FT_PR_SLOT) { d = sd[data * FT_BLOCK]; l = sl[data * FT_BLOCK]; }
        else if (type == FT_PR_CALL) {
            // lanes may need different children: run the sub-programs one at a time, each for the lanes that asked for it
            d = 0.0f; l = 0;
            bool pending = true;
            for (;;) {
                const unsigned long long m = __ballot(pending);
                if (m == 0ull) break;
                const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)data, __ffsll((long long)m) - 1);
                if (pending && data == k) {
                    const uint32_t FT_CONST* cr = reinterpret_cast<const uint32_t FT_CONST*>(consts + k);   // (first instr, end instr, slot)
                    const uint32_t slot = cr[2];
                    ft_exec<false, false, MATH>(S, cr[0], cr[1], p, sd, sl, ldsC, fastOk, nearOk);
                    d = sd[slot * FT_BLOCK]; l = sl[slot * FT_BLOCK];
                    pending = false;
                }
            }
        }
        else { d = prim_eval_t<FQ>(type, pool_at(consts, data), p); l = cur.b.z; }
        if (first) { mn = d; leaf = l; first = false; if (mn <= cap && mn < capBound) break; }   // lazy union (see eval_union_prims)
        else {
            if (d < mn) leaf = l;                                      // SdfObject.fs:41-43
            mn = ft_min(mn, d);                                        // SdfForm.fs:33
        }
    }
    outD = mn; outLeaf = leaf;
}

// ------------------------------------------------------------------------------------------------
// scene SDF: wave-uniform program over per-lane value slots kept in LDS (slot s of thread t at
// word s*FT_BLOCK + t: conflict-free).  Returns scene.Object.Form.Distance(p) and the material
// the reference's material closure would pick at p.
// ------------------------------------------------------------------------------------------------
// Latency mode (see "Latency (tail) mode" below): the grid union of ONE query point, the same in all 64 lanes.  Items.[0] is evaluated by
// every lane (SdfForm.fs:26).  Then, 64 candidates at a time, lane j loads record j of the cell's list and computes the right-hand sides of
// the two pruning tests (:30 LowerBound - distanceToCenter, :31 getMinDistance) — they do not depend on the running minimum — and the wave
// replays the reference's loop IN LIST ORDER over the candidates that can still pass (those that pass against the minimum the chunk starts
// with: the minimum only falls): it broadcasts the candidate's two values, takes :30 and :31 against the current minimum and, where the
// reference would call the candidate's Distance (:33), evaluates it — in every lane, on the same point, through the same code as the
// one-ray-per-lane walk (primitive, slot, or sub-program) — and applies Min / the strict '<' of the material pick (SdfObject.fs:41-43).
// Same tests, same order, same evaluations, same values: the result and the raised flags are those of eval_union / eval_union_prims.
template <bool FQ, bool CALLS, int MATH>
__device__ __forceinline__ void eval_union_coop(const FtSceneDev& S, const FtGrid FT_CONST& g, const f3 p,
                                                float* __restrict__ sd, uint32_t* __restrict__ sl, const float* __restrict__ ldsC,
                                                bool fastOk, bool nearOk, float& outD, uint32_t& outLeaf) {
    const uint32_t lane = threadIdx.x & 63u;
    const f3 cc = (p - mk3(g.aabbMin[0], g.aabbMin[1], g.aabbMin[2])) * mk3(g.cellSizeInv[0], g.cellSizeInv[1], g.cellSizeInv[2]);
    const int ix = ft_clamp_i(0, g.count[0] - 1, ft_floor_i(cc.x));
    const int iy = ft_clamp_i(0, g.count[1] - 1, ft_floor_i(cc.y));
    const int iz = ft_clamp_i(0, g.count[2] - 1, ft_floor_i(cc.z));
    const uint32_t cell = (uint32_t)__builtin_amdgcn_readfirstlane((int)(g.cellBase + (uint32_t)((ix * g.count[1] + iy) * g.count[2] + iz)));
    cfp ctr = as_const(S.cellCenters) + 3u * cell;
    const float distanceToCenter = ft_distance(mk3(ctr[0], ctr[1], ctr[2]), p);          // SdfForm.fs:25
    const uint32_t FT_CONST* cellStart = as_const(S.cellStart);
    const FtItemRec FT_CONST* items = as_const(S.items);
    cfp consts = as_const(S.consts);
    const uint32_t first = cellStart[cell], end = cellStart[cell + 1];

    // evaluate one candidate in every lane (typeData / mat are wave-uniform)
    auto evaluate = [&](uint32_t typeData, uint32_t mat, float& d, uint32_t& l) {
        const uint32_t type = typeData & 15u, data = typeData >> 4;
        if (type == FT_PR_SLOT) { d = sd[data * FT_BLOCK]; l = sl[data * FT_BLOCK]; }
        else if (CALLS && type == FT_PR_CALL) {
            const uint32_t FT_CONST* cr = reinterpret_cast<const uint32_t FT_CONST*>(consts + data);   // (first instr, end instr, slot)
            const uint32_t slot = cr[2];
            ft_exec<false, false, MATH>(S, cr[0], cr[1], p, sd, sl, ldsC, fastOk, nearOk);
            d = sd[slot * FT_BLOCK]; l = sl[slot * FT_BLOCK];
        }
        else { d = prim_eval_t<FQ>(type, pool_at(consts, data), p); l = mat; }
    };

    float mn; uint32_t leaf;
    {
        const ItemRegs r0 = ld_item_at(items, first);                                   // Items.[0]: unconditional (:26)
        evaluate((uint32_t)__builtin_amdgcn_readfirstlane((int)r0.b.y), (uint32_t)__builtin_amdgcn_readfirstlane((int)r0.b.z), mn, leaf);
    }
    bool done = false;
    for (uint32_t base = first + 1u; base < end && !done; base += 64u) {
        const uint32_t idx = base + lane;
        const bool have = idx < end;
        const ItemRegs r = ld_item_at(items, have ? idx : end - 1u);
        const float lb = r.a.x - distanceToCenter;                                      // :30 right-hand side
        const float md = ft_dist<FQ>(mk3(r.a.y, r.a.z, r.a.w), p) - __uint_as_float(r.b.x);   // :31 getMinDistance
        unsigned long long m = __ballot(have && mn > lb && mn > md);
        while (m != 0ull) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const float lbj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lb), j));
            const float mdj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(md), j));
            // :30 is monotonic along the sorted list: once it fails it fails for every later candidate (see eval_union_prims)
            if (__ballot(mn > lbj) == 0ull) { done = true; break; }
            if (__ballot(mn > mdj) == 0ull) continue;                                   // :31
            float d; uint32_t l;
            evaluate((uint32_t)__builtin_amdgcn_readlane((int)r.b.y, j), (uint32_t)__builtin_amdgcn_readlane((int)r.b.z, j), d, l);
            if (d < mn) leaf = l;                                                       // SdfObject.fs:41-43
            mn = ft_min(mn, d);                                                         // SdfForm.fs:33
        }
        // a candidate of this chunk that fails :30 against the minimum the chunk ends with: every later one fails too
        if (!done && __ballot(have && !(mn > lb)) != 0ull) done = true;
    }
    outD = mn; outLeaf = leaf;
}

template <bool WITH_UNION, bool CALLS, int MATH, bool COOP>
__device__ __forceinline__ void ft_exec(const FtSceneDev& S, uint32_t pc, uint32_t pcEnd, const f3 p, float* __restrict__ sd,
                                        uint32_t* __restrict__ sl, const float* __restrict__ ldsC, bool fastOk, bool nearOk, float epsHit,
                                        const float* __restrict__ cullRow, uint32_t cullN) {
    cfp consts = as_const(S.consts);
    for (; pc < pcEnd; ++pc) {
        const FtInstr in = ld_instr(as_const(S.instr) + pc);
        float* dst = sd + in.dst * FT_BLOCK;
        switch (in.op) {
        case FT_OP_PRIM:
            *dst = prim_eval(in.type, consts + in.data, p);
            break;
        case FT_OP_SETLEAF:
            sl[in.dst * FT_BLOCK] = in.aux;
            break;
        case FT_OP_SMOOTH_RUN: {                                       // SdfForm.fs:77-80
            float sum = (in.flags & 1u) ? 0.0f : *dst;
            if (fastOk && S.nStage != 0 && (in.flags & FT_FLAG_FAST) && in.data + 4u * in.count <= S.nStage) {
                // the run — or, where the culling pass ran for this wave and round (pc = FtSceneDev.cullPc), the survivors of its first FT_CULL_MAX children
                // from the wave's LDS row and then the rest of the run in full ("Exact child culling" below; the lean kernel does the same)
                const bool culled = pc == S.cullPc && cullN != FT_CULL_NONE;
                const float* c = culled ? cullRow : ldsC + in.data;
                uint32_t n = culled ? (cullN & 0xffffu) : in.count;
                uint32_t rest = culled && in.count > FT_CULL_MAX ? in.count - FT_CULL_MAX : 0u;
#pragma nounroll
                for (;;) {
                    if (MATH != 0) sum = S.mathFma ? smooth_run_spheres_libm<true, true>(c, n, in.f0, p, sum, ft_libm_tab(S))
                                                   : smooth_run_spheres_libm<false, true>(c, n, in.f0, p, sum, ft_libm_tab(S));
                    else sum = nearOk ? smooth_run_spheres_fast<true>(c, n, in.f0, p, sum)
                                      : smooth_run_spheres_fast<false>(c, n, in.f0, p, sum);
                    if (rest == 0u) break;
                    c = ldsC + in.data + 4u * FT_CULL_MAX; n = rest; rest = 0u;
                }
            } else {
                cfp c = consts + in.data;
                const uint32_t stride = prim_stride(in.type);
                for (uint32_t i = 0; i < in.count; ++i)
                    sum = sum + ft_exp_m<MATH>(in.f0 * prim_eval(in.type, c + i * stride, p), S);
            }
            *dst = sum;
            break;
        }
        case FT_OP_SMOOTH_ADD: {
            const float sum = (in.flags & 1u) ? 0.0f : *dst;
            *dst = sum + ft_exp_m<MATH>(in.f0 * sd[in.src * FT_BLOCK], S);
            break;
        }
        case FT_OP_SMOOTH_FIN:                                         // SdfForm.fs:82
            *dst = -ft_log_m<MATH>(*dst, S) * in.f0;
            break;
        case FT_OP_SUBTRACT:                                           // SdfForm.fs:46-47
            *dst = ft_max(-(sd[in.src * FT_BLOCK]), *dst);
            break;
        case FT_OP_ISECT_RUN: {                                        // SdfForm.fs:60-63
            float mx = *dst;
            cfp c = consts + in.data;
            cfp bd = consts + in.aux;
            const uint32_t stride = prim_stride(in.type);
            for (uint32_t i = 0; i < in.count; ++i)                  // the child is evaluated only where the reference calls it (:62-63)
                if (mx < ft_distance(ld3(bd + 4 * i), p) + bd[4 * i + 3]) mx = ft_max(mx, prim_eval(in.type, c + i * stride, p));
            *dst = mx;
            break;
        }
        case FT_OP_ISECT_APPLY: {
            cfp bd = consts + in.aux;
            const float mx = *dst;
            if (mx < ft_distance(ld3(bd), p) + bd[3]) *dst = ft_max(mx, sd[in.src * FT_BLOCK]);
            break;
        }
        case FT_OP_UNION: {
            if constexpr (WITH_UNION) {
                float d; uint32_t l;
                if constexpr (COOP) {                                  // latency mode: one point, candidates across the lanes
                    if (fastOk && S.fastQ) eval_union_coop<true, CALLS, MATH>(S, as_const(S.grids)[in.aux], p, sd, sl, ldsC, fastOk, nearOk, d, l);
                    else eval_union_coop<false, CALLS, MATH>(S, as_const(S.grids)[in.aux], p, sd, sl, ldsC, fastOk, nearOk, d, l);
                } else
                {
                    // Lazy union (FT_FLAG_LAZY, set by the flattener where this union is child 0 of an intersect whose child 1 is a primitive): that
                    // child's value and pruning bound at p, computed exactly as the FT_OP_ISECT_RUN behind this instruction will compute them.
                    // Only where it is >= the ray's epsilon — the intersect, and everything above it, then cannot produce a hit through this
                    // object, so the material the shortened walk reports is never asked for (epsHit = +inf: callers that want the material everywhere).
                    float cap = -__builtin_inff(), capBound = 0.0f;
                    if (in.flags & FT_FLAG_LAZY) {
                        const float d1 = prim_eval(in.type, consts + in.data, p);
                        cfp bd = consts + in.count;
                        if (d1 >= epsHit) { cap = d1; capBound = ft_distance(ld3(bd), p) + bd[3]; }
                    }
                if constexpr (CALLS) {                                 // scenes with sub-program children (FtSceneDev.fastPath == 2)
                    if (fastOk && S.fastQ) eval_union<true, MATH>(S, as_const(S.grids)[in.aux], p, sd, sl, ldsC, fastOk, nearOk, d, l, cap, capBound);
                    else eval_union<false, MATH>(S, as_const(S.grids)[in.aux], p, sd, sl, ldsC, fastOk, nearOk, d, l, cap, capBound);
                } else {
                    if (fastOk && S.fastQ) eval_union_prims<true>(S, as_const(S.grids)[in.aux], p, sd, sl, d, l, cap, capBound);
                    else eval_union_prims<false>(S, as_const(S.grids)[in.aux], p, sd, sl, d, l, cap, capBound);
                }
                }
                *dst = d; sl[in.dst * FT_BLOCK] = l;
            }
            break;
        }
        default: break;
        }
    }
}

template <bool CALLS, int MATH>
__device__ __forceinline__ void ft_eval(const FtSceneDev& S, const f3 p, float* __restrict__ sd, uint32_t* __restrict__ sl,
                                        const float* __restrict__ ldsC, float& outD, uint32_t& outLeaf, float epsHit = __builtin_inff(),
                                        const float* __restrict__ cullRow = nullptr, uint32_t cullN = FT_CULL_NONE) {
    const bool fastOk = (S.nStage != 0 || S.fastQ != 0) && fast_point_ok(p);
    const bool nearOk = MATH == 0 && fastOk && S.nStage != 0 && near_point_ok(p, S.nearR2);
    ft_exec<true, CALLS, MATH>(S, 0u, S.nInstr, p, sd, sl, ldsC, fastOk, nearOk, epsHit, cullRow, cullN);
    outD = sd[0];
    outLeaf = sl[0];
}

// latency mode: the same program on one point p that all 64 lanes share; only the grid union is spread over the lanes (eval_union_coop),
// every other instruction is computed redundantly by every lane
template <bool CALLS, int MATH>
__device__ __forceinline__ void ft_eval_coop(const FtSceneDev& S, const f3 p, float* __restrict__ sd, uint32_t* __restrict__ sl,
                                             const float* __restrict__ ldsC, float& outD, uint32_t& outLeaf) {
    const bool fastOk = (S.nStage != 0 || S.fastQ != 0) && fast_point_ok(p);
    const bool nearOk = MATH == 0 && fastOk && S.nStage != 0 && near_point_ok(p, S.nearR2);
    ft_exec<true, CALLS, MATH, true>(S, 0u, S.nInstr, p, sd, sl, ldsC, fastOk, nearOk);
    outD = sd[0];
    outLeaf = sl[0];
}

// ------------------------------------------------------------------------------------------------
// Exact child culling for the lean kernel (round 3; FT_OPT_CULL).
//
// unionSmooth adds its children's terms exp(si * d_i) one after the other in float32 (SdfForm.fs:77-80).  A term below half an ulp of
// the running sum leaves that sum unchanged — bit for bit, under round-to-nearest — so the child need not be evaluated at all.  The
// rays of a wave are neighbours (burst refill: one 8x8 tile at a time), so whether a child is such a no-op can be decided ONCE PER WAVE
// AND ROUND from bounds that hold for all of its rays; in the C3 frame a third of all (child, evaluation) pairs go that way.
//
// Once per round, all 64 lanes (also those without a ray) take part: q0 = a point in the middle of the rays' query points, rho >= the
// distance of every such point from q0.  Lane k of pass j looks at child i = 64 j + k (centre c, radius r) and computes, in plain f32
// with generous slack,  dlo <= |c - p| - r <= dhi  for every ray's p (triangle inequality: | |c - q0| - r +- rho |), hence
//     low_i  <=  the child's term in every lane          (2^(si dhi log2e - 0.02) by v_exp_f32)
//     x_i    >=  log2 of the child's term in every lane    (t = si d <= si dlo, the exponentials are within an ulp of e^t: term < 2^(x_i + 0.02))
// and an exclusive prefix sum over the children in list order gives  Slow_i  <=  the running sum every lane holds in front of child i:
// the sum of the lower bounds of ALL earlier children, times (1 - 2^-8), which covers the rounding of that prefix sum, of the lanes'
// own sequential sums (256 x 2^-24) and of their exponentials (< 1 ulp), and the culled children among them, whose lower bounds are
// counted although their terms were absorbed (they add up to less than 255 x 2^-24 of the sum).  Child i is dropped for this wave and
// round iff Slow_i is a normal number and  x_i + 0.02 <= exponent(Slow_i) - 24,  i.e.  term < half an ulp of any float
// >= Slow_i — by induction over the list the running sums are then exactly the reference's.  The first child is never dropped
// (Slow_0 = 0); NaN, infinite or huge points switch the pass off.  Survivors keep their order: their parameter records are copied,
// compacted, into the wave's own LDS row, which the unchanged sphere loops then read instead of the staged constants.
// Cost: ~250 wave-instructions per round against 26 per child and ray saved.
// ------------------------------------------------------------------------------------------------
// FT_CULL_MAX (ft_kernels.h): children per culling pass = float4 records of the wave's LDS row; a longer run's tail is evaluated in full.
// FT_CULL_ROW: floats of that row; its first FT_COOP_SEG floats double as the latency mode's row (never used in the same round).
#define FT_CULL_MIN 32u                        // runs shorter than this are not worth the pass (scene.cpp picks FtSceneDev.cullPc accordingly)
__device__ __forceinline__ f3 ft_readlane3(f3 v, int l) {
    return mk3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), l)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), l)),
               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), l)));
}
// Wave-wide maximum and prefix sum for the pass, from DPP row shifts and v_readlane: no LDS traffic, no dependent round trips (the first version
// used ds_bpermute shuffles: 40 dependent LDS round trips per round — the pass cost 3.7 ms of a 34 ms frame).  All 64 lanes must execute.
// Checked against the host in tools/experiments/dpp_scan.hip.
template <int CTRL> __device__ __forceinline__ float ft_dpp(float old, float v) {      // lanes without a source lane keep `old`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float ft_readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float ft_wave_max_all(float v) {            // v >= 0 in every lane; wave-uniform result
    v = __builtin_fmaxf(v, ft_dpp<0x111>(0.0f, v)); v = __builtin_fmaxf(v, ft_dpp<0x112>(0.0f, v));     // row_shr:1, 2, 4, 8: lane 15 of every row of 16 holds the row's maximum
    v = __builtin_fmaxf(v, ft_dpp<0x114>(0.0f, v)); v = __builtin_fmaxf(v, ft_dpp<0x118>(0.0f, v));
    return __builtin_fmaxf(__builtin_fmaxf(ft_readlane_f(v, 15), ft_readlane_f(v, 31)), __builtin_fmaxf(ft_readlane_f(v, 47), ft_readlane_f(v, 63)));
}
__device__ __forceinline__ float ft_wave_scan_incl(float v, float& total) {            // inclusive prefix sum in lane order; total = lane 63's (wave-uniform)
    const uint32_t lane = threadIdx.x & 63u;
    v += ft_dpp<0x111>(0.0f, v); v += ft_dpp<0x112>(0.0f, v); v += ft_dpp<0x114>(0.0f, v); v += ft_dpp<0x118>(0.0f, v);   // inclusive within each row of 16
    const float r0 = ft_readlane_f(v, 15), r1 = ft_readlane_f(v, 31), r2 = ft_readlane_f(v, 47), r3 = ft_readlane_f(v, 63);
    const float p2 = r0 + r1, p3 = p2 + r2;
    v += lane >= 48u ? p3 : (lane >= 32u ? p2 : (lane >= 16u ? r0 : 0.0f));
    total = p3 + r3;
    return v;
}
__device__ __forceinline__ float ft_wave_shift_right1(float v) { return ft_dpp<0x138>(0.0f, v); }   // wave_shr:1; lane 0 gets 0
// -> survivors among the first min(count, FT_CULL_MAX) children of instruction FtSceneDev.cullPc, their records in row[0 ..), or FT_CULL_NONE (row untouched).
// A run that continues an accumulator (no FT_FLAG_INIT: children of other kinds came first) starts from a sum >= 0, so the prefix sums of the run's own lower
// bounds are still lower bounds of the running sum in front of each of its children.
// active / am: lanes that evaluate p this round (am = __ballot(active) != 0).  Wave-uniform result; executed by all 64 lanes.
__device__ __forceinline__ uint32_t ft_cull_children(const FtSceneDev& S, const f3 p, bool active, unsigned long long am,
                                                     const float* __restrict__ ldsC, float* __restrict__ row) {
    if (S.cullPc == FT_CULL_NONE) return FT_CULL_NONE;
    const FtInstr FT_CONST* in = as_const(S.instr) + S.cullPc;
    const uint32_t flags = in->flags, count = in->count;
    const float si = in->f0;
    if (in->op != FT_OP_SMOOTH_RUN || !(flags & FT_FLAG_FAST) || count < FT_CULL_MIN || !(si < 0.0f)) return FT_CULL_NONE;
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p.x), __builtin_fabsf(p.y)), __builtin_fabsf(p.z));
    const bool bad = !(m < FT_FAST_P_MAX) || p.x != p.x || p.y != p.y || p.z != p.z;
    if (__ballot(active && bad) != 0ull) return FT_CULL_NONE;         // such evaluations take the exact loop anyway
    const uint32_t lane = threadIdx.x & 63u;
    // centre: half way between the first ray's point and the point farthest from it (a cheap stand-in for the smallest enclosing ball:
    // the points of a tile spread mostly along the rays); rho: the largest distance of any ray's point from that centre
    const f3 q00 = ft_readlane3(p, __ffsll((long long)am) - 1);
    const float fx = p.x - q00.x, fy = p.y - q00.y, fz = p.z - q00.z;
    const float far2 = active ? (fx * fx + fy * fy) + fz * fz : 0.0f;
    const float farMax = ft_wave_max_all(far2);
    const unsigned long long fm = __ballot(active && far2 == farMax);  // never empty: the maximum is some active lane's (kept safe all the same)
    const f3 q01 = ft_readlane3(p, __ffsll((long long)(fm != 0ull ? fm : am)) - 1);
    const f3 q0 = mk3(0.5f * (q00.x + q01.x), 0.5f * (q00.y + q01.y), 0.5f * (q00.z + q01.z));
    const float ex = p.x - q0.x, ey = p.y - q0.y, ez = p.z - q0.z;
    const float rho2 = ft_wave_max_all(active ? (ex * ex + ey * ey) + ez * ez : 0.0f);
    const float rho = __builtin_amdgcn_sqrtf(rho2) * 1.001f + 1e-6f;   // >= |p - q0| of every active lane (v_sqrt_f32: 1 ulp)
    const uint32_t n = count < FT_CULL_MAX ? count : FT_CULL_MAX;
    const float4* src = reinterpret_cast<const float4*>(ldsC + in->data);
    float4* dst = reinterpret_cast<float4*>(row);
    float carry = 0.0f;                                                // sum of the lower bounds of the passes done (wave-uniform)
    uint32_t kept = 0;
    bool apart = true;                                                 // every child looked at keeps its centre away from every ray's point (wave-uniform)
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t i = base + lane;
        const bool have = i < n;
        const float4 prm = src[have ? i : n - 1u];
        const float dx = prm.x - q0.x, dy = prm.y - q0.y, dz = prm.z - q0.z;
        const float dq = __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz);             // |c - q0| (v_sqrt_f32: 1 ulp)
        const float dc = dq - prm.w;                                   // |c - q0| - r, within a few ulp of 20000
        // |c - p| >= |c - q0| - |p - q0| >= dq (1 - 3e-7) - rho for every ray's p (rho is an upper bound by construction): with the margin below every squared
        // distance the sphere loop forms is >= 1e-12 (close differences are exact, its five roundings lose < 1e-6 of that) — far above the 2^-96 clamp
        if (__ballot(have && !(dq - rho >= 1e-5f * dq + 1e-6f)) != 0ull) apart = false;
        const float slack = 0.01f + rho;                               // 0.01 >> every rounding here and in the lanes' own distances
        const float dlo = dc - slack, dhi = dc + slack;
        const float low = have ? __builtin_amdgcn_exp2f(__builtin_fmaxf(si * dhi * 1.44269504f - 0.02f, -200.0f)) : 0.0f;
        float passTotal;
        const float incl = ft_wave_scan_incl(low, passTotal);          // inclusive prefix sum over the lanes = over the children in list order
        const float excl = ft_wave_shift_right1(incl);
        const float slow = (carry + excl) * 0.99609375f;              // 1 - 2^-8
        const uint32_t sbits = __float_as_uint(slow);
        const float x = __builtin_fminf(__builtin_fmaxf(si * dlo * 1.44269504f, -1000.0f), 1000.0f);   // log2 of the largest term any lane can compute ...
        // ... up to the rounding of t and of the exponential (< 1 ulp), both far inside the 0.02; half an ulp of a float in [2^e, 2^(e+1)) is 2^(e-24)
        const bool drop = have && x == x && sbits >= 0x00800000u && sbits < 0x7f800000u && x + 0.02f <= (float)((int)(sbits >> 23) - 127 - 24);
        const bool keep = have && !drop;
        const unsigned long long km = __ballot(keep);
        if (keep) dst[kept + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = prm;
        kept += (uint32_t)__popcll(km);
        carry += passTotal;
    }
    return kept | (apart ? FT_CULL_NOCLAMP : 0u);
}

// Lean evaluator for scenes whose whole program is {fast sphere SMOOTH_RUN..., SMOOTH_FIN, SETLEAF}
// (FtSceneDev.fastPath == 1, decided when the scene is flattened): the accumulator lives in a VGPR,
// no value slots, no primitive switch — the kernel variant built on it needs far fewer registers.
// one run of sphere children in the regime the evaluation allows (wave-uniform fastOk / nearOk): sum0 + the run's terms, in list order
template <int MATH>
__device__ __forceinline__ float ft_run_spheres(const FtSceneDev& S, const float* __restrict__ c, uint32_t count, float si, const f3 p, float sum0,
                                                bool fastOk, bool nearOk, bool noClamp = false) {
    if (MATH != 0) {                                                   // FT_OPT_MATH: glibc's expf
        const ft_u64* tab = ft_libm_tab(S);
        if (S.mathFma) return nearOk ? smooth_run_spheres_libm<true, true, true>(c, count, si, p, sum0, tab)
                            : fastOk ? smooth_run_spheres_libm<true, true>(c, count, si, p, sum0, tab)
                                     : smooth_run_spheres_libm<true, false>(c, count, si, p, sum0, tab);
        return nearOk ? smooth_run_spheres_libm<false, true, true>(c, count, si, p, sum0, tab)
             : fastOk ? smooth_run_spheres_libm<false, true>(c, count, si, p, sum0, tab)
                      : smooth_run_spheres_libm<false, false>(c, count, si, p, sum0, tab);
    }
    if (__builtin_expect(nearOk, 1)) {
#ifndef FT_SQRT_5
        const int pw = ft_strength_pw(si);                             // wave-uniform: the strength is an instruction field
        if (noClamp) {                                                 // survivors of a culling pass that found every centre away from every ray's point
            if (pw == 2) return smooth_run_spheres_fast<true, 2, false>(c, count, si, p, sum0);
            if (pw == 1) return smooth_run_spheres_fast<true, 1, false>(c, count, si, p, sum0);
            if (pw == 3) return smooth_run_spheres_fast<true, 3, false>(c, count, si, p, sum0);
            return smooth_run_spheres_fast<true, 0, false>(c, count, si, p, sum0);
        }
        if (pw == 2) return smooth_run_spheres_fast<true, 2>(c, count, si, p, sum0);
        if (pw == 1) return smooth_run_spheres_fast<true, 1>(c, count, si, p, sum0);
        if (pw == 3) return smooth_run_spheres_fast<true, 3>(c, count, si, p, sum0);
#endif
        return smooth_run_spheres_fast<true>(c, count, si, p, sum0);
    }
    if (fastOk) return smooth_run_spheres_fast<false>(c, count, si, p, sum0);
    float acc = sum0;                                                  // exact loop (SdfForm.fs:77-80, :129)
    for (uint32_t i = 0; i < count; ++i) acc = acc + ft_exp(si * (ft_distance(mk3(c[4u * i], c[4u * i + 1u], c[4u * i + 2u]), p) - c[4u * i + 3u]));
    return acc;
}

template <int MATH>
__device__ __forceinline__ void ft_eval_smooth_spheres(const FtSceneDev& S, const f3 p, const float* __restrict__ ldsC,
                                                       float& outD, uint32_t& outLeaf, const float* __restrict__ cullRow = nullptr, uint32_t cullN = FT_CULL_NONE) {
    float acc = 0.0f;
    uint32_t leaf = 0;
    const bool fastOk = fast_point_ok(p);
    const bool nearOk = fastOk && near_point_ok(p, S.nearR2);
    for (uint32_t pc = 0; pc < S.nInstr; ++pc) {
        const FtInstr FT_CONST* in = as_const(S.instr) + pc;
        const uint32_t op = in->op;
        if (op == FT_OP_SMOOTH_RUN) {
            acc = (in->flags & FT_FLAG_INIT) ? 0.0f : acc;
            // the run, or — where ft_cull_children ran for this wave and round — the survivors of its first FT_CULL_MAX children (in the wave's
            // LDS row) and then the rest of the run in full; one inlined copy of the loops serves both pieces
            const bool culled = pc == S.cullPc && cullN != FT_CULL_NONE && fastOk;
            const float* c = culled ? cullRow : ldsC + in->data;
            uint32_t n = culled ? (cullN & 0xffffu) : in->count;
            uint32_t rest = culled && in->count > FT_CULL_MAX ? in->count - FT_CULL_MAX : 0u;
            bool noClamp = culled && (cullN & FT_CULL_NOCLAMP) != 0u;     // holds for the children the pass looked at: the survivors in the row
#pragma nounroll
            for (;;) {
                acc = ft_run_spheres<MATH>(S, c, n, in->f0, p, acc, fastOk, nearOk, noClamp);
                if (rest == 0u) break;
                c = ldsC + in->data + 4u * FT_CULL_MAX; n = rest; rest = 0u; noClamp = false;
            }
        }
        else if (op == FT_OP_SMOOTH_FIN) acc = -ft_log_m<MATH>(acc, S) * in->f0;
        else leaf = in->aux;                                           // FT_OP_SETLEAF
    }
    outD = acc; outLeaf = leaf;
}

// ------------------------------------------------------------------------------------------------
// Latency ("tail") mode: ONE ray per wave, its evaluation spread over the 64 lanes.
//
// A persistent wave pays one full scene evaluation per round however few of its lanes still hold a ray, and a kernel ends when its
// longest rays end: once the job queue is empty those rays march on in nearly empty waves, one evaluation latency per step (C3: > 130
// steps of ~30 000 cycles; 20 % of one rank's share of a frame at N = 8, ~40 % of the reference's own 1000^2 frame).  When a wave holds at
// most FtRenderArgs.tailK rays, each of them is therefore evaluated cooperatively, one after the other, by all 64 lanes:
//   * smooth union of spheres (lean kernel): ft_eval_smooth_spheres_packed — all of the wave's rays in one pass, 64 / rays lanes per ray,
//     every ray's exponentials spread over its lanes and added up IN CHILD ORDER from the wave's LDS row (up to 32 rays per wave);
//   * grid union (general kernels): see eval_union_coop.
// Evaluation has no side effects and every value is computed by the same operations as in the one-ray-per-lane path, so the result is
// bit-identical by construction; it is also cheaper in wave instructions as soon as fewer than ~15 lanes hold a ray, so the mode is used
// whenever a wave is that empty, not only at the end of a launch.
// ------------------------------------------------------------------------------------------------
#define FT_COOP_SEG 256                       // children per segment = floats of the wave's LDS row the mode uses
__device__ __forceinline__ uint32_t ft_coop_lds_offset(const FtSceneDev& S, bool libm) {
    const uint32_t end = libm ? ft_libm_lds_offset(S) + 2u * FT_LIBM_TAB_DOUBLES : FT_LDS_HDR_FLOATS + 2u * S.nSlots * FT_BLOCK + S.nStage;
    return (end + 3u) & ~3u;                  // 16-byte aligned: the row is written and read as float4
}
// one child's term exp(si * (|c - p| - r)) in the regime the point allows: 2 = near (exponent-add exp), 1 = far (ldexp exp), 0 = exact
// forms — all three give the same bits wherever two of them are valid (ft_selftest_fastmath), so the choice is only about cost
template <int MATH>
__device__ __forceinline__ float ft_sphere_term(const float4 prm, const f3 p, const float si, const int regime, const FtSceneDev& S) {
    const float dx = prm.x - p.x, dy = prm.y - p.y, dz = prm.z - p.z;
    const float q = (dx * dx + dy * dy) + dz * dz;
    if (MATH != 0) {
        const float t = si * ((regime != 0 ? ft_sq<true>(q) : sqrtf(q)) - prm.w);
        return S.mathFma ? ft_glibc_expf<true>(t, ft_libm_tab(S)) : ft_glibc_expf<false>(t, ft_libm_tab(S));
    }
    if (regime == 2) return ft_exp_fast<true>(si * (ft_sq<true>(q) - prm.w));
    if (regime == 1) return ft_exp_fast<false>(si * (ft_sq<true>(q) - prm.w));
    return ft_exp(si * (sqrtf(q) - prm.w));
}
// The wave's rays — at most 32, `am` = the lanes that hold one, `q` = their query points — are evaluated by groups of g = 64 / (rays rounded
// up to a power of two) lanes each: group G serves the G-th ray.  A segment is 4g children: lane k of a group computes the exponentials
// of children 4k .. 4k+3 of the segment and stores them at its own float4 of the wave's LDS row — which makes the row, group by group,
// the segment's terms in child order — and then every lane of the group adds the group's 4g terms to its running sum, first to last (the
// reference's sequential f32 sum, SdfForm.fs:77-80).  One round costs ~(7000 / g + 400) wave instructions against ~6700 of the
// one-ray-per-lane evaluation, whatever the number of rays: cheaper from 32 rays down, and 1 / 18 of the latency for a single ray.
template <int MATH>
__device__ __forceinline__ void ft_eval_smooth_spheres_packed(const FtSceneDev& S, const f3 q, const bool active, const unsigned long long am, const uint32_t nRays,
                                                              const float* __restrict__ ldsC, float* __restrict__ row, float& outD, uint32_t& outLeaf) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
    const uint32_t lg = nRays <= 1u ? 6u : 6u - (32u - (uint32_t)__builtin_clz(nRays - 1u));      // log2 of the group size: 64 >> ceil(log2 rays)
    const uint32_t g = 1u << lg;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // the rays' query points, in rank order, through the row
    if (active) { row[3u * rank] = q.x; row[3u * rank + 1u] = q.y; row[3u * rank + 2u] = q.z; }
    wave_sync();
    const uint32_t G = lane >> lg, k = lane & (g - 1u);
    const bool valid = G < nRays;
    const f3 p = valid ? mk3(row[3u * G], row[3u * G + 1u], row[3u * G + 2u]) : mk3(0.0f, 0.0f, 0.0f);     // lanes of unused groups: any harmless point
    wave_sync();
    const bool fastOk = fast_point_ok(p);                              // wave-uniform, over all the rays of the round (as in the one-ray-per-lane path)
    const bool nearOk = fastOk && near_point_ok(p, S.nearR2);
    const int regime = nearOk ? 2 : (fastOk ? 1 : 0);
    const float* mine = row + 4u * (lane & ~(g - 1u));                 // this group's 4g terms of the current segment
    float acc = 0.0f;
    uint32_t leaf = 0;
    for (uint32_t pc = 0; pc < S.nInstr; ++pc) {
        const FtInstr FT_CONST* in = as_const(S.instr) + pc;
        const uint32_t op = in->op;
        if (op == FT_OP_SMOOTH_RUN) {
            float sum = (in->flags & FT_FLAG_INIT) ? 0.0f : acc;
            const uint32_t count = in->count;
            const float si = in->f0;
            for (uint32_t seg = 0; seg < count; seg += 4u * g) {
                const uint32_t nv = count - seg < 4u * g ? count - seg : 4u * g;       // children in this segment
                const float* c = ldsC + in->data + 4u * (seg + 4u * k);
                float4 e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (4u * k + 0u < nv) e.x = ft_sphere_term<MATH>(*reinterpret_cast<const float4*>(c), p, si, regime, S);
                if (4u * k + 1u < nv) e.y = ft_sphere_term<MATH>(*reinterpret_cast<const float4*>(c + 4), p, si, regime, S);
                if (4u * k + 2u < nv) e.z = ft_sphere_term<MATH>(*reinterpret_cast<const float4*>(c + 8), p, si, regime, S);
                if (4u * k + 3u < nv) e.w = ft_sphere_term<MATH>(*reinterpret_cast<const float4*>(c + 12), p, si, regime, S);
                *reinterpret_cast<float4*>(row + 4u * lane) = e;
                wave_sync();
                uint32_t i = 0;
                for (; i + 4u <= nv; i += 4u) {                        // the reference's sum, child by child (SdfForm.fs:77-80)
                    const float4 v = *reinterpret_cast<const float4*>(mine + i);
                    sum = sum + v.x; sum = sum + v.y; sum = sum + v.z; sum = sum + v.w;
                }
                for (; i < nv; ++i) sum = sum + mine[i];
                wave_sync();                                           // the row is rewritten by the next segment / round
            }
            acc = sum;
        }
        else if (op == FT_OP_SMOOTH_FIN) acc = -ft_log_m<MATH>(acc, S) * in->f0;
        else leaf = in->aux;                                           // FT_OP_SETLEAF
    }
    outD = __shfl(acc, (int)(rank << lg), 64);                         // the ray of rank r was served by group r = lanes r*g ...
    outLeaf = leaf;
}

// ------------------------------------------------------------------------------------------------
// "Carved union" evaluator (FtSceneDev.fastPath == 3; ft_device.h FtCarve; round 4).
//
// The reference's own workload is subtract(intersect(union [1000 tori], [sphere]), sphere) (Program.fs:67-77).  For programs of that shape —
// ONE grid union of plain primitives, then at most FT_CARVE_TAIL single-primitive intersect / subtract steps — the interpreter is replaced
// by straight-line code: the union's running minimum and material stay in registers (no LDS value slots), the primitive kind K of the
// union's children is a template parameter (no type switch per candidate; K = FT_CARVE_MIXED keeps it), the tail's primitives have
// wave-uniform constants (scalar loads), and the candidate lists end in a terminator record (LowerBound = +inf), so the walk carries no
// end index.  Same operations on the same operands in the same order as ft_exec / eval_union_prims: the value is theirs bit for bit.
//
// Early exit ("lazy union", generalised from round 3).  With d_k <= b_k for every intersect step (b_k = the pruning bound |c_k - p| + r_k,
// SdfForm.fs:62), `if v < b_k then v = Max(v, d_k)` equals Max(v, d_k) as a VALUE (v >= b_k >= d_k leaves v = Max(v, d_k) too), so the
// tail maps the union's value U to max(U, m), m = the largest of the d_k / -d_k of the tail = the tail applied to -inf.  The walk's
// running minimum only falls and ends at U <= mn; once mn <= m the scene's value is m whatever the rest of the walk finds.  The lane stops
// there provided m > 0 and m >= the ray's epsilon (no hit at this point: the material the shortened walk reports is never read; a positive
// value has one bit pattern), d_k <= b_k holds in this lane, and nothing is NaN (every comparison below is false for a NaN).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ft_vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }   // IEEE minNum, -0 < +0
__device__ __forceinline__ float ft_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// MathF.Max (ft_max: NaN propagates, -0 < +0) without branches: v_max_f32 orders the zeros and returns the other operand for a NaN
__device__ __forceinline__ float ft_max_dev(float a, float b) {
    const float r = ft_vmax(a, b);
    return __builtin_isunordered(a, b) ? __builtin_nanf("") : r;
}
template <int K, bool FQ>
__device__ __forceinline__ float carve_child(uint32_t typeData, cfp consts, const f3 p) {
    cfp c = reinterpret_cast<cfp>(reinterpret_cast<const char FT_CONST*>(consts) + (size_t)(typeData >> 4));      // itemsT keeps a BYTE offset there
    if (K == (int)FT_PR_SPHERE) return prim_sphere<FQ>(c, p);
    if (K == (int)FT_PR_CAPSULE) return prim_capsule<FQ>(c, p);
    if (K == (int)FT_PR_TORUS) return prim_torus<FQ>(c, p);
    if (K == (int)FT_PR_TRIANGLE) return prim_triangle<FQ>(c, p);
    if (K == (int)FT_PR_BOX) return prim_box(c, p);
    return prim_eval_t<FQ>(typeData & 15u, c, p);
}
template <int K, bool FQ>
__device__ __forceinline__ void carve_walk(const FtSceneDev& S, const FtCarve& CV, const f3 p, const float cap, float& outD, uint32_t& outLeaf) {
    const FtGrid FT_CONST& g = as_const(S.grids)[0];
    const f3 cc = (p - mk3(g.aabbMin[0], g.aabbMin[1], g.aabbMin[2])) * mk3(g.cellSizeInv[0], g.cellSizeInv[1], g.cellSizeInv[2]);
    const int ix = ft_clamp_i(0, g.count[0] - 1, ft_floor_i(cc.x));
    const int iy = ft_clamp_i(0, g.count[1] - 1, ft_floor_i(cc.y));
    const int iz = ft_clamp_i(0, g.count[2] - 1, ft_floor_i(cc.z));
    const uint32_t cell = (uint32_t)((ix * g.count[1] + iy) * g.count[2] + iz);                  // the scene's only grid: cellBase = 0
    cfp ctr = as_const(S.cellCenters) + 3u * cell;
    const float distanceToCenter = ft_distance(mk3(ctr[0], ctr[1], ctr[2]), p);                   // SdfForm.fs:25
    const char FT_CONST* items = reinterpret_cast<const char FT_CONST*>(as_const(CV.itemsT));
    cfp consts = as_const(S.consts);
    uint32_t off = as_const(CV.cellStartT)[cell];
    auto rec = [&](uint32_t o) { return ld_item(reinterpret_cast<const FtItemRec FT_CONST*>(items + (size_t)o)); };
    // Items.[0]: evaluated unconditionally (SdfForm.fs:26; SdfObject.fs:34-43 starts its pick there too)
    float mn; uint32_t leaf;
    {
        const ItemRegs r0 = rec(off);
        mn = carve_child<K, FQ>(r0.b.y, consts, p); leaf = r0.b.z;
        off += 32u;
    }
    // candidate i >= 1 is evaluated iff  mn > LowerBound_i - distanceToCenter (:30)  and  mn > getMinDistance_i (:31); the list is sorted by
    // LowerBound and mn never grows, so the first failure of :30 ends the walk (see eval_union_prims) — at the latest on the terminator
    auto evaluate = [&](uint32_t typeData, uint32_t mat) {
        const float d = carve_child<K, FQ>(typeData, consts, p);
        if (d < mn) leaf = mat;                                        // SdfObject.fs:41-43
        mn = d != d ? d : ft_vmin(mn, d);                              // SdfForm.fs:33 (mn is no NaN here: it passed :30)
    };
    if (!(mn <= cap)) {
        for (;;) {
            ItemRegs ra = rec(off), rb = rec(off + 32u);
            asm volatile("" : "+v"(ra.a), "+v"(ra.b), "+v"(rb.a), "+v"(rb.b));      // both records requested before anything waits for either
            const float lbA = ra.a.x - distanceToCenter, lbB = rb.a.x - distanceToCenter;
            const float mdA = ft_dist<FQ>(mk3(ra.a.y, ra.a.z, ra.a.w), p) - __uint_as_float(ra.b.x);
            const float mdB = ft_dist<FQ>(mk3(rb.a.y, rb.a.z, rb.a.w), p) - __uint_as_float(rb.b.x);
            if (!(mn > lbA)) break;
            // Both candidates of a trip are consumed: after an evaluation of A, B is judged by the new minimum with the right-hand sides computed
            // above (they do not depend on it), so no record is requested twice.  Measured against round 3's one-evaluation-site walk (which resumes
            // at B after evaluating A) and against a walk that requests the next trip's records one trip ahead: profiles/r04_carved_variants.txt.
            if (mn > mdA) { evaluate(ra.b.y, ra.b.z); if (mn <= cap) break; }
            if (!(mn > lbB)) break;
            if (mn > mdB) { evaluate(rb.b.y, rb.b.z); if (mn <= cap) break; }
            off += 64u;
        }
    }
    outD = mn; outLeaf = leaf;
}

template <int K, bool COOP = false>
__device__ __forceinline__ void ft_eval_carved(const FtSceneDev& S, const FtCarve& CV, const f3 p, float& outD, uint32_t& outLeaf, float epsHit) {
    cfp consts = as_const(S.consts);
    const bool pointOk = fast_point_ok(p);                             // wave-uniform: p finite and |p|inf < 20000 in every lane that evaluates
    // the tail's primitives first: wave-uniform kinds and constants
    float t[FT_CARVE_TAIL], b[FT_CARVE_TAIL];
    float m = -__builtin_inff();                                       // the tail applied to -inf, as a value
    bool lazyOk = true;
#pragma unroll
    for (int k = 0; k < FT_CARVE_TAIL; ++k) {
        t[k] = 0.0f; b[k] = 0.0f;
        if ((uint32_t)k < CV.nTail) {
            const uint32_t op = CV.tail[k].op, type = CV.tail[k].type;
            if (pointOk && (type & FT_CARVE_FAST_SPHERE)) {            // one clamped fast root (bit-identical to sqrtf here: ft_sq) for the distance and the bound
                cfp c = consts + CV.tail[k].data;
                const float dist = ft_dist<true>(ld3(c), p);
                t[k] = dist - c[3];                                    // SdfForm.fs:129
                b[k] = dist + c[3];                                    // SdfForm.fs:62 getMaxDistance of the sphere's own boundary (SdfForm.fs:131-135)
            } else {
                t[k] = prim_eval(type & 15u, consts + CV.tail[k].data, p);
                if (op == FT_OP_ISECT_RUN) { cfp bd = consts + CV.tail[k].bound; b[k] = ft_distance(ld3(bd), p) + bd[3]; }   // SdfForm.fs:62
            }
            if (op == FT_OP_ISECT_RUN) {
                lazyOk = lazyOk && t[k] <= b[k];
                m = ft_vmax(m, t[k]);
            } else { lazyOk = lazyOk && t[k] == t[k]; m = ft_vmax(m, -t[k]); }
        }
    }
    const float cap = (lazyOk && m >= epsHit && m > 0.0f) ? m : -__builtin_inff();      // epsHit = +inf: never
    float v; uint32_t leaf;
    const bool fastOk = S.fastQ != 0u && pointOk;
    if (COOP) {
        // latency mode (see "Latency (tail) mode"): one query point, the same in all 64 lanes; the cell's candidates are spread over the lanes and the
        // reference's decisions replayed in list order (eval_union_coop: the walk runs to its end, which the early exit above never changes)
        float* none = ft_lds;                                           // no slot children in a carved scene: never dereferenced
        if (fastOk) eval_union_coop<true, false, 0>(S, as_const(S.grids)[0], p, none, reinterpret_cast<uint32_t*>(none), none, true, false, v, leaf);
        else eval_union_coop<false, false, 0>(S, as_const(S.grids)[0], p, none, reinterpret_cast<uint32_t*>(none), none, false, false, v, leaf);
    }
    else if (fastOk) carve_walk<K, true>(S, CV, p, cap, v, leaf);
    else carve_walk<K, false>(S, CV, p, cap, v, leaf);
#pragma unroll
    for (int k = 0; k < FT_CARVE_TAIL; ++k) {
        if ((uint32_t)k < CV.nTail) {
            if (CV.tail[k].op == FT_OP_ISECT_RUN) { if (v < b[k]) v = ft_max_dev(v, t[k]); }      // SdfForm.fs:60-63
            else v = ft_max_dev(-t[k], v);                                                        // SdfForm.fs:46-47
        }
    }
    outD = v; outLeaf = leaf;
}

// ------------------------------------------------------------------------------------------------
// render / trace kernel
// ------------------------------------------------------------------------------------------------
// phases >= PH_MARCH need one scene-SDF evaluation per round
// PH_CAM (round 4, FT_OPT_REUSE): the wave's very first round in Image.render mode — every lane evaluates the scene at the camera position, where every primary ray
// of the frame starts (Camera.fs:52: Origin = Position), so that each ray's first march step can be taken from that one value instead of evaluating it per ray
enum : uint32_t { PH_IDLE = 0, PH_DONE, PH_LIGHTS, PH_AONEXT, PH_MARCH, PH_NX, PH_NY, PH_NZ, PH_NC, PH_SHADOW, PH_AO, PH_CAM };

// EXTENSION: fixed ambient-occlusion directions (16 Fibonacci-sphere points); the ray k leaves the hit
// point along normalize(Normal + FT_AO_DIRS[k]).  Same table as the oracle's AO_DIRS.
__constant__ float FT_AO_DIRS[16][3] = {
    {0x1.02414ep-3f, -0x1.4c1e18p-2f, 0x1.e00000p-1f}, {-0x1.0bab14p-1f, 0x1.082252p-2f, 0x1.a00000p-1f},
    {0x1.64fce8p-1f, 0x1.9faf66p-3f, 0x1.600000p-1f}, {-0x1.b78ec2p-2f, -0x1.69cc1ap-1f, 0x1.200000p-1f},
    {-0x1.662d3ep-3f, 0x1.c39baap-1f, 0x1.c00000p-2f}, {0x1.88019ap-1f, -0x1.1fe15ep-1f, 0x1.400000p-2f},
    {-0x1.f3fa74p-1f, -0x1.b27cbep-4f, 0x1.800000p-3f}, {0x1.5150bap-1f, 0x1.7fd8c8p-1f, 0x1.000000p-4f},
    {0x1.51e2d4p-6f, -0x1.fee3d2p-1f, -0x1.000000p-4f}, {-0x1.5b4eb4p-1f, 0x1.6bbd02p-1f, -0x1.800000p-3f},
    {0x1.e54554p-1f, -0x1.03ffa2p-4f, -0x1.400000p-2f}, {-0x1.6781e0p-1f, -0x1.1f9d7cp-1f, -0x1.c00000p-2f},
    {0x1.046c0ap-3f, 0x1.a248a2p-1f, -0x1.200000p-1f}, {0x1.9bff54p-2f, -0x1.3585eap-1f, -0x1.600000p-1f},
    {-0x1.21c850p-1f, 0x1.1e0d66p-3f, -0x1.a00000p-1f}, {0x1.38c4f8p-2f, 0x1.55799ap-3f, -0x1.e00000p-1f}};

struct LaneState {
    uint32_t phase, job, steps, lidx, leaf, outIdx;
    f3 o, dir;            // current ray (primary, then the shadow ray of light lidx)
    float len, eps;
    // in LDS (ft_sh rows), not here: hp = result.Ray.Origin after Ray.move -eps (SdfObject.fs:73) = result.Position; nrm = the three probes
    // NX..NZ, afterwards the normal; lacc = lightColor (SdfScene.fs:12); lint = intensity the current light adds when unshadowed; lcos
    // EXTENSION state, one register (round 3 kept four: the EXTENSION build of the general kernel spilled three at its 96): bits 0-7 the ambient-occlusion
    // ray counter (<= 16), 8-15 the unoccluded count, 16-23 glass interactions so far (<= 64), 31 set = inside a glass body (the march runs on -Distance)
    uint32_t xs;
    f3 thr;                   // EXTENSION: path throughput (wavelength weight x tints)
    uint32_t seed;            // EXTENSION glass: per-sample hash seed
    __device__ __forceinline__ uint32_t aoIdx() const { return xs & 255u; }
    __device__ __forceinline__ uint32_t aoOpen() const { return (xs >> 8) & 255u; }
    __device__ __forceinline__ uint32_t bounce() const { return (xs >> 16) & 255u; }
    __device__ __forceinline__ bool inside() const { return (xs >> 31) != 0u; }
};

__device__ __forceinline__ void write_rgb(float* __restrict__ out, uint32_t idx, f3 c) {
    float* o = out + 3ull * idx;
    o[0] = c.x; o[1] = c.y; o[2] = c.z;
}

// EXTENSION builds scale every finished sample by the path throughput (exactly 1 unless glass / wavelengths are on)
template <bool EXT>
__device__ __forceinline__ void emit(const FtRenderArgs& a, const LaneState& s, f3 c) {
    write_rgb(a.out, s.outIdx, EXT ? c * s.thr : c);
}

// EXTENSION: lowbias32 of the sample seed and the bounce index (same function as the oracle's glass_hash)
__device__ __forceinline__ uint32_t ft_glass_hash(uint32_t seed, uint32_t bounce) {
    uint32_t h = seed + (bounce + 1u) * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

// ft_form_try_trace / ft_object_try_trace (modes 2 / 3, EXTENSION builds of the kernel only): results are
// 10 / 16 dwords per ray — ft_form_trace_result / ft_object_trace_result
__device__ __forceinline__ void write_try_trace_miss(const FtRenderArgs& a, const LaneState& s) {
    const uint32_t n = a.mode == 2u ? 10u : 16u;
    float* o = a.out + (size_t)n * s.outIdx;
    for (uint32_t i = 0; i < n; ++i) o[i] = 0.0f;
}
__device__ __forceinline__ void write_ray(float* o, f3 origin, f3 dir, float len, float eps) {
    o[0] = origin.x; o[1] = origin.y; o[2] = origin.z; o[3] = dir.x; o[4] = dir.y; o[5] = dir.z; o[6] = len; o[7] = eps;
}

// Can the ray o + t dir, t >= 0, still come within eps of the scene's support sphere (FtSceneDev.escC / escR, scene.cpp supportOf)?  If not,
// no evaluation along it can be below eps (SdfForm.fs:98), so its march ends in a miss whatever the steps are: the lane takes that exit at
// once (FT_OPT_ESCAPE).  Outside and heading away (the distance to the centre only grows), or outside and passing by (closest approach
// |w|^2 - b^2 / |dir|^2 beyond the radius); 4e-6 |w|^2 covers the rounding of the three dot products.  Any NaN: "may still come".
// Only for rays whose remaining march stays where float32 cannot overflow (Length < 1e9, |dir| < 1e6, distance from the sphere < 1e15): there
// every skipped evaluation is finite, so no NaN flag (SdfForm.fs: a NaN distance never terminates; flagged by the kernel and the oracle) is lost.
// The line is not what the reference evaluates: it accumulates the origin in float32 step by step (Ray.fs:9-13).  escR carries a padding for that
// drift, proved sufficient (scene.cpp "drift of the marched points") for 0 <= epsilon <= escR and either of two kinds of ray:
//   long rays   |dir| >= 1/2 and a start within sqrt(escRho2) of the centre (primary rays, directional shadow rays);
//   short rays  of any |dir| whose whole remaining march stays within 2.5 escR of the centre and has Length <= 10 escR: every step is >= g, so there are at
//               most Length / g + 1 <= 20 Rp / padDrift + 1 of them, each of error <= e(2.5 Rp) — the "near" budget of that bound.  (A point light's shadow
//               ray has |dir| = 1 / distance and Length = distance, SdfLight.fs:28-30: it travels one unit.)  2 (|w|^2 + |dir|^2 Length^2) >= (|w| + |dir| Length)^2.
// Any other ray simply marches on and is asked again at its next step.
__device__ __forceinline__ bool ft_never_enters(const FtSceneDev& S, const f3 o, const f3 dir, float eps, float len) {
    if (!(S.escR >= 0.0f) || !(len < 1e9f) || !(eps >= 0.0f) || !(eps <= S.escR)) return false;
    const f3 w = o - mk3(S.escC[0], S.escC[1], S.escC[2]);
    const float re = S.escR + eps;
    const float ww = ft_dot(w, w), cc = ww - re * re, tol = 4e-6f * ww, dd = ft_dot(dir, dir);
    if (!(cc > tol) || !(dd < 1e12f)) return false;                    // inside, or too close to tell
    const bool longRay = dd >= 0.25f && ww <= S.escRho2;
    const bool shortRay = len <= 9.9f * S.escR && 2.0f * (ww + dd * (len * len)) <= 6.2f * (S.escR * S.escR);
    if (!longRay && !shortRay) return false;                           // outside what the drift bound covers
    const float b = ft_dot(w, dir);
    if (b >= 0.0f) return true;
    return cc * dd - b * b > tol * dd;
}

// The first step of a ray that starts at the hit position, from the distance already known there (FT_SH_D0): exactly what the round's switch does with an
// evaluated distance for a PH_SHADOW / PH_AO lane (SdfForm.fs:94-104).  -> 0 the ray marches on (s.o, s.len, s.steps advanced), 1 it is a hit at once, 2 it
// ends as a miss (NaN distance, flagged like there).
__device__ __forceinline__ int first_step_from_cache(LaneState& s) {
    const float d = *ft_sh(FT_SH_D0);
    if (d != d) { ft_flag(1u); return 2; }                             // reference would never terminate
    if (d < s.eps) return 1;                                           // SdfForm.fs:98
    s.o = s.o + s.dir * d;                                             // Ray.move (Ray.fs:9-13)
    s.len = s.len - d;
    s.steps = 1;
    return 0;
}

// advance a lane until it needs an SDF evaluation (or is idle): everything in SdfScene.trace that
// is not a Distance call.
template <bool EXT>
__device__ __forceinline__ void settle(const FtRenderArgs& a, LaneState& s) {
    const float piInv = 1.0f / 3.14159274101257324f;                   // Math.fs:28-30
    for (;;) {
        if (s.phase == PH_MARCH) {
            // SdfForm.fs:94 -> SdfScene.fs:10; or every further step is known to miss (EXTENSION glass: a path inside a body marches on
            // -Distance, which is below epsilon everywhere outside the support sphere — the shortcut is for paths outside bodies only)
            if (s.len <= 0.0f || ((!EXT || !s.inside()) && ft_never_enters(a.S, s.o, s.dir, s.eps, s.len))) {
                if (EXT && a.mode >= 2u) write_try_trace_miss(a, s);   // ValueNone of the tryTrace entries
                else emit<EXT>(a, s, mk3(a.S.bg[0], a.S.bg[1], a.S.bg[2]));
                s.phase = PH_IDLE;
            }
            return;
        }
        if (EXT && s.phase == PH_AO) {                                 // EXTENSION
            if (s.len <= 0.0f) { s.xs += 0x101u; s.phase = PH_AONEXT; continue; }       // unoccluded: count + 1, next ray
            return;
        }
        if (EXT && s.phase == PH_AONEXT) {                             // EXTENSION
            if (s.aoIdx() >= a.aoSamples) {
                const float f = (float)s.aoOpen() / (float)a.aoSamples;
                sh_set3(FT_SH_LACC, mk3(a.S.bg[0], a.S.bg[1], a.S.bg[2]) * f);
                s.lidx = 0; s.phase = PH_LIGHTS;
                continue;
            }
            const f3 dir = ft_normalize(sh_get3(FT_SH_NRM) + mk3(FT_AO_DIRS[s.aoIdx()][0], FT_AO_DIRS[s.aoIdx()][1], FT_AO_DIRS[s.aoIdx()][2]));
            ft_count(FT_C_EXT);
            if (dir.x != dir.x || dir.y != dir.y || dir.z != dir.z) { s.xs += 0x101u; continue; }
            s.o = sh_get3(FT_SH_HP); s.dir = dir; s.len = a.aoRadius; s.steps = 0;
            s.phase = PH_AO;
            if (a.reuse != 0u) {
                const int r = first_step_from_cache(s);
                if (r == 1) { s.xs += 1u; s.phase = PH_AONEXT; }      // occluded at once
                else if (r == 2) s.len = -1.0f;                        // resolved as unoccluded by the PH_AO branch above, like a miss of the round's switch
            }
            continue;
        }
        if (s.phase == PH_SHADOW) {
            if (s.len <= 0.0f || ((!EXT || !s.inside()) && ft_never_enters(a.S, s.o, s.dir, s.eps, s.len))) {   // shadow ray missed (or can only miss): light arrives
                const FtLight L = ld_light(as_const(a.S.lights) + s.lidx);                 // the light this shadow ray was cast for (PH_LIGHTS below)
                const f3 lv = mk3(L.v[0], L.v[1], L.v[2]), hp = sh_get3(FT_SH_HP);
                f3 lint = mk3(L.color[0], L.color[1], L.color[2]), ldir = lv;             // SdfLight.fs:9, :16
                if (L.type != FT_LIGHT_DIRECTIONAL) { ldir = ft_normalize(lv - hp); lint = lint / ft_length2(lv - hp); }      // SdfLight.fs:25, :28, :40
                sh_set3(FT_SH_LACC, sh_get3(FT_SH_LACC) + lint * ft_dot(sh_get3(FT_SH_NRM), ldir));   // SdfScene.fs:15, :23
                s.lidx += 1; s.phase = PH_LIGHTS;
                continue;
            }
            return;
        }
        if (s.phase == PH_LIGHTS) {
            if (s.lidx >= a.S.nLights) {                               // SdfScene.fs:28
                cfp m = as_const(a.S.materials) + 3u * s.leaf;
                const f3 color = mk3(m[0], m[1], m[2]);
                emit<EXT>(a, s, color * (sh_get3(FT_SH_LACC) * piInv));
                s.phase = PH_IDLE;
                return;
            }
            const FtLight L = ld_light(as_const(a.S.lights) + s.lidx);
            const f3 lv = mk3(L.v[0], L.v[1], L.v[2]);
            f3 ldir;
            const f3 hp = sh_get3(FT_SH_HP);
            if (L.type == FT_LIGHT_DIRECTIONAL) ldir = lv;             // SdfLight.fs:9
            else ldir = ft_normalize(lv - hp);                         // SdfLight.fs:25
            const float lightCos = ft_dot(sh_get3(FT_SH_NRM), ldir);   // SdfScene.fs:15
            if (lightCos > 0.0f) {                                     // SdfScene.fs:17
                s.o = hp;
                if (L.type == FT_LIGHT_DIRECTIONAL) {                  // SdfLight.fs:11-16
                    s.dir = lv; s.len = 1000.0f;
                } else {                                               // SdfLight.fs:27-37
                    const f3 diff = lv - hp;
                    const float distance2 = ft_length2(diff);
                    s.dir = diff / distance2;                          // not unit: reference quirk
                    s.len = sqrtf(distance2);
                                                                       // intensity lc / distance2 (:40): formed when the ray has missed (PH_SHADOW above)
                }
                s.steps = 0; ft_count(FT_C_SHADOW);
                s.phase = PH_SHADOW;
                if (a.reuse != 0u) {
                    const int r = first_step_from_cache(s);
                    if (r == 1) { ft_count(FT_C_HITS); s.lidx += 1; s.phase = PH_LIGHTS; }     // shadowed at once (SdfLight.fs:20)
                    else if (r == 2) s.len = -1.0f;                    // resolved as a miss by the PH_SHADOW branch above
                }
                continue;
            }
            s.lidx += 1;
            continue;
        }
        return;
    }
}

template <bool EXT>
__device__ __forceinline__ void start_job(const FtRenderArgs& a, LaneState& s, const bool camKnown, const float dCam, const uint32_t leafCam) {
    if (a.mode >= 1) {                                                 // explicit ray buffer (SdfScene.trace scene ray; 2, 3: tryTrace entries)
        const ft_ray r = a.rays[s.job];
        s.o = mk3(r.origin.x, r.origin.y, r.origin.z);
        s.dir = mk3(r.direction.x, r.direction.y, r.direction.z);
        s.len = r.length; s.eps = r.epsilon;
        s.outIdx = s.job;
        if (EXT) { s.thr = splat3(1.0f); s.seed = s.job; }
    } else {                                                           // Image.render (Image.fs:28-34)
        const uint32_t smp = EXT ? s.job / a.jobsPerPlane : 0u, jp = s.job - smp * a.jobsPerPlane;   // EXTENSION: sample plane (0 for spp = 1)
        const uint32_t t = jp >> 6, i = jp & 63u;
        const uint32_t tx = t / a.tilesY, ty = t - tx * a.tilesY;
        const uint32_t cl = tx * 8u + (i >> 3), y = ty * 8u + (i & 7u);
        if (cl >= (uint32_t)a.nCols || y >= (uint32_t)a.H) { s.phase = PH_IDLE; return; }
        const uint32_t x = (uint32_t)a.x0 + (cl / a.stripeW) * (a.stripeW * a.stripeRanks) + a.stripeRank * a.stripeW + cl % a.stripeW;
        // Image.fs:20-23: position = x / max(W,H); sample offsets (smp % n)/n, (smp / n)/n are 0 for spp = 1
        const float ox = EXT ? (float)(smp % a.sppN) / (float)a.sppN : 0.0f, oy = EXT ? (float)(smp / a.sppN) / (float)a.sppN : 0.0f;
        const float px = EXT ? ((float)x + ox) / a.maxSize : (float)x / a.maxSize;
        const float py = EXT ? ((float)y + oy) / a.maxSize : (float)y / a.maxSize;
        const f3 fw = mk3(a.cam[3], a.cam[4], a.cam[5]), up = mk3(a.cam[6], a.cam[7], a.cam[8]), rt = mk3(a.cam[9], a.cam[10], a.cam[11]);
        s.o = mk3(a.cam[0], a.cam[1], a.cam[2]);
        s.dir = ft_normalize(fw + (px - 0.5f) * rt + (py - 0.5f) * up);       // Camera.fs:48-51
        s.len = a.length; s.eps = a.eps;
        s.outIdx = smp * a.planePixels + cl * (uint32_t)a.H + y;
        if (EXT) {                                                     // EXTENSION: wavelength weight, hash seed (global pixel)
            s.thr = splat3(1.0f);
            if (a.spectral != 0u) { const uint32_t bin = smp % a.spectral; s.thr = mk3(a.spec[bin][0], a.spec[bin][1], a.spec[bin][2]); }
            s.seed = x * 0x9E3779B1u + y * 0x85EBCA77u + smp * 0xC2B2AE3Du;
        }
    }
    if (EXT) s.xs &= 0xffffu;                                          // outside, no interaction yet
    s.steps = 0; ft_count(FT_C_PRIMARY);
    s.phase = PH_MARCH;
    if (camKnown && s.len > 0.0f) {                                    // the ray's first evaluation is Distance(camera position) (SdfForm.fs:94-96): known, see PH_CAM — what the round's
        if (dCam != dCam) { ft_flag(1u); s.len = -1.0f; }              // switch does with it for a PH_MARCH lane: NaN (flagged, a miss),
        else if (dCam < s.eps) { ft_count(FT_C_HITP); s.leaf = leafCam; s.phase = PH_NX; }   // a hit at the camera itself,
        else { s.o = s.o + s.dir * dCam; s.len = s.len - dCam; s.steps = 1; }               // or the first step (Ray.fs:9-13)
    }
    settle<EXT>(a, s);
}

// EXTENSION (BASELINE.json config 5): a path segment ended on leaf s.leaf with normal s.nrm at s.hp.  Glass leaf:
// reflect or refract (Fresnel terms after the reference's dead Light.fs:30-59, repaired — see the oracle's
// "EXTENSION ... glass" block for the definition this restates operation by operation) and march on.
__device__ __forceinline__ void glass_bounce(const FtRenderArgs& a, LaneState& s) {
    cfp mx = as_const(a.materialsExt) + 4u * s.leaf;
    const bool glass = mx[0] != 0.0f;
    const f3 black = mk3(0.0f, 0.0f, 0.0f);
    if (!glass) {
        if (s.inside()) { write_rgb(a.out, s.outIdx, black); s.phase = PH_IDLE; }       // diffuse seen from inside: absorbed
        return;
    }
    const f3 N = sh_get3(FT_SH_NRM), D = s.dir, hp = sh_get3(FT_SH_HP);
    if (s.bounce() >= a.maxBounces || N.x != N.x || N.y != N.y || N.z != N.z) { write_rgb(a.out, s.outIdx, black); s.phase = PH_IDLE; return; }
    float cosi = -ft_dot(N, D);
    if (!(cosi > 0.0f)) cosi = 0.0f;
    float n = mx[1];
    if (a.spectral != 0u) {
        const uint32_t smp = s.job / a.jobsPerPlane;
        n = mx[1] + mx[2] * a.spec[smp % a.spectral][3];
    }
    const float n1 = !s.inside() ? 1.0f : n, n2 = !s.inside() ? n : 1.0f;
    const float eta = n1 / n2;                                         // Light.fs:36
    const float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
    bool reflect = true;
    float cost = 0.0f;
    if (k >= 0.0f) {
        cost = sqrtf(k);
        float rs, rp;
        { const float p = n2 * cosi, q = n1 * cost, x = (p - q) / (p + q); rs = x * x; }   // Light.fs:41-45
        { const float p = n1 * cosi, q = n2 * cost, x = (p - q) / (p + q); rp = x * x; }   // Light.fs:47-51
        const float reflectance = 0.5f * (rs + rp);                    // Light.fs:53
        const float u = (float)(ft_glass_hash(s.seed, s.bounce()) >> 8) * (1.0f / 16777216.0f);
        reflect = u < reflectance;
    }
    ft_count(FT_C_EXT);
    if (reflect) {
        s.dir = ft_normalize(D + N * (2.0f * cosi));                   // Light.fs:56
        s.o = hp + N * (2.0f * s.eps);
    } else {
        s.dir = ft_normalize(D * eta + N * (eta * cosi - cost));       // Light.fs:58
        s.o = hp - N * (4.0f * s.eps);
        s.xs ^= 0x80000000u;
        if (s.inside()) { cfp t = as_const(a.S.materials) + 3u * s.leaf; s.thr = s.thr * mk3(t[0], t[1], t[2]); }
    }
    s.xs += 0x10000u;
    s.len = a.length; s.steps = 0;
    s.phase = PH_MARCH;
}

__device__ __forceinline__ unsigned long long wave_sum(uint32_t v) {
    unsigned long long x = v;
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

template <int VARIANT, bool EXT, int MATH = 0, int K = 0>
__device__ __forceinline__ void ft_trace_body(const FtRenderArgs& a) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // the first wave of block 0 reports the shader clock it ran at (statistics only); its start clocks wait in LDS, not in registers
    unsigned long long* clk0 = reinterpret_cast<unsigned long long*>(ft_lds + 32);
    if (blockIdx.x == 0 && tid == 0) { clk0[0] = clock64(); clk0[1] = wall_clock64(); }
    float* sd = ft_lds + FT_LDS_HDR_FLOATS + tid;
    uint32_t* sl = reinterpret_cast<uint32_t*>(ft_lds + FT_LDS_HDR_FLOATS + a.S.nSlots * FT_BLOCK) + tid;
    float* ldsC = ft_lds + FT_LDS_HDR_FLOATS + 2u * a.S.nSlots * FT_BLOCK; // staged constant pool ("SDF op stack" in LDS)
    for (uint32_t i = tid; i < a.S.nStage; i += FT_BLOCK) ldsC[i] = a.S.consts[i];
    if (MATH != 0 && tid < FT_LIBM_TAB_DOUBLES) const_cast<ft_u64*>(ft_libm_tab(a.S))[tid] = ft_libm_tab_g[tid];   // FT_OPT_MATH: glibc's tables
    if (tid < 32u) reinterpret_cast<uint32_t*>(ft_lds)[tid] = 0u;       // the per-wave statistics words
#ifdef FT_UNION_PROFILE
    for (uint32_t k = 0; k < FT_LDS_DBG_ROWS; ++k) reinterpret_cast<uint32_t*>(ft_lds)[FT_LDS_CNT_WORDS + tid + k * FT_BLOCK] = 0u;
#endif
    __syncthreads();

    // this wave's row (lean kernel: latency mode and culled children; other kernels: culled children of the scene's cull site) — a wave-uniform address
    float* coopRow = ft_lds + ft_coop_lds_offset(a.S, MATH != 0) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) * FT_CULL_ROW;
    uint32_t cullSkipped = 0, cullTotal = 0;                           // (child, ray) pairs the culling pass dropped / looked at (wave-uniform sums)
    uint32_t chunkNext = 0, chunkEnd = 0;                              // wave-uniform
    uint32_t waveEvals = 0;                                            // evaluation rounds of this wave (lane-utilisation statistic)
    uint32_t coopEvals = 0;                                            // evaluations done in latency mode (wave-uniform)
    uint32_t nEvals = 0;                                               // scene evaluations of this wave's rays (wave-uniform: summed per round)
    bool exhausted = false;
    LaneState s;
    s.phase = PH_IDLE; s.job = 0; s.steps = 0; s.lidx = 0; s.leaf = 0; s.outIdx = 0;
    s.o = s.dir = mk3(0, 0, 0);
    s.len = 0; s.eps = 0;
    s.xs = 0;
    s.thr = splat3(1.0f); s.seed = 0;
    // FT_OPT_REUSE: the first round of a wave in Image.render mode evaluates the scene at the camera position in every lane (PH_CAM)
    bool camKnown = false;
    float dCam = 0.0f; uint32_t leafCam = 0u;                          // wave-uniform
    if (a.reuse != 0u && a.mode == 0u) { s.phase = PH_CAM; s.o = mk3(a.cam[0], a.cam[1], a.cam[2]); s.eps = a.eps; }

    for (;;) {
        // ---- refill idle lanes from the wave's chunk ------------------------------------------
        for (int round = 0; round < 3; ++round) {
            const unsigned long long idle = __ballot(s.phase == PH_IDLE);
            if (idle == 0ull) break;
            // Burst refill: new rays are taken only when at least refillMin lanes are idle (or nothing is left to evaluate).  Rays
            // that start together on one 8x8 tile stay close in depth, so the lanes of a wave keep visiting the same lookup
            // cells and list positions: their record loads coalesce and their walks have similar lengths.  That is worth far
            // more to the grid-union kernels than the idle lanes cost (capi.cpp launchTrace picks refillMin per kernel).
            if ((uint32_t)__popcll(idle) < a.refillMin && __ballot(s.phase >= PH_MARCH) != 0ull) break;
            if (chunkNext == chunkEnd) {
                if (exhausted) break;
                // Guided hand-out at the end of the queue (FT_OPT_GUIDED, lean kernel, OFF by default; everywhere else shrink1 = shrink2 = nJobs):
                // from job shrink1 on a wave takes half a tile, from shrink2 on a quarter, so that the last generation of work is spread over
                // 2 - 4 times as many waves.  Measured: the 32 / 16-ray rounds of the latency mode cost +34 % / +46 % per ray, more than the
                // shorter drain returns (N = 8 share of C3 8.67 -> 9.13 ms).  The cursor is read first to pick the size.
                uint32_t base = 0, take = a.chunk;
                if (lane == 0) {
                    if (a.shrink1 < a.nJobs) {
                        const uint32_t cur = __hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        take = cur >= a.shrink2 ? a.chunk / 4u : (cur >= a.shrink1 ? a.chunk / 2u : a.chunk);
                    }
                    base = atomicAdd(a.counter, take);
                }
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t took = (uint32_t)__builtin_amdgcn_readfirstlane((int)take);
                if (base >= a.nJobs) { exhausted = true; break; }
                chunkNext = base;
                chunkEnd = (a.nJobs - base < took) ? a.nJobs : base + took;
            }
            const uint32_t avail = chunkEnd - chunkNext;
            const uint32_t nIdle = (uint32_t)__popcll(idle);
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (s.phase == PH_IDLE && rank < avail) { s.job = chunkNext + rank; start_job<EXT>(a, s, camKnown, dCam, leafCam); }
            chunkNext += (nIdle < avail) ? nIdle : avail;
        }
        if (s.phase == PH_IDLE && exhausted && chunkNext == chunkEnd) s.phase = PH_DONE;
        if (__ballot(s.phase != PH_DONE) == 0ull) break;

        // ---- one scene-SDF evaluation per active lane -----------------------------------------
        const bool active = s.phase >= PH_MARCH;
        waveEvals += 1;
        FT_UDBG_T0(tRound);
        auto query_point = [&]() {
            f3 q = s.o;
            if (s.phase >= PH_NX && s.phase <= PH_NC) {                // SdfForm.fs:106-115
                const f3 base = s.o + s.dir * (-s.eps);                // Ray.get (-eps)
                const float h = s.eps * 0.125f;
                q = base;
                if (s.phase == PH_NX) q.x = base.x + h;
                if (s.phase == PH_NY) q.y = base.y + h;
                if (s.phase == PH_NZ) q.z = base.z + h;
            }
            return q;
        };
        // ---- latency mode: at most tailK rays left in this wave -> each is evaluated by all 64 lanes together ("Latency (tail) mode") ----
        const unsigned long long am = __ballot(active);
        const bool camRound = __ballot(s.phase == PH_CAM) != 0ull;     // all lanes are in PH_CAM together; that one evaluation per wave is not counted in the statistics
        if (!camRound) nEvals += (uint32_t)__popcll(am);
        const bool coop = (uint32_t)__popcll(am) <= a.tailK;           // tailK = 0: never
        float dCoop = 0.0f; uint32_t leafCoop = 0;
        if (coop && am != 0ull) {
            const f3 qm = query_point();
            if (VARIANT == 1) {                                        // smooth union of spheres: all the rays at once, 64 / rays lanes each
                ft_eval_smooth_spheres_packed<MATH>(a.S, qm, active, am, (uint32_t)__popcll(am), ldsC, coopRow, dCoop, leafCoop);
                if (!camRound) coopEvals += (uint32_t)__popcll(am);
            } else {                                                   // general scenes: one ray after the other, all 64 lanes each
                unsigned long long m = am;
                while (m != 0ull) {
                    const int L = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    const f3 qL = ft_readlane3(qm, L);
                    float dL; uint32_t leafL;
                    if (VARIANT == 3) ft_eval_carved<K, true>(a.S, a.carve, qL, dL, leafL, __builtin_inff());
                    else ft_eval_coop<VARIANT == 2, MATH>(a.S, qL, sd, sl, ldsC, dL, leafL);
                    if ((int)lane == L) { dCoop = dL; leafCoop = leafL; }
                    if (!camRound) coopEvals += 1;
                }
            }
        }
        // ---- lean kernel: drop the children whose terms no ray of this wave can feel this round ("Exact child culling") ----
        uint32_t cullN = FT_CULL_NONE;
        if (VARIANT != 3 && a.cull != 0u && !coop && am != 0ull) {
            cullN = ft_cull_children(a.S, query_point(), active, am, ldsC, coopRow);
            if (cullN != FT_CULL_NONE) {
                const uint32_t cnt = (as_const(a.S.instr) + a.S.cullPc)->count;
                const uint32_t looked = cnt < FT_CULL_MAX ? cnt : FT_CULL_MAX;
                cullTotal += looked * (uint32_t)__popcll(am) >> 6; cullSkipped += (looked - (cullN & 0xffffu)) * (uint32_t)__popcll(am) >> 6;   // in units of 64 pairs
            }
        }
        if (active) {
            float d; uint32_t leaf;
            FT_UDBG_T0(tEval);
            if (coop) { d = dCoop; leaf = leafCoop; }
            else {
                const f3 q = query_point();
                if (VARIANT == 1) ft_eval_smooth_spheres<MATH>(a.S, q, ldsC, d, leaf, coopRow, cullN);
                else if (VARIANT == 3) ft_eval_carved<K>(a.S, a.carve, q, d, leaf, a.lazy == 0u ? __builtin_inff() : s.eps);
                else ft_eval<VARIANT == 2, MATH>(a.S, q, sd, sl, ldsC, d, leaf,          // lazy unions: off (+inf) inside a glass body, whose exit is a "hit" at large values
                                                  (EXT && s.inside()) || a.lazy == 0u ? __builtin_inff() : s.eps, coopRow, cullN);
            }
            FT_UDBG_T1(5, tEval); FT_UDBG_WAVE(6);
            if (EXT) d = __uint_as_float(__float_as_uint(d) ^ (s.xs & 0x80000000u));   // EXTENSION glass: inside, march on -Distance

            switch (s.phase) {
            case PH_MARCH:
            case PH_SHADOW:
            case PH_AO: {
                bool miss = false;
                if (d != d) { ft_flag(1u); miss = true; }           // reference would never terminate
                else if (d < s.eps) {                                  // SdfForm.fs:98
                    if (s.phase == PH_MARCH) {
                        ft_count(FT_C_HITP); s.leaf = leaf; s.phase = PH_NX;
                        if (EXT && a.mode == 2u) {                     // SdfForm.tryTrace: {Ray = ray; Distance = distance} (SdfForm.fs:98-102)
                            float* o = a.out + 10ull * s.outIdx;
                            write_ray(o, s.o, s.dir, s.len, s.eps);
                            o[8] = d; reinterpret_cast<int32_t*>(o)[9] = 1;
                            s.phase = PH_IDLE;
                        }
                    }
                    else if (s.phase == PH_SHADOW) { ft_count(FT_C_HITS); s.lidx += 1; s.phase = PH_LIGHTS; }   // shadowed (SdfLight.fs:20)
                    else { s.xs += 1u; s.phase = PH_AONEXT; }          // EXTENSION: occluded
                } else {
                    s.o = s.o + s.dir * d;                             // Ray.move (Ray.fs:9-13)
                    s.len = s.len - d;
                    s.steps += 1;
                    if (s.steps >= FT_STEP_CAP) { ft_flag(4u); miss = true; }
                }
                if (miss) s.len = -1.0f;                               // resolved as a miss by settle()
                break;
            }
            case PH_CAM:                                               // all 64 lanes, the same point: the value every primary ray's first step is taken from
                dCam = ft_readlane_f(d, 0); leafCam = (uint32_t)__builtin_amdgcn_readlane((int)leaf, 0); camKnown = true;
                s.phase = PH_IDLE;
                break;
            case PH_NX: *ft_sh(FT_SH_NRM) = d; s.phase = PH_NY; break;
            case PH_NY: *ft_sh(FT_SH_NRM + 1) = d; s.phase = PH_NZ; break;
            case PH_NZ: *ft_sh(FT_SH_NRM + 2) = d; s.phase = PH_NC; break;
            case PH_NC: {
                *ft_sh(FT_SH_D0) = d;                                  // D at the hit position: the first evaluation of every ray that starts there
                const f3 nrm = ft_normalize(sh_get3(FT_SH_NRM) - splat3(d));   // SdfForm.fs:107-112
                const f3 hp = s.o + s.dir * (-s.eps);                  // SdfObject.fs:73
                sh_set3(FT_SH_NRM, nrm);
                sh_set3(FT_SH_HP, hp);
                sh_set3(FT_SH_LACC, mk3(a.S.bg[0], a.S.bg[1], a.S.bg[2]));     // SdfScene.fs:12
                s.lidx = 0;
                s.phase = PH_LIGHTS;
                if (EXT && a.aoSamples != 0u) { s.xs &= 0xffff0000u; s.phase = PH_AONEXT; }   // EXTENSION: AO counters to 0
                if (EXT && a.maxBounces != 0u) glass_bounce(a, s);     // EXTENSION
                if (EXT && a.mode == 3u) {                             // SdfObject.tryTrace result (SdfObject.fs:72-77)
                    float* o = a.out + 16ull * s.outIdx;
                    write_ray(o, hp, s.dir, s.len - (-s.eps), s.eps);              // Ray.move -eps: Length - (-eps)
                    cfp m = as_const(a.S.materials) + 3u * s.leaf;
                    o[8] = nrm.x; o[9] = nrm.y; o[10] = nrm.z; o[11] = m[0]; o[12] = m[1]; o[13] = m[2];
                    reinterpret_cast<int32_t*>(o)[14] = 1; o[15] = 0.0f;
                    s.phase = PH_IDLE;
                }
                break;
            }
            default: break;
            }
            settle<EXT>(a, s);
        }
        FT_UDBG_T1(7, tRound);
    }

    // ---- statistics -------------------------------------------------------------------------
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(ft_lds) + (tid >> 6);      // this wave's statistics words
    const unsigned long long e = nEvals, sh = cw[FT_C_SHADOW * 4], hp = cw[FT_C_HITP * 4], hs = cw[FT_C_HITS * 4], pr = cw[FT_C_PRIMARY * 4], ex = cw[FT_C_EXT * 4];
    const unsigned long long fl = cw[FT_C_FLAGS * 4] & 3u, fc = cw[FT_C_FLAGS * 4] & 4u;
    if (lane == 0) {
        atomicAdd(&a.stats->sdf_evals, e);
        atomicAdd(&a.stats->rays_shadow, sh);
        atomicAdd(&a.stats->hits_primary, hp);
        atomicAdd(&a.stats->hits_shadow, hs);
        atomicAdd(&a.stats->rays_primary, pr);
        if (ex) atomicAdd(&a.stats->rays_ext, ex);
        atomicAdd(&a.stats->wave_evals, (unsigned long long)waveEvals);
        if (coopEvals) atomicAdd(&a.stats->coop_evals, (unsigned long long)coopEvals);
        if (cullTotal) { atomicAdd(&a.stats->cull_total, (unsigned long long)cullTotal); atomicAdd(&a.stats->cull_skipped, (unsigned long long)cullSkipped); }
        if (fl | fc) atomicOr(&a.stats->flags, fl | fc);
        if (blockIdx.x == 0 && tid == 0) { atomicAdd(&a.stats->clk_shader, clock64() - clk0[0]); atomicAdd(&a.stats->clk_ref, wall_clock64() - clk0[1]); }
    }
#ifdef FT_UNION_PROFILE
    for (uint32_t k = 0; k < 12; ++k) {
        const unsigned long long v = wave_sum(reinterpret_cast<const uint32_t*>(ft_lds)[FT_LDS_CNT_WORDS + tid + k * FT_BLOCK]);
        if (lane == 0 && v) atomicAdd(&ft_union_dbg[k], v);
    }
#endif
}

// Occupancy hints.  The general kernels are bound by the latency of their dependent steps (grid-union walk: time falls almost in
// proportion to the resident waves up to 5 per SIMD), so the register allocator is asked for one wave more
// than it would settle on by itself where that costs (almost) no spills: 6 waves (80 VGPRs) for the plain kernel (1000-torus scene
// 23.5 -> 22.4 ms; 7 waves spill and are slower), 5 waves (96 VGPRs) for its EXTENSION build (C2 + AO 16.9 -> 14.9 ms) and for the
// kernel with on-demand sub-programs (24.7 -> 24.1 ms).
#ifndef FT_GENERAL_WAVES
#define FT_GENERAL_WAVES 6
#endif
#define FT_OCC(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#ifndef FT_CALLS_WAVES
#define FT_CALLS_WAVES 5
#endif
#ifndef FT_EXT_WAVES
#define FT_EXT_WAVES 5
#endif
#define FT_CALLS_OCC FT_OCC(FT_CALLS_WAVES)
#define FT_EXT_OCC FT_OCC(FT_EXT_WAVES)
// general scenes
extern "C" __global__ void __launch_bounds__(FT_BLOCK) FT_OCC(FT_GENERAL_WAVES) ft_trace_kernel(const FtRenderArgs a) { ft_trace_body<0, false>(a); }
// scenes that are one smooth union of spheres (BASELINE.json config 3/4)
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_smooth_spheres(const FtRenderArgs a) { ft_trace_body<1, false>(a); }
// EXTENSION builds of both (spp > 1 and / or ambient occlusion); the reference path never pays for them
extern "C" __global__ void __launch_bounds__(FT_BLOCK) FT_EXT_OCC ft_trace_kernel_ext(const FtRenderArgs a) { ft_trace_body<0, true>(a); }
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_smooth_spheres_ext(const FtRenderArgs a) { ft_trace_body<1, true>(a); }
// general scenes whose unions have combinator children evaluated on demand (FT_PR_CALL, FtSceneDev.fastPath == 2)
extern "C" __global__ void __launch_bounds__(FT_BLOCK) FT_CALLS_OCC ft_trace_kernel_calls(const FtRenderArgs a) { ft_trace_body<2, false>(a); }
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_calls_ext(const FtRenderArgs a) { ft_trace_body<2, true>(a); }
// scenes that are one grid union of plain primitives of kind K with at most two intersect / subtract steps behind it (ft_device.h "Carved union":
// the reference's own Program.fs scene is the torus one).  No exponential anywhere: no *_libm twins; EXTENSION launches take ft_trace_kernel_ext.
// Resident waves per SIMD asked of the register allocator: 6 (80 VGPRs) where the inlined primitive fits without spills — 7 (72) measured the same
// and leaves no register to spare — and 5 (96) for triangles and the type switch of mixed kinds (93 registers; at 6 and more they spill into the walk).
// Measured: profiles/r04_carved_variants.txt.
#define FT_CARVE_KERNEL(name, kind, waves) extern "C" __global__ void __launch_bounds__(FT_BLOCK) FT_OCC(waves) name(const FtRenderArgs a) { ft_trace_body<3, false, 0, (int)(kind)>(a); }
FT_CARVE_KERNEL(ft_trace_kernel_carved_spheres, FT_PR_SPHERE, 6)
FT_CARVE_KERNEL(ft_trace_kernel_carved_capsules, FT_PR_CAPSULE, 6)
FT_CARVE_KERNEL(ft_trace_kernel_carved_tori, FT_PR_TORUS, 6)
FT_CARVE_KERNEL(ft_trace_kernel_carved_triangles, FT_PR_TRIANGLE, 5)
FT_CARVE_KERNEL(ft_trace_kernel_carved_mixed, FT_CARVE_MIXED, 5)
static const void* ft_carved_kernel(unsigned kind) {
    switch (kind) {
        case FT_PR_SPHERE: return (const void*)ft_trace_kernel_carved_spheres;
        case FT_PR_CAPSULE: return (const void*)ft_trace_kernel_carved_capsules;
        case FT_PR_TORUS: return (const void*)ft_trace_kernel_carved_tori;
        case FT_PR_TRIANGLE: return (const void*)ft_trace_kernel_carved_triangles;
        default: return (const void*)ft_trace_kernel_carved_mixed;     // boxes (EXTENSION) and mixed kinds
    }
}
// FT_OPT_MATH = glibc: the same six with MathF.Exp / Log as glibc's expf / logf (scenes that contain a unionSmooth only; every other scene
// has no exponential and runs the kernels above whatever the option says)
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_libm(const FtRenderArgs a) { ft_trace_body<0, false, 1>(a); }
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_smooth_spheres_libm(const FtRenderArgs a) { ft_trace_body<1, false, 1>(a); }
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_ext_libm(const FtRenderArgs a) { ft_trace_body<0, true, 1>(a); }
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_smooth_spheres_ext_libm(const FtRenderArgs a) { ft_trace_body<1, true, 1>(a); }
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_calls_libm(const FtRenderArgs a) { ft_trace_body<2, false, 1>(a); }
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_trace_kernel_calls_ext_libm(const FtRenderArgs a) { ft_trace_body<2, true, 1>(a); }

// scene.Object.Form.Distance at explicit points (test / diagnostic entry)
template <int MATH>
__device__ __forceinline__ void ft_eval_points_body(const FtSceneDev& S, const float* __restrict__ pts, long long n, float* __restrict__ outD, int* __restrict__ outM) {
    const uint32_t tid = threadIdx.x;
    float* sd = ft_lds + FT_LDS_HDR_FLOATS + tid;                  // same LDS layout as the trace kernel (flag words first, unused here)
    uint32_t* sl = reinterpret_cast<uint32_t*>(ft_lds + FT_LDS_HDR_FLOATS + S.nSlots * FT_BLOCK) + tid;
    float* ldsC = ft_lds + FT_LDS_HDR_FLOATS + 2u * S.nSlots * FT_BLOCK;
    for (uint32_t i = tid; i < S.nStage; i += FT_BLOCK) ldsC[i] = S.consts[i];
    if (MATH != 0 && tid < FT_LIBM_TAB_DOUBLES) const_cast<ft_u64*>(ft_libm_tab(S))[tid] = ft_libm_tab_g[tid];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * FT_BLOCK + tid; i < n; i += (long long)gridDim.x * FT_BLOCK) {
        float d; uint32_t leaf;
        ft_eval<true, MATH>(S, mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), sd, sl, ldsC, d, leaf);
        outD[i] = d;
        if (outM) outM[i] = (int)leaf;
    }
}
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_eval_points_kernel(const FtSceneDev S, const float* __restrict__ pts,
                                                                            long long n, float* __restrict__ outD, int* __restrict__ outM) {
    ft_eval_points_body<0>(S, pts, n, outD, outM);
}
extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_eval_points_kernel_libm(const FtSceneDev S, const float* __restrict__ pts,
                                                                                 long long n, float* __restrict__ outD, int* __restrict__ outM) {
    ft_eval_points_body<1>(S, pts, n, outD, outM);
}

// device math primitives, for bit-parity tests against the oracle
extern "C" __global__ void ft_math_kernel(int op, const float* __restrict__ x, const float* __restrict__ y, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        float r;
        switch (op) {
            case 0: r = ft_exp(v); break;
            case 1: r = ft_log(v); break;
            case 2: r = sqrtf(v); break;
            case 4: r = ft_sqrt_fast(v); break;
            case 5: r = ft_exp_fast<false>(v); break;
            case 6: r = ft_glibc_expf<true>(v, ft_libm_tab_g); break;      // glibc restatements (ft_libm.h): FMA build ...
            case 7: r = ft_glibc_expf<false>(v, ft_libm_tab_g); break;     // ... SSE2 build
            case 8: r = ft_glibc_logf<true>(v, ft_libm_tab_g); break;
            case 9: r = ft_glibc_logf<false>(v, ft_libm_tab_g); break;
            case 10: r = ft_glibc_powf<true>(v, y[i], ft_libm_tab_g); break;
            case 11: r = ft_glibc_powf<false>(v, y[i], ft_libm_tab_g); break;
            case 12: r = ft_pow(v, y[i]); break;
            case 13: r = ft_max_dev(v, y[i]); break;                         // the carved kernels' MathF.Max
            case 14: r = y[i] != y[i] ? y[i] : ft_vmin(v, y[i]); break;      // ... and their MathF.Min (first operand never NaN there)
            default: r = v / y[i]; break;
        }
        out[i] = r;
    }
}

// EXTENSION (spp > 1): pixel = (sample 0 + sample 1 + ... in order) / spp, per float, fixed order
extern "C" __global__ void ft_resolve_kernel(const float* __restrict__ planes, float* __restrict__ out, unsigned long long nFloats, unsigned spp) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nFloats; i += (unsigned long long)gridDim.x * blockDim.x) {
        float c = planes[i];
        for (unsigned k = 1; k < spp; ++k) c = c + planes[(unsigned long long)k * nFloats + i];
        out[i] = c / (float)spp;
    }
}

// ------------------------------------------------------------------------------------------------
// Tone map on the device: Image.toColors (Image.fs:37-50) + FColor.gammaInverse / toColor
// (FColor.fs:43-55) + the scan-line order of Image.toBitmap (Image.fs:61-86).  HBM-bound: 12 B/pixel read twice
// (max pass, map pass; the second read mostly hits the L2 / MALL for frames up to a few hundred MB), 3 B/pixel
// written; the frame leaves the GPU as 3 bytes per pixel instead of 12.
// ------------------------------------------------------------------------------------------------
// pass 1: Image.fs:40-43, max = Max(0.01, max over pixels of getMaxColor).  getMaxColor is Max(Z, Max(Y, X)) with MathF.Max, which
// propagates NaN (Math.fs:83), and Seq.max / Array.max (Array2D.fs:45-50) keep `acc` unless `curr > acc`: a pixel with a NaN channel
// is skipped WHOLE — also when another of its channels would have been the global maximum.  (In the reference that holds for a
// NaN pixel anywhere but at the head of a column or of the frame, where NaN sticks to `acc` instead; and any NaN channel then makes
// Color.FromArgb throw in FColor.toColor.  Oracle and kernel take the position-independent rule)
// Every candidate is >= 0.01 > 0, so unsigned integer comparison of the bit patterns orders them.
__device__ __forceinline__ float tonemap_pixel_max(float m, float x, float y, float z) {
    const bool nan = x != x || y != y || z != z;
    const float pm = __builtin_fmaxf(z, __builtin_fmaxf(y, x));
    return nan ? m : __builtin_fmaxf(m, pm);
}
extern "C" __global__ void __launch_bounds__(256) ft_tonemap_max_kernel(const float* __restrict__ frame, unsigned long long nFloats, uint32_t* __restrict__ maxBits) {
    float m = 0.01f;
    const unsigned long long nPix = nFloats / 3ull, n4 = nPix / 4ull;          // groups of 4 pixels = 3 float4 (the frame is 16-byte aligned)
    const float4* f4 = reinterpret_cast<const float4*>(frame);
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float4 a = f4[3ull * i], b = f4[3ull * i + 1ull], c = f4[3ull * i + 2ull];
        m = tonemap_pixel_max(m, a.x, a.y, a.z);
        m = tonemap_pixel_max(m, a.w, b.x, b.y);
        m = tonemap_pixel_max(m, b.z, b.w, c.x);
        m = tonemap_pixel_max(m, c.y, c.z, c.w);
    }
    for (unsigned long long i = n4 * 4ull + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nPix; i += (unsigned long long)gridDim.x * blockDim.x)
        m = tonemap_pixel_max(m, frame[3ull * i], frame[3ull * i + 1ull], frame[3ull * i + 2ull]);
    for (int off = 32; off > 0; off >>= 1) m = __builtin_fmaxf(m, __shfl_down(m, off, 64));
    __shared__ float part[4];
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = __builtin_fmaxf(__builtin_fmaxf(part[0], part[1]), __builtin_fmaxf(part[2], part[3]));
        atomicMax(maxBits, __float_as_uint(m));
    }
}

// MathF.Pow: math = 0 the fixed algorithm (ft_math.h), 1 / 2 glibc's powf, FMA / SSE2 build (ft_libm.h, tables in LDS at `tab`)
__device__ __forceinline__ float ft_pow_m(float x, float g, int math, const ft_u64* tab) {
    if (math == 0) return ft_pow(x, g);
    return math == 1 ? ft_glibc_powf<true>(x, g, tab) : ft_glibc_powf<false>(x, g, tab);
}
__device__ __forceinline__ uint32_t tonemap_pixel(const float* __restrict__ px, float mx, float gammaInv, uint32_t dither, uint32_t seed, uint32_t x, uint32_t y,
                                                  int math, const ft_u64* tab) {
    // fcolor / max |> gammaInverse gammaInv |> toColor rng  (Image.fs:47-49): R, G, B in the order the reference draws its noise
    const float r = ft_pow_m(px[0] / mx, gammaInv, math, tab), g = ft_pow_m(px[1] / mx, gammaInv, math, tab), b = ft_pow_m(px[2] / mx, gammaInv, math, tab);
    const float ur = dither ? ft_dither_u(x, y, 0u, seed) : 0.5f, ug = dither ? ft_dither_u(x, y, 1u, seed) : 0.5f, ub = dither ? ft_dither_u(x, y, 2u, seed) : 0.5f;
    return ft_to_byte(r, ur) | (ft_to_byte(g, ug) << 8) | (ft_to_byte(b, ub) << 16);
}

// pass 2, Color[X,Y] order: out[(x * Y + y) * 3 + {0,1,2}] = R, G, B  (the value of Image.toColors)
extern "C" __global__ void __launch_bounds__(256) ft_tonemap_map_kernel(const float* __restrict__ frame, uint32_t X, uint32_t Y, const uint32_t* __restrict__ maxBits,
                                                                       float gammaInv, uint32_t dither, uint32_t seed, unsigned char* __restrict__ out, int math) {
    __shared__ ft_u64 tab[FT_LIBM_TAB_DOUBLES];
    if (threadIdx.x < FT_LIBM_TAB_DOUBLES) tab[threadIdx.x] = ft_libm_tab_g[threadIdx.x];
    __syncthreads();
    const float mx = __uint_as_float(*maxBits);
    const unsigned long long n = (unsigned long long)X * Y;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(i / Y), y = (uint32_t)(i - (unsigned long long)x * Y);
        const uint32_t c = tonemap_pixel(frame + 3ull * i, mx, gammaInv, dither, seed, x, y, math, tab);
        unsigned char* o = out + 3ull * i;
        o[0] = (unsigned char)c; o[1] = (unsigned char)(c >> 8); o[2] = (unsigned char)(c >> 16);
    }
}

// pass 2, bitmap order (Image.toBitmap, Image.fs:61-86: after the index arithmetic and Array.rev the scan0 buffer holds, at
// row r from the top and column c, the pixel image[X-1-c, r] as bytes B, G, R; stride X * 3).  A 64 x 64 tile is read with
// y fastest (contiguous in the frame), turned in LDS and written with c fastest (contiguous in the bitmap).
#define FT_TM_TILE 64
extern "C" __global__ void __launch_bounds__(256) ft_tonemap_bmp_kernel(const float* __restrict__ frame, uint32_t X, uint32_t Y, const uint32_t* __restrict__ maxBits,
                                                                       float gammaInv, uint32_t dither, uint32_t seed, unsigned char* __restrict__ out, int math) {
    __shared__ uint32_t tile[FT_TM_TILE][FT_TM_TILE + 1];
    __shared__ ft_u64 tab[FT_LIBM_TAB_DOUBLES];
    if (threadIdx.x < FT_LIBM_TAB_DOUBLES) tab[threadIdx.x] = ft_libm_tab_g[threadIdx.x];
    __syncthreads();
    const float mx = __uint_as_float(*maxBits);
    const uint32_t tilesY = (Y + FT_TM_TILE - 1) / FT_TM_TILE;
    const uint32_t x0 = (blockIdx.x / tilesY) * FT_TM_TILE, y0 = (blockIdx.x % tilesY) * FT_TM_TILE;
    for (uint32_t k = threadIdx.x; k < FT_TM_TILE * FT_TM_TILE; k += 256) {
        const uint32_t lx = k / FT_TM_TILE, ly = k % FT_TM_TILE, x = x0 + lx, y = y0 + ly;
        if (x < X && y < Y) tile[lx][ly] = tonemap_pixel(frame + 3ull * ((unsigned long long)x * Y + y), mx, gammaInv, dither, seed, x, y, math, tab);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < FT_TM_TILE * FT_TM_TILE; k += 256) {
        const uint32_t ly = k / FT_TM_TILE, lc = k % FT_TM_TILE;             // consecutive threads: consecutive bitmap columns
        const uint32_t lx = FT_TM_TILE - 1 - lc, x = x0 + lx, y = y0 + ly;
        if (x < X && y < Y) {
            const uint32_t c = tile[lx][ly];
            unsigned char* o = out + 3ull * ((unsigned long long)y * X + (X - 1u - x));
            o[0] = (unsigned char)(c >> 16); o[1] = (unsigned char)(c >> 8); o[2] = (unsigned char)c;     // B, G, R
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Device-side SdfBoundary.buildSpatialLookup, per-cell part (SdfBoundary.fs:245-274).
// One workgroup per cell: distances to all item boundaries (same float operations as the host loop),
// upperBound = min(getMaxDistance) + |cellSize/2|, keep items with getMinDistance < upperBound, sort by
// (LowerBound, item index) — the host's stable sort by LowerBound — with a bitonic network in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long grid_key(float lo, uint32_t idx) {
    uint32_t u = __float_as_uint(lo);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                    // order-preserving map of float to uint
    return ((unsigned long long)u << 32) | idx;
}

extern "C" __global__ void __launch_bounds__(FT_BLOCK) ft_grid_build_kernel(const FtGridBuildArgs a) {
    __shared__ unsigned long long keys[FT_GRID_BUILD_MAX_ITEMS];
    __shared__ float red[FT_BLOCK];
    __shared__ uint32_t count;
    const uint32_t tid = threadIdx.x, cell = blockIdx.x;
    const uint32_t z = cell % a.c, y = (cell / a.c) % a.c, x = cell / (a.c * a.c);
    const f3 cs = mk3(a.cellSize[0], a.cellSize[1], a.cellSize[2]);
    const f3 center = mk3(a.aabbMin[0], a.aabbMin[1], a.aabbMin[2]) + cs * 0.5f + cs * mk3((float)x, (float)y, (float)z);   // :246
    if (tid == 0) { a.centers[3 * cell] = center.x; a.centers[3 * cell + 1] = center.y; a.centers[3 * cell + 2] = center.z; count = 0; }
    float m = INFINITY;
    for (uint32_t i = tid; i < a.n; i += FT_BLOCK) {
        const float* b = a.bounds + 4 * i;
        const float v = ft_distance(mk3(b[0], b[1], b[2]), center) + b[3];          // getMaxDistance (:251)
        if (v < m) m = v;
    }
    red[tid] = m;
    __syncthreads();
    for (uint32_t s = FT_BLOCK / 2; s > 0; s >>= 1) { if (tid < s && red[tid + s] < red[tid]) red[tid] = red[tid + s]; __syncthreads(); }
    const float upperBound = red[0] + a.halfDiag;                                    // :253
    bool nan = false;
    for (uint32_t i = tid; i < a.n; i += FT_BLOCK) {
        const float* b = a.bounds + 4 * i;
        const float lo = ft_distance(mk3(b[0], b[1], b[2]), center) - b[3];         // getMinDistance (:257)
        nan |= lo != lo;
        if (lo < upperBound) keys[atomicAdd(&count, 1u)] = grid_key(lo, i);          // order fixed by the sort below
    }
    if (nan) atomicOr(a.flags, 1u);
    __syncthreads();
    const uint32_t cnt = count;
    if (cnt == 0) { if (tid == 0) { atomicOr(a.flags, 2u); a.counts[cell] = 0; } return; }
    uint32_t pow2 = 1; while (pow2 < cnt) pow2 <<= 1;
    for (uint32_t i = cnt + tid; i < pow2; i += FT_BLOCK) keys[i] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= pow2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < pow2; i += FT_BLOCK) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long p = keys[i], q = keys[l];
                    if (((i & k) == 0) ? (p > q) : (p < q)) { keys[i] = q; keys[l] = p; }
                }
            }
            __syncthreads();
        }
    FtItem* out = a.tmp + (size_t)cell * a.n;
    for (uint32_t i = tid; i < cnt; i += FT_BLOCK) {
        const unsigned long long kx = keys[i];
        uint32_t u = (uint32_t)(kx >> 32);
        u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
        out[i].lowerBound = __uint_as_float(u);
        out[i].child = (uint32_t)kx;
    }
    if (tid == 0) a.counts[cell] = cnt;
}

extern "C" __global__ void ft_grid_compact_kernel(const FtItem* __restrict__ tmp, const uint32_t* __restrict__ cellStart, uint32_t n, FtItem* __restrict__ items) {
    const uint32_t cell = blockIdx.x, beg = cellStart[cell], cnt = cellStart[cell + 1] - beg;
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) items[beg + i] = tmp[(size_t)cell * n + i];
}

// multi-GPU: gathered slabs [rank][stripe j][S columns] -> frame [stripe j][rank][S columns] (ft_render_multi; the Python path
// does the same with one strided torch copy).  One stripe = stripeVec float4 (or float) elements, contiguous on both sides.
template <class T>
__global__ void ft_deinterleave_kernel(const T* __restrict__ recv, T* __restrict__ frame, unsigned long long stripeVec, uint32_t nStripes, uint32_t nRanks) {
    const unsigned long long total = stripeVec * nStripes * nRanks;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long t = i % stripeVec, sj = i / stripeVec;          // sj = j * nRanks + r in the frame
        const unsigned long long j = sj / nRanks, r = sj % nRanks;
        frame[i] = recv[(r * nStripes + j) * stripeVec + t];
    }
}
extern "C" hipError_t ft_launch_deinterleave(const float* recv, float* frame, unsigned long long stripeFloats, uint32_t nStripes, uint32_t nRanks, hipStream_t st) {
    if (stripeFloats % 4ull == 0ull && ((uintptr_t)recv & 15u) == 0 && ((uintptr_t)frame & 15u) == 0)
        hipLaunchKernelGGL(ft_deinterleave_kernel<float4>, dim3(4096), dim3(256), 0, st, reinterpret_cast<const float4*>(recv), reinterpret_cast<float4*>(frame),
                           stripeFloats / 4ull, nStripes, nRanks);
    else
        hipLaunchKernelGGL(ft_deinterleave_kernel<float>, dim3(4096), dim3(256), 0, st, recv, frame, stripeFloats, nStripes, nRanks);
    return hipGetLastError();
}
// exhaustive proof of the fast forms: every float bit pattern in [lo, hi] (same sign), fast vs exact
extern "C" __global__ void ft_selftest_kernel(int op, uint32_t lo, uint32_t hi, unsigned long long* mismatches) {
    unsigned long long bad = 0;
    for (unsigned long long u = (unsigned long long)lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u <= hi;
         u += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)u);
        if (op == 5) {                                                 // strength * (s - w) through the subtraction's output modifier, x = s, four radii
            const float ws[4] = {0x1p-20f, 0.1f, 1.0f, 1.0e4f};
            for (int k = 0; k < 4; ++k) {
                f3 y = mk3(x, ws[k], 0.0f);
                const uint32_t mode = ft_omod_on(y);
                float t1 = ft_strength_times_diff<1>(0.0f, y.x, y.y), t2 = ft_strength_times_diff<2>(0.0f, y.x, y.y), t3 = ft_strength_times_diff<3>(0.0f, y.x, y.y);
                asm volatile("" : "+v"(t1), "+v"(t2), "+v"(t3));         // all three are computed before the mode is restored
                float tie = t1;
                ft_omod_off(mode, tie);
                const float d = x - ws[k];
                const float e1 = -2.0f * d, e2 = -4.0f * d, e3 = -0.5f * d;
                bad += !(t1 == e1 && t2 == e2 && t3 == e3) || (d != 0.0f && (__float_as_uint(t1) != __float_as_uint(e1) || __float_as_uint(t2) != __float_as_uint(e2) || __float_as_uint(t3) != __float_as_uint(e3)));
                if (tie != tie) bad += 1;
            }
            continue;
        }
        if (op >= 3) {                                                 // the NEAR sphere loop's forms under its mode: 3 = four-instruction root, 4 = exponent-add exp
            f3 y = mk3(x, x, x);
            const uint32_t mode = ft_omod_on(y);
            float a = op == 3 ? ft_sqrt_fast_omod(y.x) : ft_exp_fast<true>(y.x);
            ft_omod_off(mode, a);
            bad += __float_as_uint(a) != __float_as_uint(op == 3 ? sqrtf(x) : ft_exp(x));
            continue;
        }
        const float a = op == 0 ? ft_sqrt_fast(x) : (op == 1 ? ft_exp_fast<false>(x) : ft_exp_fast<true>(x));
        const float b = op == 0 ? sqrtf(x) : ft_exp(x);
        bad += __float_as_uint(a) != __float_as_uint(b);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (kept in this translation unit so the C ABI file is plain C++)
// ------------------------------------------------------------------------------------------------
#ifdef FT_EXPERIMENT
// Diagnostic build only (`make experiment`, tools/asm_variants*.py): the lean kernel can be replaced at run time by the same
// kernel from a re-assembled code object, so that edits of its machine code (instruction placement) can be timed.
#include <hip/hip_runtime_api.h>
static hipModule_t ft_exp_module = nullptr;
static hipFunction_t ft_exp_fn = nullptr;
extern "C" int ft_debug_set_hsaco(const char* path) {
    if (ft_exp_module) { (void)hipModuleUnload(ft_exp_module); ft_exp_module = nullptr; ft_exp_fn = nullptr; }
    if (!path || !*path) return 0;
    if (hipModuleLoad(&ft_exp_module, path) != hipSuccess) return -1;
    if (hipModuleGetFunction(&ft_exp_fn, ft_exp_module, "ft_trace_kernel_smooth_spheres") != hipSuccess) return -2;
    return 0;
}
#endif
extern "C" hipError_t ft_launch_trace(const FtRenderArgs* a, unsigned blocks, size_t ldsBytes, hipStream_t st) {
    const bool ext = a->ext != 0u;
    const unsigned v = a->S.fastPath;
    if (a->math != 0u) {                // FT_OPT_MATH = glibc, scene with a unionSmooth
        if (v == 1 && ext) hipLaunchKernelGGL(ft_trace_kernel_smooth_spheres_ext_libm, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
        else if (v == 1) hipLaunchKernelGGL(ft_trace_kernel_smooth_spheres_libm, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
        else if (v == 2 && ext) hipLaunchKernelGGL(ft_trace_kernel_calls_ext_libm, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
        else if (v == 2) hipLaunchKernelGGL(ft_trace_kernel_calls_libm, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
        else if (ext) hipLaunchKernelGGL(ft_trace_kernel_ext_libm, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
        else hipLaunchKernelGGL(ft_trace_kernel_libm, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
        return hipGetLastError();
    }
#ifdef FT_EXPERIMENT
    if (v == 1 && !ext && ft_exp_fn) {
        FtRenderArgs args = *a;
        size_t size = sizeof(args);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        return hipModuleLaunchKernel(ft_exp_fn, blocks, 1, 1, FT_BLOCK, 1, 1, (unsigned)ldsBytes, st, nullptr, extra);
    }
#endif                                  // 0 general, 1 lean smooth-sphere, 2 general with call children, 3 carved union
    if (v == 3 && !ext) {
        FtRenderArgs args = *a;
        void* kp[] = {&args};
        return hipLaunchKernel(ft_carved_kernel(a->carve.kind), dim3(blocks), dim3(FT_BLOCK), kp, ldsBytes, st);
    }
    if (v == 1 && ext) hipLaunchKernelGGL(ft_trace_kernel_smooth_spheres_ext, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
    else if (v == 1) hipLaunchKernelGGL(ft_trace_kernel_smooth_spheres, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
    else if (v == 2 && ext) hipLaunchKernelGGL(ft_trace_kernel_calls_ext, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
    else if (v == 2) hipLaunchKernelGGL(ft_trace_kernel_calls, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
    else if (ext) hipLaunchKernelGGL(ft_trace_kernel_ext, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
    else hipLaunchKernelGGL(ft_trace_kernel, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *a);
    return hipGetLastError();
}
extern "C" hipError_t ft_launch_eval_points(const FtSceneDev* S, int math, const float* pts, long long n, float* outD, int* outM,
                                            unsigned blocks, size_t ldsBytes, hipStream_t st) {
    if (math) hipLaunchKernelGGL(ft_eval_points_kernel_libm, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *S, pts, n, outD, outM);
    else hipLaunchKernelGGL(ft_eval_points_kernel, dim3(blocks), dim3(FT_BLOCK), ldsBytes, st, *S, pts, n, outD, outM);
    return hipGetLastError();
}
// checksums of the glibc restatements over whole ranges of float bit patterns (ft_selftest_libm): chunk c covers the 2^24 patterns from
// lo + c * 2^24; sum of splitmix64((input bits << 32) | result bits), NaN results canonicalised — the host forms the same sums with the real libm
__device__ __forceinline__ ft_u64 ft_splitmix64(ft_u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
extern "C" __global__ void __launch_bounds__(256) ft_libm_checksum_kernel(int op, int variant, float y, uint32_t lo, unsigned long long* __restrict__ sums) {
    __shared__ ft_u64 tab[FT_LIBM_TAB_DOUBLES];
    __shared__ unsigned long long part[4];
    if (threadIdx.x < FT_LIBM_TAB_DOUBLES) tab[threadIdx.x] = ft_libm_tab_g[threadIdx.x];
    __syncthreads();
    // 64 blocks per chunk of 2^24 inputs: 2^18 per block, 2^10 per thread
    const uint32_t chunk = blockIdx.x >> 6, sub = blockIdx.x & 63u;
    const uint32_t base = lo + (chunk << 24) + (sub << 18);
    ft_u64 acc = 0;
    for (uint32_t k = threadIdx.x; k < (1u << 18); k += 256u) {
        const uint32_t u = base + k;
        const float x = __uint_as_float(u);
        float r;
        if (op == 0) r = variant == 1 ? ft_glibc_expf<true>(x, tab) : ft_glibc_expf<false>(x, tab);
        else if (op == 1) r = variant == 1 ? ft_glibc_logf<true>(x, tab) : ft_glibc_logf<false>(x, tab);
        else r = variant == 1 ? ft_glibc_powf<true>(x, y, tab) : ft_glibc_powf<false>(x, y, tab);
        const uint32_t v = r != r ? 0x7fc00000u : __float_as_uint(r);
        acc += ft_splitmix64(((ft_u64)u << 32) | v);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[chunk], part[0] + part[1] + part[2] + part[3]);
}
extern "C" hipError_t ft_launch_libm_checksum(int op, int variant, float y, uint32_t lo, uint32_t nChunks, unsigned long long* d_sums, hipStream_t st) {
    hipLaunchKernelGGL(ft_libm_checksum_kernel, dim3(nChunks * 64u), dim3(256), 0, st, op, variant, y, lo, d_sums);
    return hipGetLastError();
}
extern "C" hipError_t ft_launch_math(int op, const float* x, const float* y, long long n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(ft_math_kernel, dim3(1024), dim3(256), 0, st, op, x, y, n, out);
    return hipGetLastError();
}
extern "C" hipError_t ft_launch_grid_build(const FtGridBuildArgs* a, hipStream_t st) {
    hipLaunchKernelGGL(ft_grid_build_kernel, dim3(a->c * a->c * a->c), dim3(FT_BLOCK), 0, st, *a);
    return hipGetLastError();
}
extern "C" hipError_t ft_launch_grid_compact(const FtItem* tmp, const uint32_t* cellStart, uint32_t ncells, uint32_t n, FtItem* items, hipStream_t st) {
    hipLaunchKernelGGL(ft_grid_compact_kernel, dim3(ncells), dim3(128), 0, st, tmp, cellStart, n, items);
    return hipGetLastError();
}
extern "C" hipError_t ft_launch_resolve(const float* planes, float* out, unsigned long long nFloats, unsigned spp, hipStream_t st) {
    hipLaunchKernelGGL(ft_resolve_kernel, dim3(2048), dim3(256), 0, st, planes, out, nFloats, spp);
    return hipGetLastError();
}
extern "C" hipError_t ft_launch_tonemap(const float* frame, uint32_t X, uint32_t Y, uint32_t* maxBits, float gammaInv, uint32_t dither, uint32_t seed,
                                        int bmpOrder, unsigned char* out, unsigned numCUs, int math, hipStream_t st) {
    const unsigned long long nFloats = 3ull * X * Y;
    hipError_t e = hipMemsetAsync(maxBits, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    const unsigned maxBlocks = numCUs * 8u;
    const unsigned long long want = (nFloats / 12ull + 255ull) / 256ull;      // one thread per 4 pixels
    hipLaunchKernelGGL(ft_tonemap_max_kernel, dim3((unsigned)(want < 1 ? 1 : (want < maxBlocks ? want : maxBlocks))), dim3(256), 0, st, frame, nFloats, maxBits);
    if (bmpOrder) {
        const unsigned tiles = ((X + FT_TM_TILE - 1) / FT_TM_TILE) * ((Y + FT_TM_TILE - 1) / FT_TM_TILE);
        hipLaunchKernelGGL(ft_tonemap_bmp_kernel, dim3(tiles), dim3(256), 0, st, frame, X, Y, (const uint32_t*)maxBits, gammaInv, dither, seed, out, math);
    } else {
        const unsigned long long wantP = ((unsigned long long)X * Y + 255ull) / 256ull;
        hipLaunchKernelGGL(ft_tonemap_map_kernel, dim3((unsigned)(wantP < maxBlocks * 4ull ? wantP : maxBlocks * 4ull)), dim3(256), 0, st, frame, X, Y,
                           (const uint32_t*)maxBits, gammaInv, dither, seed, out, math);
    }
    return hipGetLastError();
}
extern "C" hipError_t ft_launch_selftest(int op, uint32_t lo, uint32_t hi, unsigned long long* d_mismatches, hipStream_t st) {
    hipLaunchKernelGGL(ft_selftest_kernel, dim3(4096), dim3(256), 0, st, op, lo, hi, d_mismatches);
    return hipGetLastError();
}
#ifdef FT_UNION_PROFILE
extern "C" hipError_t ft_debug_union_counters(unsigned long long out[12]) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(ft_union_dbg), sizeof(unsigned long long) * 12);
    if (e != hipSuccess) return e;
    unsigned long long zero[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(ft_union_dbg), zero, sizeof(zero));
}
#endif
extern "C" hipError_t ft_trace_occupancy(unsigned fastPath, unsigned carveKind, bool ext, bool libm, size_t ldsBytes, int* blocksPerCU) {
    if (fastPath == 3 && !ext) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, ft_carved_kernel(carveKind), FT_BLOCK, ldsBytes);
    if (libm) {
        const void* kl = fastPath == 1 ? (ext ? (const void*)ft_trace_kernel_smooth_spheres_ext_libm : (const void*)ft_trace_kernel_smooth_spheres_libm)
                       : fastPath == 2 ? (ext ? (const void*)ft_trace_kernel_calls_ext_libm : (const void*)ft_trace_kernel_calls_libm)
                                       : (ext ? (const void*)ft_trace_kernel_ext_libm : (const void*)ft_trace_kernel_libm);
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, kl, FT_BLOCK, ldsBytes);
    }
    const void* k = fastPath == 1 ? (ext ? (const void*)ft_trace_kernel_smooth_spheres_ext : (const void*)ft_trace_kernel_smooth_spheres)
                  : fastPath == 2 ? (ext ? (const void*)ft_trace_kernel_calls_ext : (const void*)ft_trace_kernel_calls)
                                  : (ext ? (const void*)ft_trace_kernel_ext : (const void*)ft_trace_kernel);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocksPerCU, k, FT_BLOCK, ldsBytes);
}
