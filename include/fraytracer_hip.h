/* fraytracer_hip.h — C ABI of libfraytracer_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for FrayTracer's per-pixel hot path.  The reference has no FFI: the
 * seam is `SdfScene.trace : SdfScene -> Ray -> FColor` (src/FrayTracer/SdfScene.fs:7-8)
 * driven by `Image.render` (src/FrayTracer/Image.fs:26-35), called once from
 * src/FrayTracer.Console/Program.fs:90-93.  Because an F# scene is a record of opaque
 * closures (src/FrayTracer/Types.fs:40-55), every scene constructor of the reference gets a
 * native twin here; the F# layer calls the twin next to building its closure and keeps the
 * returned handle (INTEGRATION.md shows the [<DllImport>] stubs).  The library owns
 * flattening (boundaries, uniform grids) and all device work.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Functions return 0 / a handle
 * >= 0 on success and a negative ft_status on failure; ft_last_error() gives the message
 * for the calling thread.  All structs are float/int32 only, 4-byte aligned, and match the
 * sequential layout of the F# [<Struct>] records they mirror, so they are blittable.
 * Inputs are copied; the caller keeps ownership.  A context is not re-entrant; distinct
 * contexts may be used from distinct threads.  There is NO CPU fallback: every render /
 * trace entry point fails with FT_ERR_NO_DEVICE when the context has no GPU.
 */
#ifndef FRAYTRACER_HIP_H
#define FRAYTRACER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FT_ABI_VERSION 5

typedef enum ft_status {
    FT_OK = 0,
    FT_ERR_INVALID = -1,      /* bad handle / argument */
    FT_ERR_NO_DEVICE = -2,    /* context was created without a GPU, or HIP reports none */
    FT_ERR_HIP = -3,          /* a HIP runtime call failed (message has the HIP error string) */
    FT_ERR_UNSUPPORTED = -4,  /* scene exceeds a documented limit (DESIGN.md "Limits") */
    FT_ERR_EMPTY = -5,        /* empty child list: the reference throws "No SdfObjects given." */
    FT_ERR_COMM = -6          /* RCCL failure in ft_render_multi */
} ft_status;

typedef int32_t ft_handle;                 /* index into a context-owned table; < 0 is an ft_status */
typedef struct ft_ctx ft_ctx;
typedef struct ft_scene ft_scene;

/* ---- blittable mirrors of the reference records ------------------------------------ */
typedef struct ft_vec3 { float x, y, z; } ft_vec3;                       /* System.Numerics.Vector3 */
typedef struct ft_ray {                                                  /* Types.fs:9-17 (32 B) */
    ft_vec3 origin; ft_vec3 direction; float length; float epsilon;
} ft_ray;
typedef struct ft_boundary { ft_vec3 center; float radius; } ft_boundary;/* Types.fs:19-24 (16 B) */
/* SdfFormTraceResult voption (Types.fs:32-37): `hit` = 0 is ValueNone (the other fields are then 0) */
typedef struct ft_form_trace_result { ft_ray ray; float distance; int32_t hit; } ft_form_trace_result;              /* 40 B */
/* SdfObjectTraceResult voption (Types.fs:57-65): Ray (origin pulled back by epsilon, SdfObject.fs:73), Normal, Color */
typedef struct ft_object_trace_result { ft_ray ray; ft_vec3 normal; ft_vec3 color; int32_t hit; int32_t reserved; } ft_object_trace_result;   /* 64 B */
typedef struct ft_sphere { ft_vec3 center; float radius; } ft_sphere;    /* SdfForm.fs:118-123 */
typedef struct ft_capsule { ft_vec3 from; ft_vec3 to; float radius; } ft_capsule;        /* SdfForm.fs:137-143 */
typedef struct ft_torus { ft_vec3 center; ft_vec3 normal; float major_radius; float minor_radius; } ft_torus; /* SdfForm.fs:172-179 */
typedef struct ft_triangle { ft_vec3 v1; ft_vec3 v2; ft_vec3 v3; float radius; } ft_triangle;  /* SdfForm.fs:205-212 */
typedef struct ft_box { ft_vec3 center; ft_vec3 half_extent; } ft_box;   /* EXTENSION: not in the reference */
typedef struct ft_camera {                                               /* Camera.fs:16-22 (48 B) */
    ft_vec3 position; ft_vec3 forward; ft_vec3 up_scaled; ft_vec3 right_scaled;
} ft_camera;

/* Image.render arguments (Image.fs:26) plus the column tiling used for multi-GPU.
 * Local column c in [0, n_columns) maps to image column
 *     x = x0 + (c / stripe_width) * stripe_width * stripe_ranks + stripe_rank * stripe_width + c % stripe_width
 * (stripe_ranks = 1, stripe_rank = 0 gives the contiguous range [x0, x0 + n_columns)).
 * Output is n_columns x height x 3 float32, column-major / y contiguous: exactly the
 * FColor[X,Y] layout Array2D.Parallel.init produces (Array2D.fs:30-38). */
typedef struct ft_render_params {
    int32_t width, height;        /* ImageSize.X, ImageSize.Y (Image.fs:8-13) */
    int32_t x0, n_columns;
    int32_t stripe_width, stripe_ranks, stripe_rank;
    int32_t spp;                  /* 1 = the reference (one corner sample). >1 is an EXTENSION. */
    float epsilon, length;        /* Image.render's first two arguments */
    int32_t ao_samples;           /* 0 = the reference. >0: EXTENSION ambient-occlusion rays */
    float ao_radius;
    int32_t max_bounces;          /* 0 = the reference (glass shades as createSolid tint). >0: EXTENSION, glass
                                   * materials refract / reflect; a path ends black after this many interactions */
    int32_t spectral;             /* 0 = off. 1..16: EXTENSION, sample k is traced at wavelength bin k % spectral
                                   * (must divide spp) and weighted with that bin's RGB response */
} ft_render_params;

typedef struct ft_stats {         /* filled per call; all counts are exact */
    uint64_t rays_primary, rays_shadow, rays_ext;
    uint64_t hits_primary, hits_shadow;
    uint64_t sdf_evals;           /* scene-SDF evaluations (march steps + normal probes) */
    uint64_t flags;               /* bit0 NaN distance met, bit2 step cap hit (reference would not terminate) */
    float kernel_ms;              /* HIP-event time of the render kernel(s) of this call */
    float culled_fraction;        /* lean kernel: share of the (child, ray) pairs of the smooth union that were dropped as exact no-ops (FT_OPT_CULL) */
    uint64_t wave_evals;          /* wave-level evaluation rounds: sdf_evals / (64 * wave_evals) = lane utilisation */
    float shader_mhz;             /* shader clock the render kernel(s) of this call ran at: s_memtime ticks / s_memrealtime ticks
                                   * (constant 100 MHz) of one wave that lives as long as the kernel; 0 if unknown */
    float tail_fraction;          /* share of sdf_evals done in latency mode (one ray per wave, FT_OPT_TAIL_K) */
} ft_stats;

/* ---- context ------------------------------------------------------------------------ */
/* device >= 0: HIP device ordinal.  device = -1: host-only context (scene construction and
 * ft_scene_export work; anything that needs the GPU fails with FT_ERR_NO_DEVICE). */
int ft_abi_version(void);
int ft_ctx_create(int device, ft_ctx** out);
void ft_ctx_destroy(ft_ctx* ctx);
const char* ft_last_error(void);
/* use an existing HIP stream (e.g. torch's current stream) for all launches; NULL = own stream */
int ft_ctx_set_stream(ft_ctx* ctx, void* hip_stream);
/* "src=<hash>;kind=<product|profile|experiment>": hash of the sources this library was built from
 * (fraytracer_amd/csrc/source_hash.py computes the same value from a source tree) and the kind of build. */
const char* ft_build_info(void);
/* Per-context switches (the library reads no environment variables).  FT_OPT_MATH selects the arithmetic of MathF.Exp / Log / Pow
 * (below); every other option is for experiments and A/B measurements only and never changes a rendered bit. */
typedef enum ft_option {
    FT_OPT_REFILL_MIN = 1,        /* 1..64 (default 64): idle lanes a wave waits for before it takes new rays */
    FT_OPT_MAX_BLOCKS_PER_CU = 2, /* 0 (default) = the occupancy limit; 1..8 caps the resident workgroups per CU */
    FT_OPT_HOST_CHUNKS = 3,       /* 0 (default) = automatic; 1..16 column chunks of ft_render's host-output pipeline */
    FT_OPT_HOST_PIN = 4,          /* 1 (default): ft_render page-locks an unregistered destination for the call; 0: leaves it pageable */
    FT_OPT_TAIL_K = 5,            /* latency mode: a wave holding at most this many rays evaluates them one at a time with all 64 lanes;
                                   * -1 (default) = the kernel's own threshold, 0 = off, 1..64 */
    FT_OPT_MATH = 6,              /* ft_math_mode (below); default FT_MATH_FIXED */
    FT_OPT_CHUNK = 8,             /* 64 (default): rays a wave takes per grab = one 8x8 tile; 32 / 16: half / quarter tiles (experiments) */
    FT_OPT_CULL = 9,              /* 1 (default): the smooth-union kernel drops, per wave and round, the children whose terms are below half an ulp
                                   * of the running sum in every ray of the wave (exact: the sum is unchanged bit for bit); 0: every child, every round */
    FT_OPT_ESCAPE = 10,           /* 1 (default): a ray that can no longer come within epsilon of the scene's support sphere — outside it and heading away, or
                                   * passing it by — ends as the miss its march is bound to end in, without further evaluations (same frame, fewer sdf_evals);
                                   * 0: every ray marches until its Length is used up, as the reference does */
    FT_OPT_LAZY_UNION = 11,       /* 1 (default): SdfForm.union as the first child of an intersect (the reference's own scene: intersect(union of tori, sphere))
                                   * stops its candidate walk at Items.[0] wherever that distance is already <= the next child's, which then decides the
                                   * intersect's value (exact; only where no hit is possible); 0: every union walk runs to its end */
    FT_OPT_CARVED = 12,           /* 1 (default): a scene that is one SdfForm.union of primitives followed by at most two single-primitive intersect / subtract
                                   * steps — the reference's own subtract(intersect(union tori, sphere), sphere) — is traced by a kernel specialised for that
                                   * shape (registers instead of LDS value slots, no interpreter, the tail decides early exits of the walk; same bits);
                                   * 0: the general interpreter kernel */
    FT_OPT_REUSE = 13,            /* 1 (default): every shadow ray (and EXTENSION ambient-occlusion ray) starts at the hit position, where the fourth probe of SdfForm.normal
                                   * has just evaluated the scene (SdfForm.fs:112, SdfObject.fs:73: the same point, bit for bit); the first evaluation of its march is
                                   * that value and is not computed again; likewise every primary ray of ft_render starts at the camera position, which each wave evaluates
                                   * once (same frame, same ray and hit counters, fewer sdf_evals); 0: evaluated once per ray, as the reference does */
    FT_OPT_GUIDED = 7             /* 1: the last jobs of a launch are handed out in half and quarter tiles (lean kernel); 0 (default): whole tiles only */
} ft_option;
/* MathF.Exp / MathF.Log (SdfForm.unionSmooth, SdfForm.fs:80,82) and MathF.Pow (FColor.gammaInverse, FColor.fs:50-55) are the C runtime's
 * expf / logf / powf under .NET: platform arithmetic, not one fixed function.
 *   FT_MATH_FIXED       one fixed algorithm per function (IEEE + - * / fma only): the same bits on every machine; <= 0.93 ulp (exp),
 *                       < 0.51 ulp (log), correctly rounded pow except at ~1e-7 of the operands.
 *   FT_MATH_GLIBC_FMA   glibc 2.35's expf / logf / powf as its x86-64 FMA build computes them (`__expf_fma` ...: the variant glibc's ifunc
 *                       selects on every CPU with FMA and AVX2) — bit for bit what the reference's CPU path returns on such a Linux host.
 *   FT_MATH_GLIBC_SSE2  the same functions as glibc's SSE2 build rounds them (x86-64 CPUs without FMA / AVX2).
 * Both glibc modes are restatements running on the GPU (csrc/ft_libm.h), proved against the running libm over all 2^32 inputs
 * (ft_selftest_libm).  Scenes without a unionSmooth trace identically in every mode. */
typedef enum ft_math_mode { FT_MATH_FIXED = 0, FT_MATH_GLIBC_FMA = 1, FT_MATH_GLIBC_SSE2 = 2 } ft_math_mode;
int ft_ctx_set_option(ft_ctx* ctx, int32_t option, int32_t value);
int ft_ctx_get_option(const ft_ctx* ctx, int32_t option, int32_t* value);

/* ---- scene construction: one entry per reference constructor ---------------------------- */
ft_handle ft_form_sphere(ft_ctx*, const ft_sphere*);                       /* SdfForm.Primitive.sphere  SdfForm.fs:125-135 */
ft_handle ft_form_capsule(ft_ctx*, const ft_capsule*);                     /* SdfForm.Primitive.capsule SdfForm.fs:145-170 */
ft_handle ft_form_torus(ft_ctx*, const ft_torus*);                         /* SdfForm.Primitive.torus   SdfForm.fs:181-203 */
ft_handle ft_form_triangle(ft_ctx*, const ft_triangle*);                   /* SdfForm.Primitive.triangle SdfForm.fs:214-268 */
ft_handle ft_form_box(ft_ctx*, const ft_box*);                             /* EXTENSION */
ft_handle ft_form_union(ft_ctx*, const ft_handle* forms, int32_t n);       /* SdfForm.union        SdfForm.fs:14-40 */
ft_handle ft_form_subtract(ft_ctx*, ft_handle a, ft_handle b);             /* SdfForm.subtract     SdfForm.fs:42-49 */
ft_handle ft_form_intersect(ft_ctx*, const ft_handle* forms, int32_t n);   /* SdfForm.intersect    SdfForm.fs:51-67 */
ft_handle ft_form_union_smooth(ft_ctx*, float strength, const ft_handle* forms, int32_t n); /* SdfForm.unionSmooth SdfForm.fs:69-91 */
int ft_form_boundary(ft_ctx*, ft_handle form, ft_boundary* out);           /* SdfForm.Boundary */

ft_handle ft_material_solid(ft_ctx*, const float rgb[3]);                  /* SdfMaterial.createSolid SdfMaterial.fs:4-7 */
/* EXTENSION (BASELINE.json config 5; the reference's SdfMaterial cannot spawn rays, Types.fs:46-49): glass with
 * index of refraction `ior` at 550 nm and Cauchy dispersion n(lambda) = ior + dispersion * (1/lambda_um^2 - 1/0.55^2).
 * Fresnel terms follow the reference's dead Light.fs:30-59 (repaired). */
ft_handle ft_material_glass(ft_ctx*, const float tint[3], float ior, float dispersion);
/* EXTENSION: the wavelength table behind ft_render_params.spectral = nw: out[nw][4] = RGB weight, Cauchy term */
int ft_spectral_table(int32_t nw, float* out);
ft_handle ft_object_create(ft_ctx*, ft_handle material, ft_handle form);   /* SdfObject.create    SdfObject.fs:6-10 */
ft_handle ft_object_union(ft_ctx*, const ft_handle* objects, int32_t n);   /* SdfObject.union     SdfObject.fs:12-48 */
ft_handle ft_object_subtract(ft_ctx*, ft_handle object, ft_handle form);   /* SdfObject.subtract  SdfObject.fs:50-54 */
ft_handle ft_object_intersect(ft_ctx*, ft_handle object, const ft_handle* forms, int32_t n); /* SdfObject.intersect SdfObject.fs:56-64 */
ft_handle ft_object_form(ft_ctx*, ft_handle object);                       /* object.Form */

ft_handle ft_light_directional(ft_ctx*, const float direction[3], const float rgb[3]);  /* SdfLight.directional SdfLight.fs:6-21 */
ft_handle ft_light_point(ft_ctx*, const float position[3], const float rgb[3]);         /* SdfLight.point       SdfLight.fs:23-42 */

/* SdfScene record (Types.fs:74-79): flattens the immutable tree and uploads it. */
int ft_scene_create(ft_ctx*, ft_handle object, const float background_rgb[3],
                    const ft_handle* lights, int32_t n_lights, ft_scene** out);
void ft_scene_destroy(ft_scene*);

/* Lens.create (Camera.fs:11-14) and Camera.lookAt (Camera.fs:33-42); host-side, once per frame. */
float ft_lens_create(float field_of_view);
int ft_camera_look_at(const float position[3], const float look_at[3], const float up[3],
                      float near_plane_size, ft_camera* out);

/* ---- the hot path --------------------------------------------------------------------- */
/* Image.render epsilon length imageSize camera (SdfScene.trace scene)  — Image.fs:26-35 +
 * SdfScene.fs:7-28.  Synchronous.  `out` is host memory (pinned or pageable; a pageable buffer is page-locked for the
 * duration of the call while the GPU renders, and the frame is copied chunk by chunk behind the rendering). */
int ft_render(ft_ctx*, const ft_scene*, const ft_camera*, const ft_render_params*,
              float* out, ft_stats* stats);
/* Page-lock a host buffer the caller reuses as ft_render's `out` (the F# side pins its FColor[,] with GCHandle, as
 * Image.fs:77-86 does for the bitmap): the frame is then written by DMA at link rate without the per-call pinning
 * ft_render otherwise does itself.  Unregister before freeing the buffer. */
int ft_host_register(ft_ctx*, void* p, uint64_t bytes);
int ft_host_unregister(ft_ctx*, void* p);
/* Same, output left in device memory `d_out`, launched on the context's stream and NOT
 * synchronised: for callers that keep the frame in HBM (multi-GPU gather, bench). */
int ft_render_device(ft_ctx*, const ft_scene*, const ft_camera*, const ft_render_params*,
                     void* d_out);
/* counters / kernel time of the launches since the last call; synchronises the stream */
int ft_collect_stats(ft_ctx*, ft_stats* stats);

/* ---- around the hot path: tone map + 8-bit output -------------------------------- */
/* Image.toColors gamma rng image (Image.fs:37-50): max = Max(0.01, max over all channels); per channel
 * Pow(c / max, 1 / gamma) * 254.5 + u, rounded half-to-even, `min 255` (FColor.fs:43-55).  The reference draws u from one
 * System.Random shared by a parallel map (racy, not reproducible): here u is 0.5 (dither = 0) or a counter-based hash of
 * (x, y, channel, seed) in [0, 1) (dither = 1) — comparable to the reference to +-1 LSB.  MathF.Pow is platform libm;
 * the library uses one fixed double-precision algorithm (ft_math.h ft_pow) shared with the test oracle.
 * bmp_order = 0: out[(x * Y + y) * 3 + k] = R, G, B of image[x, y] — the Color[X,Y] value of Image.toColors.
 * bmp_order = 1: the scan0 buffer Image.toBitmap builds (Image.fs:61-86): row r from the top, column c holds
 *                image[X-1-c, r] as bytes B, G, R, stride X * 3 — ready for a 24-bpp bitmap. */
typedef struct ft_tonemap_params {
    float gamma;                  /* Image.toColors' gamma (Program.fs:98 passes 2.2f) */
    int32_t dither;               /* 0: u = 0.5 everywhere; 1: hashed noise */
    uint32_t seed;
    int32_t bmp_order;
} ft_tonemap_params;
/* frame in device memory (X x Y x 3 float32 as ft_render_device writes it) -> 8-bit image in device memory; asynchronous
 * on the context's stream */
int ft_tone_map_device(ft_ctx*, const void* d_frame, int32_t X, int32_t Y, const ft_tonemap_params*, void* d_out);
/* same, 8-bit image copied to host memory (3 bytes per pixel cross PCIe instead of 12); *max_out = the normalisation */
int ft_tone_map(ft_ctx*, const void* d_frame, int32_t X, int32_t Y, const ft_tonemap_params*, uint8_t* out, float* max_out);
/* Image.toColors on a host FColor[X,Y] (drop-in for the reference call on an image that is already on the host) */
int ft_tone_map_host(ft_ctx*, const float* frame, int32_t X, int32_t Y, const ft_tonemap_params*, uint8_t* out, float* max_out);
/* Program.fs:90-100 in one call: Image.render + Image.toColors (+ toBitmap order) on the device; only the 8-bit image
 * leaves the GPU.  The render parameters must describe the whole frame. */
int ft_render_colors(ft_ctx*, const ft_scene*, const ft_camera*, const ft_render_params*, const ft_tonemap_params*,
                     uint8_t* out, float* max_out, ft_stats* stats);

/* SdfScene.trace over an explicit ray buffer (the "ray buffer" form): out_rgb is n x 3 floats. */
int ft_trace_rays(ft_ctx*, const ft_scene*, const ft_ray* rays, int64_t n, float* out_rgb, ft_stats* stats);

/* SdfForm.tryTrace scene.Object.Form ray (SdfForm.fs:93-104) over a ray buffer: the ray as it stands at the hit
 * (Origin moved, Length reduced) and the last Distance. */
int ft_form_try_trace(ft_ctx*, const ft_scene*, const ft_ray* rays, int64_t n, ft_form_trace_result* out, ft_stats* stats);
/* SdfObject.tryTrace scene.Object ray (SdfObject.fs:66-78) over a ray buffer: march, normalFromRay, Ray.move -eps,
 * material colour picked at the un-pulled hit origin. */
int ft_object_try_trace(ft_ctx*, const ft_scene*, const ft_ray* rays, int64_t n, ft_object_trace_result* out, ft_stats* stats);

/* scene.Object.Form.Distance at n points (+ index of the material the hit would pick, or
 * NULL).  Test/diagnostic entry: lets parity tests compare single SDF evaluations. */
int ft_eval_distance(ft_ctx*, const ft_scene*, const ft_vec3* points, int64_t n, float* out_distance, int32_t* out_material);

/* Single-process multi-GPU form (what an F# host would call): the flattened scene is
 * re-uploaded to each further device with ft_scene_clone, every device renders its column
 * stripes (stripe_rank = index in ctxs[], stripe_ranks = n) on its own host thread, and the
 * slabs are collected with ONE ncclGather to ctxs[0]'s device, de-interleaved
 * there and copied to `out` (full width x height x 3 host image).  full_frame's stripe_* and
 * x0/n_columns fields are ignored except stripe_width (0 = contiguous blocks of width/n). */
int ft_scene_clone(const ft_scene* src, ft_ctx* dst_ctx, ft_scene** out);
int ft_render_multi(ft_ctx* const* ctxs, const ft_scene* const* scenes, int32_t n,
                    const ft_camera*, const ft_render_params* full_frame, float* out, ft_stats* stats);

/* ---- introspection of the flattened scene (tests; not needed by a caller) ---------------- */
typedef struct ft_scene_info {
    int32_t n_instr, n_slots, n_consts, n_grids, n_children, n_cells, n_items, n_lights, n_materials;
    int32_t fast_path;            /* which specialised evaluator the scene selected: 0 general interpreter, 1 smooth union of spheres, 2 general with
                                   * on-demand sub-programs, 3 carved union (one union of primitives + at most two intersect / subtract steps) */
    int32_t cull_pc;              /* instruction of the program whose sphere run the child-culling pass serves; -1: none */
} ft_scene_info;
int ft_scene_info_get(const ft_scene*, ft_scene_info* out);
/* grid g: info = aabbMin[3], cellSizeInv[3]; counts[3]; arrays sized from ft_scene_info /
 * ft_scene_grid_shape: cell_start[ncells+1], centers[3*ncells], lower[nitems], child[nitems] */
int ft_scene_grid_shape(const ft_scene*, int32_t g, float info[6], int32_t counts[3], int32_t* n_cells, int32_t* n_items);
int ft_scene_grid_dump(const ft_scene*, int32_t g, uint32_t* cell_start, float* centers, float* lower, int32_t* child);
/* the scene's support sphere (centre xyz, radius): no point farther than epsilon from it can be a hit, which is what FT_OPT_ESCAPE relies on;
 * radius < 0: none is known for this scene (degenerate shapes, a unionSmooth of strength <= 0) and every ray marches to its end */
int ft_scene_support_sphere(const ft_scene*, float centre_radius[4]);

/* math primitives of the device path, evaluated on the GPU: op 0 exp, 1 log, 2 sqrt, 3 a/b, 4 fast sqrt, 5 fast exp
 * (y = second operand, may be NULL otherwise).  Used by tests/test_math_parity.py. */
int ft_math_eval(ft_ctx*, int32_t op, const float* x, const float* y, int64_t n, float* out);
/* further ops of ft_math_eval: 6 / 7 glibc expf (FMA / SSE2 build), 8 / 9 glibc logf, 10 / 11 glibc powf(x, y), 12 the fixed pow(x, y),
 * 13 / 14 the branch-free MathF.Max(x, y) / MathF.Min(x, y) of the carved-union kernels (14: x never NaN).
 * ft_selftest_libm: checksums of the device restatement of glibc's expf (op 0), logf (1) or powf(x, y) (2) in `variant`
 * (FT_MATH_GLIBC_FMA / FT_MATH_GLIBC_SSE2) over n_chunks x 2^24 consecutive float bit patterns from lo_bits:
 * sums[c] = sum over the chunk of splitmix64((input bits << 32) | result bits) mod 2^64, every NaN result taken as 0x7fc00000.  A test forms
 * the same sums with the libm of the machine it runs on (256 chunks = every float). */
int ft_selftest_libm(ft_ctx*, int32_t op, int32_t variant, float y, uint32_t lo_bits, int32_t n_chunks, uint64_t* sums);
/* Exhaustive check, on the GPU, of the fast sqrt / exp forms used inside the smooth-union loop against
 * the exact forms, over EVERY float of the ranges they are used on; mismatches[0] = sqrt (both the five-instruction
 * form and the four-instruction form that runs with output modifiers enabled), [1] = exp (ldexp form, [-2.9e6, 88]),
 * [2] = exp (exponent-add form, [-87, 88], in the normal mode and under the near loop's mode); all must be 0. */
int ft_selftest_fastmath(ft_ctx*, uint64_t mismatches[3]);

#ifdef __cplusplus
}
#endif
#endif /* FRAYTRACER_HIP_H */
