// ft_kernels.h — kernel argument block and host-callable launchers (implemented in kernels.hip / launch.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/fraytracer_hip.h"
#include "ft_device.h"

#define FT_BLOCK 256          // 4 waves; every wave is an independent persistent worker
// dynamic LDS of a trace workgroup starts with a header: 64 words (per-wave statistics, the reporting wave's start clocks), in the diagnostic
// build 13 rows of per-lane union-walk counters (kernels.hip FT_UDBG: 12 + 1 scratch row), then FT_SH_ROWS rows of per-lane shading state
#define FT_LDS_CNT_WORDS 64
#ifdef FT_UNION_PROFILE
#define FT_LDS_DBG_ROWS 13
#else
#define FT_LDS_DBG_ROWS 0
#endif
#define FT_SH_ROWS 10         // hit position, normal, accumulated light (3 each), the distance at the hit position
#define FT_LDS_SH_BASE (FT_LDS_CNT_WORDS + FT_LDS_DBG_ROWS * FT_BLOCK)
#define FT_LDS_HDR_FLOATS (FT_LDS_SH_BASE + FT_SH_ROWS * FT_BLOCK)
// lean kernel: every wave owns a row of FT_CULL_MAX float4 records behind everything else (kernels.hip "Exact child culling").  With 256 staged
// spheres a workgroup then needs 34 304 bytes of LDS: four per CU.  224 records would let a fifth in, and lose: the LAST children of a union
// are the ones dropped most often (the running sum is largest in front of them) — C3 4096^2 34.3 ms at 256, 36.5 at 224
#define FT_CULL_MAX 256
#define FT_CULL_ROW (4 * FT_CULL_MAX)

struct FtRenderArgs {
    FtSceneDev S;
    float cam[12];            // Position, Forward, UpScaled, RightScaled (Camera.fs:16-22)
    int32_t W, H, x0, nCols;
    uint32_t stripeW, stripeRanks, stripeRank, mode;    // mode 0: Image.render pixels, 1: explicit ray buffer
    float maxSize, eps, length, pad0;
    const ft_ray* rays;
    float* out;
    uint32_t* counter;        // global job cursor (zeroed before every launch)
    FtStatsDev* stats;
    uint32_t nJobs, chunk, tilesY;
    uint32_t cull;            // lean kernel: 1 = drop, per wave and round, the children whose terms are exact no-ops (kernels.hip "Exact child culling")
    // EXTENSIONS (spp = 1, aoSamples = 0 is the reference): sample plane s of the frame is written at
    // out + s * planeFloats and resolved by ft_resolve_kernel; ambient-occlusion rays per primary hit
    uint32_t spp, sppN, aoSamples, jobsPerPlane;
    float aoRadius; uint32_t planePixels;
    uint32_t ext;             // 1: launch the EXTENSION build of the kernel (set by the host, see capi.cpp)
    uint32_t maxBounces;      // EXTENSION glass: interactions per path; 0 = glass shades as a solid
    uint32_t spectral;        // EXTENSION: wavelength bins (0 = off)
    uint32_t lazy;            // 1: unions under an intersect stop at Items.[0] where the intersect's next child already decides (FT_OPT_LAZY_UNION; kernels.hip)
    uint32_t refillMin;       // idle lanes a wave waits for before it takes new rays (1 = refill at once; kernels.hip "Burst refill")
    uint32_t math;            // 0: the default kernels; 1: launch the *_libm build (FT_OPT_MATH = glibc and the scene has a unionSmooth)
    uint32_t shrink1, shrink2;   // guided hand-out: from job shrink1 on a wave takes chunk / 2 jobs at a time, from shrink2 on chunk / 4 (nJobs: never)
    uint32_t tailK;           // latency mode: a wave holding at most this many rays evaluates them one at a time with all 64 lanes (0 = off)
    const float* materialsExt;   // EXTENSION: 4 floats per material (glass flag, ior, dispersion, 0); kept out of
                                 // FtSceneDev so that the reference kernels' argument layout does not move
    float spec[16][4];        // per bin: RGB weight, Cauchy term (ft_spectral_table)
    uint32_t reuse;           // 1: a secondary ray's first evaluation — at the hit position — is the value the normal's centre probe computed there (FT_OPT_REUSE; kernels.hip FT_SH_D0)
    FtCarve carve;            // FtSceneDev.fastPath == 3: the union's tail and its terminated candidate lists (ft_device.h "Carved union")
};

#ifdef __cplusplus
extern "C" {
#endif
hipError_t ft_launch_trace(const FtRenderArgs* a, unsigned blocks, size_t ldsBytes, hipStream_t st);
hipError_t ft_launch_eval_points(const FtSceneDev* S, int math, const float* pts, long long n, float* outD, int* outM,
                                 unsigned blocks, size_t ldsBytes, hipStream_t st);
hipError_t ft_launch_math(int op, const float* x, const float* y, long long n, float* out, hipStream_t st);
// ft_selftest_libm: per chunk of 2^24 consecutive float bit patterns starting at lo, the sum of splitmix64(input bits << 32 | result bits)
// of the device restatement of glibc's expf (op 0) / logf (1) / powf(x, y) (2); variant 1 = FMA build, 2 = SSE2 build
hipError_t ft_launch_libm_checksum(int op, int variant, float y, uint32_t lo, uint32_t nChunks, unsigned long long* d_sums, hipStream_t st);
// device-side buildSpatialLookup (SdfBoundary.fs:245-274): one workgroup per cell
struct FtGridBuildArgs {
    const float* bounds;      // n x (cx, cy, cz, r)
    uint32_t n, c;            // items, cells per axis
    float aabbMin[3], cellSize[3], halfDiag;
    float* centers;           // ncells x 3
    uint32_t* counts;         // ncells
    FtItem* tmp;              // ncells x n, each cell's candidates sorted by (LowerBound, index)
    uint32_t* flags;          // bit0: NaN LowerBound, bit1: empty cell
};
hipError_t ft_launch_grid_build(const FtGridBuildArgs* a, hipStream_t st);
hipError_t ft_launch_grid_compact(const FtItem* tmp, const uint32_t* cellStart, uint32_t ncells, uint32_t n, FtItem* items, hipStream_t st);
#define FT_GRID_BUILD_MAX_ITEMS 4096
hipError_t ft_launch_resolve(const float* planes, float* out, unsigned long long nFloats, unsigned spp, hipStream_t st);
// Image.toColors (+ toBitmap order) on the device: max pass + map pass on one stream; maxBits = 4 bytes of device scratch
hipError_t ft_launch_tonemap(const float* frame, uint32_t X, uint32_t Y, uint32_t* maxBits, float gammaInv, uint32_t dither, uint32_t seed,
                             int bmpOrder, unsigned char* out, unsigned numCUs, int math, hipStream_t st);   // math: FT_OPT_MATH (0 fixed pow, 1 / 2 glibc powf)
// ft_render_multi: gathered slabs [rank][stripe][...] -> frame [stripe][rank][...] on the device
hipError_t ft_launch_deinterleave(const float* recv, float* frame, unsigned long long stripeFloats, uint32_t nStripes, uint32_t nRanks, hipStream_t st);
hipError_t ft_launch_selftest(int op, uint32_t lo, uint32_t hi, unsigned long long* d_mismatches, hipStream_t st);
hipError_t ft_trace_occupancy(unsigned fastPath, unsigned carveKind, bool ext, bool libm, size_t ldsBytes, int* blocksPerCU);
#ifdef __cplusplus
}
#endif
