// capi.cpp — the C ABI of libfraytracer_hip.so (include/fraytracer_hip.h).  Plain host C++;
// device code and launchers live in kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fraytracer_hip.h"
#include "build_hash.h"      // FT_SOURCE_HASH: written by the Makefile (source_hash.py)
#include "ft_kernels.h"
#include "ft_libm.h"         // FT_LIBM_TAB_DOUBLES (LDS footprint of the *_libm kernels)
#include "scene.hpp"

#ifndef FT_BUILD_KIND
#define FT_BUILD_KIND "product"
#endif

namespace {

thread_local std::string g_err;
int setErr(int code, const std::string& m) { g_err = m; return code; }
int hipFail(hipError_t e, const char* what) {
    return setErr(FT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hipFail(_e, #expr); } while (0)

inline f3 tof3(const float* p) { return mk3(p[0], p[1], p[2]); }
inline f3 tof3(const ft_vec3& v) { return mk3(v.x, v.y, v.z); }

}  // namespace

struct ft_ctx {
    int device = -1;
    bool hasDevice = false;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    int numCUs = 0;
    ft::Builder builder;
    ft::GridFiller* filler = nullptr;
    uint32_t* dCounter = nullptr;
    FtStatsDev* dStats = nullptr;
    void* scratch = nullptr; size_t scratchBytes = 0;     // staging for host-output entry points
    void* planes = nullptr; size_t planesBytes = 0;       // EXTENSION spp > 1: per-sample frames before the resolve
    void* aux = nullptr; size_t auxBytes = 0;             // tone map: [256 B: max bits | 8-bit image]
    hipStream_t lane1 = nullptr, copyStream = nullptr;    // ft_render's host-output pipeline: second render lane, DMA stream (lazily created)
    std::vector<hipEvent_t> syncEvents;                   // untimed events of that pipeline
    std::vector<std::pair<void*, size_t>> hostRegs;       // ranges pinned through ft_host_register
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events; // one pair per kernel launch since last collect (at most FT_MAX_PENDING_EVENTS)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> eventPool;
    double foldedMs = 0.0;                                 // kernel time of launches whose event pair was already recycled
    // ft_ctx_set_option (experiments / A-B runs; every setting renders the same bits)
    int optRefillMin = 64;                                 // idle lanes a wave waits for before it takes new rays (kernels.hip "Burst refill")
    int optMaxBlocksPerCU = 0;                             // 0: the occupancy limit
    int optHostChunks = 0;                                 // 0: automatic (4 for frames >= 16 MB)
    int optHostPin = 1;                                    // page-lock an unregistered ft_render destination for the call
    int optMath = FT_MATH_FIXED;                           // FT_OPT_MATH: arithmetic of MathF.Exp / Log / Pow
    int optTailK = -1;                                     // FT_OPT_TAIL_K: latency mode threshold (-1: per kernel default, 0: off)
    int optChunk = 64;                                     // FT_OPT_CHUNK: jobs per grab (experiments: 64 = one 8x8 tile, 32, 16)
    int optEscape = 1;                                     // FT_OPT_ESCAPE: rays that can no longer reach the scene's support sphere end as misses at once (kernels.hip ft_never_enters)
    int optLazyUnion = 1;                                  // FT_OPT_LAZY_UNION: a union under an intersect stops at Items.[0] where the intersect's next child decides (kernels.hip)
    int optCull = 1;                                       // FT_OPT_CULL: exact child culling in the lean kernel (kernels.hip); 0 = every child, every round
    int optReuse = 1;                                      // FT_OPT_REUSE: a secondary ray's first evaluation is taken from the normal's centre probe (kernels.hip FT_SH_D0); 0 = evaluated again, as the reference does
    int optCarved = 1;                                     // FT_OPT_CARVED: scenes of the "carved union" shape take their specialised kernel (kernels.hip ft_eval_carved); 0 = the general interpreter
    int optGuided = 0;                                     // FT_OPT_GUIDED: smaller chunks at the end of the job queue (lean kernel; measured: no gain)
};

struct ft_scene {
    ft_ctx* ctx = nullptr;
    ft::FlatScene flat;
    void* dBlob = nullptr;
    FtSceneDev dev{};
    const float* dMaterialsExt = nullptr;    // EXTENSION table, handed to the kernel through FtRenderArgs
    FtCarve carve{};                         // fastPath == 3: tail + device pointers of the terminated candidate lists
    bool usesExpLog = false;                 // the program has a unionSmooth (SdfForm.fs:80,82): the only place FT_OPT_MATH matters while tracing
};

namespace {

int requireDevice(ft_ctx* c) {
    if (!c) return setErr(FT_ERR_INVALID, "null context");
    if (!c->hasDevice) return setErr(FT_ERR_NO_DEVICE, "context has no GPU: libfraytracer_hip has no CPU fallback");
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return hipFail(e, "hipSetDevice");
    return FT_OK;
}

int ensureScratch(ft_ctx* c, size_t bytes) {
    if (bytes <= c->scratchBytes) return FT_OK;
    if (c->scratch) { HIP_TRY(hipFree(c->scratch)); c->scratch = nullptr; c->scratchBytes = 0; }
    HIP_TRY(hipMalloc(&c->scratch, bytes));
    c->scratchBytes = bytes;
    return FT_OK;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int ensureAux(ft_ctx* c, size_t bytes) {
    if (bytes <= c->auxBytes) return FT_OK;
    if (c->aux) { HIP_TRY(hipFree(c->aux)); c->aux = nullptr; c->auxBytes = 0; }
    HIP_TRY(hipMalloc(&c->aux, bytes));
    c->auxBytes = bytes;
    return FT_OK;
}

// Device-side per-cell grid build (kernels.hip ft_grid_build_kernel).  Declines (host build) for tiny grids,
// more than FT_GRID_BUILD_MAX_ITEMS items or a scratch need above 512 MB.
struct DeviceGridFiller : ft::GridFiller {
    ft_ctx* c;
    explicit DeviceGridFiller(ft_ctx* ctx) : c(ctx) {}
    bool fill(ft::HostGrid& g, const std::vector<ft::Boundary>& bounds, float halfDiag, std::string& err) override {
        const size_t n = bounds.size(), ncells = g.centers.size();
        const size_t tmpBytes = ncells * n * sizeof(FtItem);
        if (n > FT_GRID_BUILD_MAX_ITEMS || ncells * n < 100000 || tmpBytes > ((size_t)512 << 20)) return false;
        if (hipSetDevice(c->device) != hipSuccess) return false;
        const size_t oB = 0, oC = align256(oB + n * 16), oN = align256(oC + ncells * 12), oS = align256(oN + ncells * 4),
                     oF = align256(oS + (ncells + 1) * 4), oT = align256(oF + 256), total = oT + tmpBytes;
        if (ensureScratch(c, total) != FT_OK) return false;
        unsigned char* base = static_cast<unsigned char*>(c->scratch);
        std::vector<float> hb(n * 4);
        for (size_t i = 0; i < n; ++i) { hb[4 * i] = bounds[i].center.x; hb[4 * i + 1] = bounds[i].center.y; hb[4 * i + 2] = bounds[i].center.z; hb[4 * i + 3] = bounds[i].radius; }
        auto ok = [](hipError_t e) { return e == hipSuccess; };
        if (!ok(hipMemcpyAsync(base + oB, hb.data(), n * 16, hipMemcpyHostToDevice, c->stream)) ||
            !ok(hipMemsetAsync(base + oF, 0, 4, c->stream))) return false;
        FtGridBuildArgs a{};
        a.bounds = reinterpret_cast<const float*>(base + oB); a.n = (uint32_t)n; a.c = (uint32_t)g.count[0];
        a.aabbMin[0] = g.aabbMin.x; a.aabbMin[1] = g.aabbMin.y; a.aabbMin[2] = g.aabbMin.z;
        a.cellSize[0] = g.cellSize.x; a.cellSize[1] = g.cellSize.y; a.cellSize[2] = g.cellSize.z; a.halfDiag = halfDiag;
        a.centers = reinterpret_cast<float*>(base + oC); a.counts = reinterpret_cast<uint32_t*>(base + oN);
        a.tmp = reinterpret_cast<FtItem*>(base + oT); a.flags = reinterpret_cast<uint32_t*>(base + oF);
        if (!ok(ft_launch_grid_build(&a, c->stream))) return false;
        std::vector<uint32_t> counts(ncells);
        uint32_t flags = 0;
        if (!ok(hipMemcpyAsync(counts.data(), base + oN, ncells * 4, hipMemcpyDeviceToHost, c->stream)) ||
            !ok(hipMemcpyAsync(&flags, base + oF, 4, hipMemcpyDeviceToHost, c->stream)) ||
            !ok(hipMemcpyAsync(g.centers.data(), base + oC, ncells * 12, hipMemcpyDeviceToHost, c->stream)) ||
            !ok(hipStreamSynchronize(c->stream))) return false;
        if (flags & 1u) { err = "union: NaN boundary"; return false; }
        if (flags & 2u) { err = "union: a lookup cell has no candidates (the reference would throw at Items.[0])"; return false; }
        uint32_t total_items = 0;
        for (size_t ci = 0; ci < ncells; ++ci) { g.cellStart[ci] = total_items; total_items += counts[ci]; }
        g.cellStart[ncells] = total_items;
        g.items.resize(total_items);
        FtItem* dItems = nullptr;                                       // CSR items, exact size
        if (!ok(hipMalloc((void**)&dItems, std::max<size_t>(16, (size_t)total_items * sizeof(FtItem))))) return false;
        const bool done =
            ok(hipMemcpyAsync(base + oS, g.cellStart.data(), (ncells + 1) * 4, hipMemcpyHostToDevice, c->stream)) &&
            ok(ft_launch_grid_compact(a.tmp, reinterpret_cast<const uint32_t*>(base + oS), (uint32_t)ncells, (uint32_t)n, dItems, c->stream)) &&
            ok(hipMemcpyAsync(g.items.data(), dItems, (size_t)total_items * sizeof(FtItem), hipMemcpyDeviceToHost, c->stream)) &&
            ok(hipStreamSynchronize(c->stream));
        (void)hipFree(dItems);
        return done;
    }
};

template <class T> size_t placed(size_t& cursor, const std::vector<T>& v) {
    const size_t at = cursor;
    cursor = align256(cursor + std::max<size_t>(v.size() * sizeof(T), 16));
    return at;
}

int uploadScene(ft_ctx* c, ft_scene* s) {
    const ft::FlatScene& f = s->flat;
    size_t cur = 0;
    const size_t oInstr = placed(cur, f.instr), oConsts = placed(cur, f.consts), oGrids = placed(cur, f.grids),
                 oKids = placed(cur, f.children), oCtr = placed(cur, f.cellCenters), oStart = placed(cur, f.cellStart),
                 oItems = placed(cur, f.items), oLights = placed(cur, f.lights), oMats = placed(cur, f.materials),
                 oMatX = placed(cur, f.materialsExt), oItemsT = placed(cur, f.itemsT), oStartT = placed(cur, f.cellStartT);
    std::vector<unsigned char> host(cur, 0);
    auto put = [&](size_t at, const void* p, size_t n) { if (n) memcpy(host.data() + at, p, n); };
    put(oInstr, f.instr.data(), f.instr.size() * sizeof(FtInstr));
    put(oConsts, f.consts.data(), f.consts.size() * 4);
    put(oGrids, f.grids.data(), f.grids.size() * sizeof(FtGrid));
    put(oKids, f.children.data(), f.children.size() * sizeof(FtChild));
    put(oCtr, f.cellCenters.data(), f.cellCenters.size() * 4);
    put(oStart, f.cellStart.data(), f.cellStart.size() * 4);
    put(oItems, f.items.data(), f.items.size() * sizeof(FtItemRec));
    put(oLights, f.lights.data(), f.lights.size() * sizeof(FtLight));
    put(oMats, f.materials.data(), f.materials.size() * 4);
    put(oMatX, f.materialsExt.data(), f.materialsExt.size() * 4);
    put(oItemsT, f.itemsT.data(), f.itemsT.size() * sizeof(FtItemRec));
    put(oStartT, f.cellStartT.data(), f.cellStartT.size() * 4);
    FtSceneDev& d = s->dev;
    d = FtSceneDev{};
    d.nInstr = f.nMainInstr; d.nSlots = f.nSlots; d.nLights = (uint32_t)f.lights.size(); d.fastPath = f.fastPath;
    d.bg[0] = f.bg[0]; d.bg[1] = f.bg[1]; d.bg[2] = f.bg[2];
    d.nStage = f.nStage; d.nearR2 = f.nearR2; d.fastQ = f.fastQ; d.nGlass = f.nGlass;
    d.escC[0] = f.escC[0]; d.escC[1] = f.escC[1]; d.escC[2] = f.escC[2]; d.escR = f.escR; d.escRho2 = f.escRho2; d.cullPc = f.cullPc;
    if (!c->hasDevice) return FT_OK;                       // host-only context: introspection only
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(&s->dBlob, cur));
    HIP_TRY(hipMemcpy(s->dBlob, host.data(), cur, hipMemcpyHostToDevice));
    unsigned char* b = static_cast<unsigned char*>(s->dBlob);
    d.instr = reinterpret_cast<const FtInstr*>(b + oInstr);
    d.consts = reinterpret_cast<const float*>(b + oConsts);
    d.grids = reinterpret_cast<const FtGrid*>(b + oGrids);
    d.children = reinterpret_cast<const FtChild*>(b + oKids);
    d.cellCenters = reinterpret_cast<const float*>(b + oCtr);
    d.cellStart = reinterpret_cast<const uint32_t*>(b + oStart);
    d.items = reinterpret_cast<const FtItemRec*>(b + oItems);
    d.lights = reinterpret_cast<const FtLight*>(b + oLights);
    d.materials = reinterpret_cast<const float*>(b + oMats);
    s->dMaterialsExt = reinterpret_cast<const float*>(b + oMatX);
    s->carve = f.carve;
    s->carve.itemsT = reinterpret_cast<const FtItemRec*>(b + oItemsT);
    s->carve.cellStartT = reinterpret_cast<const uint32_t*>(b + oStartT);
    return FT_OK;
}

// statistics header, per-lane value slots (distance + material index), the staged constant-pool prefix; the *_libm kernels keep glibc's
// tables behind that, 8-byte aligned (kernels.hip ft_libm_lds_offset)
// (kernels.hip ft_libm_lds_offset); the lean kernel keeps one row of FT_COOP_SEG floats per wave behind everything (16-byte aligned) for
// the latency mode (kernels.hip ft_coop_lds_offset)
#define FT_COOP_SEG_FLOATS FT_CULL_ROW        // the wave's row of culled children's records (ft_kernels.h); its first 256 floats serve the latency mode
// traceLaunch: the lean trace kernel keeps its accumulator in a register and is launched with nSlots = 0 (launchTrace) — 2 KB per workgroup
// that decide between 6 and 7 resident workgroups per CU; every other user of a lean scene (ft_eval_distance) runs the general interpreter
// variant: the kernel family of a trace launch (launchTrace: FtSceneDev.fastPath, or 0 where a carved scene takes the general kernel)
size_t ldsBytes(const ft_scene* s, bool libm = false, bool traceLaunch = false, unsigned variant = 0, bool cullRows = true) {
    const size_t nSlots = (traceLaunch && (variant == 1u || variant == 3u)) ? 0 : s->dev.nSlots;       // the lean and the carved kernels keep their values in registers
    size_t floats = (size_t)FT_LDS_HDR_FLOATS + nSlots * FT_BLOCK * 2 + (size_t)s->dev.nStage;
    if (libm) floats = ((floats + 1) & ~(size_t)1) + (size_t)FT_LIBM_TAB_DOUBLES * 2;
    // one row per wave behind everything: the lean kernel's latency mode and culled children; any other trace kernel's culled children where the scene has a cull site
    if (s->dev.fastPath == 1u || (cullRows && traceLaunch && variant != 3u && s->dev.cullPc != 0xffffffffu)) floats = ((floats + 3) & ~(size_t)3) + (size_t)FT_COOP_SEG_FLOATS * (FT_BLOCK / 64);
    return floats * 4;
}
// Latency-mode thresholds (rays per wave at or below which each ray is evaluated by all 64 lanes; measured)
constexpr int FT_TAIL_K_LEAN = 32, FT_TAIL_K_GENERAL = 2, FT_TAIL_K_CARVED = 1;      // carved: 1.26 ms at 0 / 1 against 1.29 at 2 on the 1000^2 Program.fs frame (profiles/r04_carved_variants.txt)
// does this launch take the glibc build of the kernels?
bool libmLaunch(const ft_ctx* c, const ft_scene* s) { return c->optMath != FT_MATH_FIXED && s->usesExpLog; }

// A frame loop that never calls ft_collect_stats must not grow the event list: beyond this many pending pairs the
// oldest one is folded into foldedMs (it has long completed: launches on one stream finish in order) and recycled.
constexpr size_t FT_MAX_PENDING_EVENTS = 64;
int foldOldestEvents(ft_ctx* c) {
    while (c->events.size() >= FT_MAX_PENDING_EVENTS) {
        auto p = c->events.front();
        HIP_TRY(hipEventSynchronize(p.second));
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, p.first, p.second));
        c->foldedMs += t;
        c->eventPool.push_back(p);
        c->events.erase(c->events.begin());
    }
    return FT_OK;
}

int acquireEvents(ft_ctx* c, hipEvent_t& a, hipEvent_t& b) {
    if (!c->eventPool.empty()) { a = c->eventPool.back().first; b = c->eventPool.back().second; c->eventPool.pop_back(); return FT_OK; }
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    return FT_OK;
}

// launch the persistent trace kernel over nJobs jobs
// lane 0 = the context's stream; lane 1 = a second stream with its own job counter, so that two launches can be in flight
// (the drain of one overlaps the start of the next: DESIGN.md section 6)
int launchTrace(ft_ctx* c, const ft_scene* s, FtRenderArgs& a, int lane = 0) {
    hipStream_t stream = lane ? c->lane1 : c->stream;
    uint32_t* counter = c->dCounter + (lane ? 16 : 0);
    int perCU = 0;
    const bool libm = libmLaunch(c, s);
    // kernel family: a "carved union" scene takes the general kernels for EXTENSION launches and with FT_OPT_CARVED = 0
    const unsigned variant = (s->dev.fastPath == 3u && (a.ext != 0u || !c->optCarved)) ? 0u : s->dev.fastPath;
    size_t lds = ldsBytes(s, libm, true, variant);
    bool cullRows = true;
    {   // a scene too large for the workgroup's LDS is refused here, not by a launch failure
        int maxLds = 0;
        HIP_TRY(hipDeviceGetAttribute(&maxLds, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device));
        if (lds > (size_t)maxLds && variant != 1u && s->dev.cullPc != 0xffffffffu) {       // the general kernels' culling rows are optional: without them the pass is off
            cullRows = false;
            lds = ldsBytes(s, libm, true, variant, false);
        }
        if (lds > (size_t)maxLds) return setErr(FT_ERR_UNSUPPORTED, "scene needs " + std::to_string(lds) + " bytes of LDS per workgroup; the device offers " + std::to_string(maxLds));
    }
    HIP_TRY(ft_trace_occupancy(variant, s->carve.kind, a.ext != 0u, libm, lds, &perCU));
    if (perCU < 1) return setErr(FT_ERR_UNSUPPORTED, "the trace kernel does not fit a compute unit with this scene's LDS footprint");
    perCU = std::min(perCU, 8);
    if (c->optMaxBlocksPerCU > 0) perCU = std::min(perCU, c->optMaxBlocksPerCU);       // FT_OPT_MAX_BLOCKS_PER_CU (experiments)
    else if (variant != 1u) {
        // Small frames: the job queue can only even out the load while there are several tiles per resident wave, and the grid-union kernels lose little
        // throughput at half their occupancy (the 4000^2 Program.fs frame: 5.4 ms at 7 workgroups per CU, 6.0 at 4).  With about as many tiles as waves
        // every tile is handed out at once, the heavy ones sit several deep on some SIMDs while others idle, and the frame lasts as long as the
        // slowest of them at full contention.  So: about one resident wave per four tiles, at least two workgroups per CU (measured per size,
        // profiles/r04_blocks_by_frame_size.jsonl: the reference's own 1000^2 frame 1.45 -> 1.27 ms, C2 at 1024^2 0.90 -> 0.61 ms).
        const uint64_t tiles = ((uint64_t)a.nJobs + (uint64_t)c->optChunk - 1) / (uint64_t)c->optChunk;
        const uint64_t wavesPerLayer = (uint64_t)c->numCUs * (FT_BLOCK / 64);
        const uint64_t want = (tiles + 2 * wavesPerLayer) / (4 * wavesPerLayer);              // round(tiles / (4 x waves of one workgroup per CU))
        perCU = (int)std::min<uint64_t>((uint64_t)perCU, std::max<uint64_t>(2, want));
    }
    const uint64_t maxBlocks = (uint64_t)c->numCUs * perCU;
    const uint64_t wantBlocks = ((uint64_t)a.nJobs + FT_BLOCK - 1) / FT_BLOCK;
    const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min(maxBlocks, wantBlocks));
    // one 8x8 tile per grab: measured faster than larger chunks (lanes of a wave stay on neighbouring
    // pixels) and 2.6e5 atomics per 4096^2 frame are far below the rate one counter sustains
    a.chunk = (uint32_t)c->optChunk;
    // burst refill (kernels.hip): a wave takes new rays only when all 64 lanes are idle, i.e. it works through one 8x8 tile at a
    // time.  Measured (profiles/r02_refill_sweep.txt, kernel ms at refillMin = 1 / 32 / 64): 1000-torus scene 4000^2 22.5 /
    // 15.8 / 12.1, at 1000^2 2.98 / 2.72 / 2.24, C2 4096^2 4.90 / 4.40 / 3.84, 300 on-demand combinators 24.6 / 14.8 / 10.3,
    // glass config 39.1 / 31.5 / 27.3, and even the VALU-bound C3 kernel 61.0 / 66.5 / 60.3: rays that start together stay in
    // step (march, the four normal probes, shadow rays), so a wave's lanes share lookup cells, list positions and branches.
    a.refillMin = (uint32_t)c->optRefillMin;               // 64 unless FT_OPT_REFILL_MIN says otherwise (experiments)
    a.tailK = (uint32_t)(c->optTailK >= 0 ? c->optTailK : (variant == 1u ? FT_TAIL_K_LEAN : variant == 3u ? FT_TAIL_K_CARVED : FT_TAIL_K_GENERAL));
    // guided hand-out of the last jobs (kernels.hip refill): one half tile, then one quarter tile per resident wave — only where the latency
    // mode makes a part-filled wave cheap (lean kernel, tailK >= 32) and only for the reference's sampling (tile-major job order)
    a.shrink1 = a.shrink2 = a.nJobs;
    if (s->dev.fastPath == 1u && a.tailK >= 32u && c->optGuided && a.mode == 0u) {
        const uint64_t waves = (uint64_t)blocks * (FT_BLOCK / 64);
        const uint64_t q = waves * (a.chunk / 4u), h = waves * (a.chunk / 2u);
        if ((uint64_t)a.nJobs > 4u * (q + h)) { a.shrink2 = (uint32_t)(a.nJobs - q); a.shrink1 = (uint32_t)(a.nJobs - q - h); }
    }
    a.counter = counter;
    a.stats = c->dStats;
    a.S = s->dev;
    a.S.fastPath = variant;
    a.carve = s->carve;
    if (variant == 1u || variant == 3u) a.S.nSlots = 0;    // the lean and carved kernels use no value slots: their LDS layout has none (ldsBytes)
    a.math = libm ? 1u : 0u;
    a.cull = (s->dev.cullPc != 0xffffffffu && variant != 3u && cullRows && c->optCull) ? 1u : 0u;
    if (!c->optEscape) a.S.escR = -1.0f;
    a.lazy = c->optLazyUnion ? 1u : 0u;
    a.reuse = c->optReuse ? 1u : 0u;
    a.S.mathFma = c->optMath == FT_MATH_GLIBC_FMA ? 1u : 0u;
    a.materialsExt = s->dMaterialsExt;
    HIP_TRY(hipMemsetAsync(counter, 0, sizeof(uint32_t), stream));
    hipEvent_t e0, e1;
    int rc = foldOldestEvents(c); if (rc) return rc;
    if ((rc = acquireEvents(c, e0, e1))) return rc;
    HIP_TRY(hipEventRecord(e0, stream));
    HIP_TRY(ft_launch_trace(&a, blocks, lds, stream));
    HIP_TRY(hipEventRecord(e1, stream));
    c->events.emplace_back(e0, e1);
    return FT_OK;
}

int checkParams(const ft_render_params* p) {
    if (!p) return setErr(FT_ERR_INVALID, "null render params");
    if (p->width <= 0 || p->height <= 0 || p->n_columns <= 0) return setErr(FT_ERR_INVALID, "empty image");
    if (p->stripe_width <= 0 || p->stripe_ranks <= 0 || p->stripe_rank < 0 || p->stripe_rank >= p->stripe_ranks)
        return setErr(FT_ERR_INVALID, "bad stripe description");
    int sn = 1; while (sn * sn < p->spp) ++sn;
    if (p->spp < 1 || p->spp > 64 || sn * sn != p->spp) return setErr(FT_ERR_INVALID, "spp (extension) must be a square number <= 64");
    if (p->ao_samples < 0 || p->ao_samples > 16) return setErr(FT_ERR_INVALID, "ao_samples (extension) must be in [0, 16]");
    if (p->max_bounces < 0 || p->max_bounces > 64) return setErr(FT_ERR_INVALID, "max_bounces (extension) must be in [0, 64]");
    if (p->spectral < 0 || p->spectral > 16 || (p->spectral > 0 && p->spp % p->spectral != 0))
        return setErr(FT_ERR_INVALID, "spectral (extension) must be in [0, 16] and divide spp");
    // last local column must map inside the image
    const int64_t c = (int64_t)p->n_columns - 1;
    const int64_t x = p->x0 + (c / p->stripe_width) * (int64_t)p->stripe_width * p->stripe_ranks + (int64_t)p->stripe_rank * p->stripe_width + c % p->stripe_width;
    if (p->x0 < 0 || x >= p->width) return setErr(FT_ERR_INVALID, "column range leaves the image");
    const uint64_t tiles = (uint64_t)((p->n_columns + 7) / 8) * (uint64_t)((p->height + 7) / 8);
    if (tiles * 64 * (uint64_t)p->spp >= 0xFFFF0000ull) return setErr(FT_ERR_UNSUPPORTED, "more than 2^32 samples in one call");
    return FT_OK;
}

}  // namespace

extern "C" {

int ft_abi_version(void) { return FT_ABI_VERSION; }
const char* ft_last_error(void) { return g_err.c_str(); }
const char* ft_build_info(void) { return "src=" FT_SOURCE_HASH ";kind=" FT_BUILD_KIND; }

int ft_ctx_set_option(ft_ctx* c, int32_t option, int32_t value) {
    if (!c) return setErr(FT_ERR_INVALID, "null context");
    switch (option) {
    case FT_OPT_REFILL_MIN: if (value < 1 || value > 64) return setErr(FT_ERR_INVALID, "FT_OPT_REFILL_MIN: 1 .. 64"); c->optRefillMin = value; return FT_OK;
    case FT_OPT_MAX_BLOCKS_PER_CU: if (value < 0 || value > 8) return setErr(FT_ERR_INVALID, "FT_OPT_MAX_BLOCKS_PER_CU: 0 (no cap) .. 8"); c->optMaxBlocksPerCU = value; return FT_OK;
    case FT_OPT_HOST_CHUNKS: if (value < 0 || value > 16) return setErr(FT_ERR_INVALID, "FT_OPT_HOST_CHUNKS: 0 (automatic) .. 16"); c->optHostChunks = value; return FT_OK;
    case FT_OPT_HOST_PIN: if (value != 0 && value != 1) return setErr(FT_ERR_INVALID, "FT_OPT_HOST_PIN: 0 or 1"); c->optHostPin = value; return FT_OK;
    case FT_OPT_TAIL_K: if (value < -1 || value > 64) return setErr(FT_ERR_INVALID, "FT_OPT_TAIL_K: -1 (default), 0 (off) .. 64"); c->optTailK = value; return FT_OK;
    case FT_OPT_CHUNK: if (value != 64 && value != 32 && value != 16) return setErr(FT_ERR_INVALID, "FT_OPT_CHUNK: 64, 32 or 16"); c->optChunk = value; return FT_OK;
    case FT_OPT_ESCAPE: if (value != 0 && value != 1) return setErr(FT_ERR_INVALID, "FT_OPT_ESCAPE: 0 or 1"); c->optEscape = value; return FT_OK;
    case FT_OPT_LAZY_UNION: if (value != 0 && value != 1) return setErr(FT_ERR_INVALID, "FT_OPT_LAZY_UNION: 0 or 1"); c->optLazyUnion = value; return FT_OK;
    case FT_OPT_CULL: if (value != 0 && value != 1) return setErr(FT_ERR_INVALID, "FT_OPT_CULL: 0 or 1"); c->optCull = value; return FT_OK;
    case FT_OPT_GUIDED: if (value != 0 && value != 1) return setErr(FT_ERR_INVALID, "FT_OPT_GUIDED: 0 or 1"); c->optGuided = value; return FT_OK;
    case FT_OPT_CARVED: if (value != 0 && value != 1) return setErr(FT_ERR_INVALID, "FT_OPT_CARVED: 0 or 1"); c->optCarved = value; return FT_OK;
    case FT_OPT_REUSE: if (value != 0 && value != 1) return setErr(FT_ERR_INVALID, "FT_OPT_REUSE: 0 or 1"); c->optReuse = value; return FT_OK;
    case FT_OPT_MATH:
        if (value != FT_MATH_FIXED && value != FT_MATH_GLIBC_FMA && value != FT_MATH_GLIBC_SSE2) return setErr(FT_ERR_INVALID, "FT_OPT_MATH: 0 fixed, 1 glibc (FMA build), 2 glibc (SSE2 build)");
        c->optMath = value; return FT_OK;
    default: return setErr(FT_ERR_INVALID, "unknown option");
    }
}
int ft_ctx_get_option(const ft_ctx* c, int32_t option, int32_t* value) {
    if (!c || !value) return setErr(FT_ERR_INVALID, "null argument");
    switch (option) {
    case FT_OPT_REFILL_MIN: *value = c->optRefillMin; return FT_OK;
    case FT_OPT_MAX_BLOCKS_PER_CU: *value = c->optMaxBlocksPerCU; return FT_OK;
    case FT_OPT_HOST_CHUNKS: *value = c->optHostChunks; return FT_OK;
    case FT_OPT_HOST_PIN: *value = c->optHostPin; return FT_OK;
    case FT_OPT_MATH: *value = c->optMath; return FT_OK;
    case FT_OPT_TAIL_K: *value = c->optTailK; return FT_OK;
    case FT_OPT_GUIDED: *value = c->optGuided; return FT_OK;
    case FT_OPT_CARVED: *value = c->optCarved; return FT_OK;
    case FT_OPT_REUSE: *value = c->optReuse; return FT_OK;
    case FT_OPT_CULL: *value = c->optCull; return FT_OK;
    case FT_OPT_LAZY_UNION: *value = c->optLazyUnion; return FT_OK;
    case FT_OPT_ESCAPE: *value = c->optEscape; return FT_OK;
    case FT_OPT_CHUNK: *value = c->optChunk; return FT_OK;
    default: return setErr(FT_ERR_INVALID, "unknown option");
    }
}

int ft_ctx_create(int device, ft_ctx** out) {
    if (!out) return setErr(FT_ERR_INVALID, "null out pointer");
    *out = nullptr;
    ft_ctx* c = new ft_ctx();
    c->device = device;
    if (device >= 0) {
        // every failure below leaves through ft_ctx_destroy, which releases whatever was created so far
        auto fail = [&](int rc) { c->hasDevice = c->stream != nullptr || c->dCounter != nullptr || c->dStats != nullptr; ft_ctx_destroy(c); return rc; };
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) return fail(setErr(FT_ERR_NO_DEVICE, "no HIP device visible (libfraytracer_hip has no CPU fallback)"));
        if (device >= n) return fail(setErr(FT_ERR_INVALID, "device ordinal out of range"));
        if ((e = hipSetDevice(device)) != hipSuccess) return fail(hipFail(e, "hipSetDevice"));
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail(hipFail(e, "hipGetDeviceProperties"));
        c->numCUs = prop.multiProcessorCount;
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { c->stream = nullptr; return fail(hipFail(e, "hipStreamCreate")); }
        c->ownStream = true;
        if ((e = hipMalloc((void**)&c->dCounter, 256)) != hipSuccess) { c->dCounter = nullptr; return fail(hipFail(e, "hipMalloc")); }
        if ((e = hipMalloc((void**)&c->dStats, sizeof(FtStatsDev))) != hipSuccess) { c->dStats = nullptr; return fail(hipFail(e, "hipMalloc")); }
        if ((e = hipMemsetAsync(c->dStats, 0, sizeof(FtStatsDev), c->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(c->stream)) != hipSuccess) return fail(hipFail(e, "hipMemset"));
        c->hasDevice = true;
        c->filler = new DeviceGridFiller(c);
        c->builder.gridFiller = c->filler;
    }
    *out = c;
    return FT_OK;
}

void ft_ctx_destroy(ft_ctx* c) {
    if (!c) return;
    if (c->hasDevice) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        for (auto& p : c->events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        for (auto& p : c->eventPool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        if (c->scratch) (void)hipFree(c->scratch);
        if (c->planes) (void)hipFree(c->planes);
        if (c->aux) (void)hipFree(c->aux);
        for (auto& r : c->hostRegs) (void)hipHostUnregister(r.first);
        for (auto e : c->syncEvents) (void)hipEventDestroy(e);
        if (c->lane1) { (void)hipStreamSynchronize(c->lane1); (void)hipStreamDestroy(c->lane1); }
        if (c->copyStream) { (void)hipStreamSynchronize(c->copyStream); (void)hipStreamDestroy(c->copyStream); }
        if (c->dCounter) (void)hipFree(c->dCounter);
        if (c->dStats) (void)hipFree(c->dStats);
        if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c->filler;
    delete c;
}

int ft_ctx_set_stream(ft_ctx* c, void* hip_stream) {
    int rc = requireDevice(c); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->ownStream && c->stream) { HIP_TRY(hipStreamDestroy(c->stream)); }
    if (hip_stream) { c->stream = static_cast<hipStream_t>(hip_stream); c->ownStream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->ownStream = true; }
    return FT_OK;
}

// ---- scene construction ------------------------------------------------------------------------
#define CTX_OR_FAIL(c) do { if (!(c)) return setErr(FT_ERR_INVALID, "null context"); } while (0)
static int builderResult(ft_ctx* c, int h) { if (h < 0) g_err = c->builder.err; return h; }

ft_handle ft_form_sphere(ft_ctx* c, const ft_sphere* s) {
    CTX_OR_FAIL(c); if (!s) return setErr(FT_ERR_INVALID, "null primitive");
    return c->builder.sphere(tof3(s->center), s->radius);
}
ft_handle ft_form_capsule(ft_ctx* c, const ft_capsule* s) {
    CTX_OR_FAIL(c); if (!s) return setErr(FT_ERR_INVALID, "null primitive");
    return c->builder.capsule(tof3(s->from), tof3(s->to), s->radius);
}
ft_handle ft_form_torus(ft_ctx* c, const ft_torus* s) {
    CTX_OR_FAIL(c); if (!s) return setErr(FT_ERR_INVALID, "null primitive");
    return c->builder.torus(tof3(s->center), tof3(s->normal), s->major_radius, s->minor_radius);
}
ft_handle ft_form_triangle(ft_ctx* c, const ft_triangle* s) {
    CTX_OR_FAIL(c); if (!s) return setErr(FT_ERR_INVALID, "null primitive");
    return c->builder.triangle(tof3(s->v1), tof3(s->v2), tof3(s->v3), s->radius);
}
ft_handle ft_form_box(ft_ctx* c, const ft_box* s) {
    CTX_OR_FAIL(c); if (!s) return setErr(FT_ERR_INVALID, "null primitive");
    return c->builder.box(tof3(s->center), tof3(s->half_extent));
}
ft_handle ft_form_union(ft_ctx* c, const ft_handle* forms, int32_t n) { CTX_OR_FAIL(c); return builderResult(c, c->builder.formUnion(forms, n)); }
ft_handle ft_form_subtract(ft_ctx* c, ft_handle a, ft_handle b) { CTX_OR_FAIL(c); return builderResult(c, c->builder.formSubtract(a, b)); }
ft_handle ft_form_intersect(ft_ctx* c, const ft_handle* forms, int32_t n) { CTX_OR_FAIL(c); return builderResult(c, c->builder.formIntersect(forms, n)); }
ft_handle ft_form_union_smooth(ft_ctx* c, float strength, const ft_handle* forms, int32_t n) {
    CTX_OR_FAIL(c); return builderResult(c, c->builder.formUnionSmooth(strength, forms, n));
}
int ft_form_boundary(ft_ctx* c, ft_handle form, ft_boundary* out) {
    CTX_OR_FAIL(c);
    if (!c->builder.okForm(form) || !out) return setErr(FT_ERR_INVALID, "invalid form handle");
    const ft::Boundary& b = c->builder.forms[form].boundary;
    out->center.x = b.center.x; out->center.y = b.center.y; out->center.z = b.center.z; out->radius = b.radius;
    return FT_OK;
}
ft_handle ft_material_solid(ft_ctx* c, const float rgb[3]) { CTX_OR_FAIL(c); if (!rgb) return setErr(FT_ERR_INVALID, "null colour"); return c->builder.materialSolid(tof3(rgb)); }
ft_handle ft_material_glass(ft_ctx* c, const float tint[3], float ior, float dispersion) {
    CTX_OR_FAIL(c); if (!tint) return setErr(FT_ERR_INVALID, "null colour");
    return builderResult(c, c->builder.materialGlass(tof3(tint), ior, dispersion));
}
int ft_spectral_table(int32_t nw, float* out) {
    if (!out || nw < 1 || nw > 16) return setErr(FT_ERR_INVALID, "1 <= nw <= 16 and a buffer of nw x 4 floats");
    float t[16][4];
    ft::spectralTable(nw, t);
    memcpy(out, t, sizeof(float) * 4 * (size_t)nw);
    return FT_OK;
}
ft_handle ft_object_create(ft_ctx* c, ft_handle material, ft_handle form) { CTX_OR_FAIL(c); return builderResult(c, c->builder.objectCreate(material, form)); }
ft_handle ft_object_union(ft_ctx* c, const ft_handle* objs, int32_t n) { CTX_OR_FAIL(c); return builderResult(c, c->builder.objectUnion(objs, n)); }
ft_handle ft_object_subtract(ft_ctx* c, ft_handle obj, ft_handle form) { CTX_OR_FAIL(c); return builderResult(c, c->builder.objectSubtract(obj, form)); }
ft_handle ft_object_intersect(ft_ctx* c, ft_handle obj, const ft_handle* forms, int32_t n) {
    CTX_OR_FAIL(c); return builderResult(c, c->builder.objectIntersect(obj, forms, n));
}
ft_handle ft_object_form(ft_ctx* c, ft_handle obj) {
    CTX_OR_FAIL(c);
    if (!c->builder.okObject(obj)) return setErr(FT_ERR_INVALID, "invalid object handle");
    return c->builder.objects[obj].form;
}
ft_handle ft_light_directional(ft_ctx* c, const float d[3], const float rgb[3]) {
    CTX_OR_FAIL(c); if (!d || !rgb) return setErr(FT_ERR_INVALID, "null argument");
    return c->builder.lightDirectional(tof3(d), tof3(rgb));
}
ft_handle ft_light_point(ft_ctx* c, const float p[3], const float rgb[3]) {
    CTX_OR_FAIL(c); if (!p || !rgb) return setErr(FT_ERR_INVALID, "null argument");
    return c->builder.lightPoint(tof3(p), tof3(rgb));
}

int ft_scene_create(ft_ctx* c, ft_handle object, const float bg[3], const ft_handle* lights, int32_t n, ft_scene** out) {
    CTX_OR_FAIL(c);
    if (!out || !bg || n < 0 || (n > 0 && !lights)) return setErr(FT_ERR_INVALID, "bad argument");
    *out = nullptr;
    ft_scene* s = new ft_scene();
    s->ctx = c;
    std::string err;
    if (!ft::flatten(c->builder, object, bg, lights, n, s->flat, err)) { delete s; return setErr(FT_ERR_UNSUPPORTED, err); }
    // the union walk addresses candidate records and the constant pool with 32-bit byte offsets (kernels.hip ld_item_at / pool_at)
    if (s->flat.items.size() >= (1ull << 27) || s->flat.consts.size() >= (1ull << 30)) {
        delete s; return setErr(FT_ERR_UNSUPPORTED, "scene has 2^27 or more (cell, candidate) records or 2^30 or more constants");
    }
    for (const FtInstr& in : s->flat.instr)
        if (in.op == FT_OP_SMOOTH_RUN || in.op == FT_OP_SMOOTH_ADD || in.op == FT_OP_SMOOTH_FIN) s->usesExpLog = true;
    int rc = uploadScene(c, s);
    if (rc) { delete s; return rc; }
    *out = s;
    return FT_OK;
}

int ft_scene_clone(const ft_scene* src, ft_ctx* dst, ft_scene** out) {
    if (!src || !dst || !out) return setErr(FT_ERR_INVALID, "bad argument");
    ft_scene* s = new ft_scene();
    s->ctx = dst;
    s->flat = src->flat;
    s->usesExpLog = src->usesExpLog;
    int rc = uploadScene(dst, s);
    if (rc) { delete s; return rc; }
    *out = s;
    return FT_OK;
}

void ft_scene_destroy(ft_scene* s) {
    if (!s) return;
    if (s->dBlob) { (void)hipSetDevice(s->ctx->device); (void)hipFree(s->dBlob); }
    delete s;
}

float ft_lens_create(float fov) { return ft::lensCreate(fov); }
int ft_camera_look_at(const float pos[3], const float look[3], const float up[3], float nps, ft_camera* out) {
    if (!pos || !look || !up || !out) return setErr(FT_ERR_INVALID, "null argument");
    f3 o[4];
    ft::cameraLookAt(tof3(pos), tof3(look), tof3(up), nps, o);
    out->position = ft_vec3{o[0].x, o[0].y, o[0].z}; out->forward = ft_vec3{o[1].x, o[1].y, o[1].z};
    out->up_scaled = ft_vec3{o[2].x, o[2].y, o[2].z}; out->right_scaled = ft_vec3{o[3].x, o[3].y, o[3].z};
    return FT_OK;
}

// ---- hot path ------------------------------------------------------------------------------------
static int renderLane(ft_ctx* c, const ft_scene* s, const ft_camera* cam, const ft_render_params* p, void* d_out, int lane) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!s || s->ctx != c || !cam || !d_out) return setErr(FT_ERR_INVALID, "bad argument (scene must belong to this context)");
    if ((rc = checkParams(p))) return rc;
    if (lane != 0 && p->spp != 1) return setErr(FT_ERR_INVALID, "internal: the sample planes belong to lane 0");
    FtRenderArgs a{};
    memcpy(a.cam, cam, sizeof(float) * 12);
    a.W = p->width; a.H = p->height; a.x0 = p->x0; a.nCols = p->n_columns;
    a.stripeW = (uint32_t)p->stripe_width; a.stripeRanks = (uint32_t)p->stripe_ranks; a.stripeRank = (uint32_t)p->stripe_rank;
    a.mode = 0;
    a.maxSize = (float)std::max(p->width, p->height);              // Image.fs:18
    a.eps = p->epsilon; a.length = p->length;
    a.tilesY = (uint32_t)((p->height + 7) / 8);
    a.jobsPerPlane = (uint32_t)((p->n_columns + 7) / 8) * a.tilesY * 64u;
    a.spp = (uint32_t)p->spp; a.sppN = 1; while (a.sppN * a.sppN < a.spp) ++a.sppN;
    a.aoSamples = (uint32_t)p->ao_samples; a.aoRadius = p->ao_radius;
    a.planePixels = (uint32_t)p->n_columns * (uint32_t)p->height;
    a.nJobs = a.jobsPerPlane * a.spp;
    // EXTENSION glass / wavelengths: without a glass material in the scene bounces change nothing
    a.maxBounces = s->dev.nGlass ? (uint32_t)p->max_bounces : 0u;
    a.spectral = (uint32_t)p->spectral;
    if (a.spectral) ft::spectralTable((int)a.spectral, a.spec);
    a.ext = (a.spp != 1u || a.aoSamples != 0u || a.maxBounces != 0u || a.spectral != 0u) ? 1u : 0u;
    if (a.spp == 1) { a.out = static_cast<float*>(d_out); return launchTrace(c, s, a, lane); }
    // EXTENSION: one frame per sample, then a fixed-order resolve
    const size_t planeFloats = (size_t)a.planePixels * 3;
    const size_t need = planeFloats * a.spp * sizeof(float);
    if (need > c->planesBytes) {
        if (c->planes) { HIP_TRY(hipFree(c->planes)); c->planes = nullptr; c->planesBytes = 0; }
        HIP_TRY(hipMalloc(&c->planes, need));
        c->planesBytes = need;
    }
    a.out = static_cast<float*>(c->planes);
    if ((rc = launchTrace(c, s, a))) return rc;
    HIP_TRY(ft_launch_resolve(static_cast<const float*>(c->planes), static_cast<float*>(d_out), planeFloats, a.spp, c->stream));
    return FT_OK;
}

int ft_render_device(ft_ctx* c, const ft_scene* s, const ft_camera* cam, const ft_render_params* p, void* d_out) {
    return renderLane(c, s, cam, p, d_out, 0);
}

int ft_collect_stats(ft_ctx* c, ft_stats* st) {
    int rc = requireDevice(c); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double ms = c->foldedMs;
    c->foldedMs = 0.0;
    for (auto& p : c->events) {
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, p.first, p.second));
        ms += t;
        c->eventPool.push_back(p);
    }
    c->events.clear();
    // Read and reset ON THE CONTEXT'S STREAM.  The stream is hipStreamNonBlocking, i.e. it does not synchronise with the legacy
    // null stream, and hipMemset of device memory returns before the fill has run: round 2 reset the block with a null-stream
    // hipMemset, which could still be pending when the next launch on c->stream began adding to it.
    FtStatsDev h{};
    HIP_TRY(hipMemcpyAsync(&h, c->dStats, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(c->dStats, 0, sizeof(FtStatsDev), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (st) {
        st->rays_primary = h.rays_primary; st->rays_shadow = h.rays_shadow; st->rays_ext = h.rays_ext;
        st->hits_primary = h.hits_primary; st->hits_shadow = h.hits_shadow; st->sdf_evals = h.sdf_evals;
        st->flags = h.flags; st->kernel_ms = (float)ms; st->wave_evals = h.wave_evals;
        st->shader_mhz = h.clk_ref ? (float)((double)h.clk_shader / (double)h.clk_ref * 100.0) : 0.0f;   // s_memrealtime: 100 MHz
        st->tail_fraction = h.sdf_evals ? (float)((double)h.coop_evals / (double)h.sdf_evals) : 0.0f;
        st->culled_fraction = h.cull_total ? (float)((double)h.cull_skipped / (double)h.cull_total) : 0.0f;
    }
    return FT_OK;
}

// ---- host output: Image.render returns a host FColor[,] (Image.fs:26-35) ------------------------------------------
// The frame (12 B / pixel) has to cross PCIe.  A pageable destination is copied by the runtime through its own staging
// at 4-8 GB/s (23-50 ms for a 4096^2 frame); a page-locked one is written by the DMA engines at link rate.  ft_render
// therefore (a) page-locks the caller's buffer for the duration of the call unless it already is (ft_host_register, or
// memory from hipHostMalloc) — the pinning runs on the calling thread while the GPU already renders — and (b) renders
// a large contiguous frame in four column chunks on two alternating streams (the drain of one chunk overlaps the start of
// the next) while a third stream copies every finished chunk, so that only the last chunk's copy is left at the end.
int ft_host_register(ft_ctx* c, void* p, uint64_t bytes) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!p || bytes == 0) return setErr(FT_ERR_INVALID, "bad argument");
    HIP_TRY(hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault));
    c->hostRegs.emplace_back(p, (size_t)bytes);
    return FT_OK;
}

int ft_host_unregister(ft_ctx* c, void* p) {
    int rc = requireDevice(c); if (rc) return rc;
    for (size_t i = 0; i < c->hostRegs.size(); ++i)
        if (c->hostRegs[i].first == p) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipHostUnregister(p));
            c->hostRegs.erase(c->hostRegs.begin() + (long)i);
            return FT_OK;
        }
    return setErr(FT_ERR_INVALID, "this pointer was not registered through ft_host_register");
}

namespace {

bool isPageLocked(ft_ctx* c, const void* p, size_t bytes) {
    const char* b = static_cast<const char*>(p);
    for (auto& r : c->hostRegs)
        if (b >= static_cast<const char*>(r.first) && b + bytes <= static_cast<const char*>(r.first) + r.second) return true;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc memory: "invalid value"
    return attr.type == hipMemoryTypeHost;
}

int ensurePipeline(ft_ctx* c, size_t nEvents) {
    if (!c->lane1) HIP_TRY(hipStreamCreateWithFlags(&c->lane1, hipStreamNonBlocking));
    if (!c->copyStream) HIP_TRY(hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
    while (c->syncEvents.size() < nEvents) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->syncEvents.push_back(e);
    }
    return FT_OK;
}

}  // namespace

namespace {

// The frame of `p` rendered into the context's scratch buffer in column chunks on the two render lanes (the drain of one chunk
// overlaps the start of the next); chunk i's completion is syncEvents[2 + i].  c0 receives the chunk boundaries (columns).
struct ChunkPlan { int n = 1; std::vector<int> c0; size_t colBytes = 0, bytes = 0; };

int launchChunks(ft_ctx* c, const ft_scene* s, const ft_camera* cam, const ft_render_params* p, ChunkPlan& plan) {
    int rc;
    plan.colBytes = (size_t)p->height * 3 * sizeof(float);
    plan.bytes = (size_t)p->n_columns * plan.colBytes;
    if ((rc = ensureScratch(c, plan.bytes))) return rc;
    // chunks: only the reference's sampling (spp = 1: no shared sample planes) of a contiguous column range that is worth it
    plan.n = (p->spp == 1 && p->stripe_ranks == 1 && p->n_columns >= 256 && plan.bytes >= ((size_t)16 << 20)) ? 4 : 1;
    if (const int v = c->optHostChunks; v >= 1 && p->spp == 1 && p->stripe_ranks == 1 && p->n_columns >= 8 * v) plan.n = v;   // FT_OPT_HOST_CHUNKS
    if ((rc = ensurePipeline(c, (size_t)plan.n + 2))) return rc;
    HIP_TRY(hipEventRecord(c->syncEvents[0], c->stream));      // whatever the caller queued on the context's stream comes first
    HIP_TRY(hipStreamWaitEvent(c->lane1, c->syncEvents[0], 0));
    HIP_TRY(hipStreamWaitEvent(c->copyStream, c->syncEvents[0], 0));
    plan.c0.assign(plan.n + 1, 0);
    for (int i = 0; i < plan.n; ++i) plan.c0[i] = (int)(((int64_t)p->n_columns * i / plan.n) & ~(int64_t)7);   // chunks start on a tile boundary
    plan.c0[plan.n] = p->n_columns;
    char* dFrame = static_cast<char*>(c->scratch);
    for (int i = 0; i < plan.n; ++i) {
        ft_render_params q = *p;
        q.x0 = p->x0 + plan.c0[i]; q.n_columns = plan.c0[i + 1] - plan.c0[i];
        if (p->stripe_ranks == 1) q.stripe_width = q.n_columns;
        const int lane = plan.n > 1 ? (i & 1) : 0;
        if ((rc = renderLane(c, s, cam, &q, dFrame + (size_t)plan.c0[i] * plan.colBytes, lane))) return rc;
        HIP_TRY(hipEventRecord(c->syncEvents[2 + i], lane ? c->lane1 : c->stream));
    }
    return FT_OK;
}

void drainPipeline(ft_ctx* c) {
    (void)hipStreamSynchronize(c->copyStream); (void)hipStreamSynchronize(c->lane1); (void)hipStreamSynchronize(c->stream);
}

// page-lock `p` unless it already is; true if this call pinned it (the caller unpins)
bool pinForCall(ft_ctx* c, void* p, size_t bytes) {
    if (bytes < ((size_t)1 << 20) || !c->optHostPin || isPageLocked(c, p, bytes)) return false;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) return true;
    (void)hipGetLastError();                                   // not fatal: the runtime's pageable path still works
    return false;
}

}  // namespace

int ft_render(ft_ctx* c, const ft_scene* s, const ft_camera* cam, const ft_render_params* p, float* out, ft_stats* st) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!out) return setErr(FT_ERR_INVALID, "null output");
    if ((rc = checkParams(p))) return rc;
    ChunkPlan plan;
    bool pinnedHere = false;
    // whatever goes wrong, nothing of this call may still be in flight when it returns (the scratch frame and the caller's
    // buffer are reused), and a buffer pinned here is released
    auto finish = [&](int code) { if (c->copyStream) drainPipeline(c); if (pinnedHere) (void)hipHostUnregister(out); return code; };
    if ((rc = launchChunks(c, s, cam, p, plan))) return finish(rc);
    pinnedHere = pinForCall(c, out, plan.bytes);               // the GPU is rendering: page-lock the destination meanwhile
    char* dFrame = static_cast<char*>(c->scratch);
    hipError_t err = hipSuccess;
    for (int i = 0; i < plan.n && err == hipSuccess; ++i) {
        const size_t off = (size_t)plan.c0[i] * plan.colBytes, n = (size_t)(plan.c0[i + 1] - plan.c0[i]) * plan.colBytes;
        if ((err = hipStreamWaitEvent(c->copyStream, c->syncEvents[2 + i], 0)) != hipSuccess) break;
        err = hipMemcpyAsync(reinterpret_cast<char*>(out) + off, dFrame + off, n, hipMemcpyDeviceToHost, c->copyStream);
    }
    if (err == hipSuccess) err = hipEventRecord(c->syncEvents[1], c->copyStream);
    if (err == hipSuccess) err = hipStreamWaitEvent(c->stream, c->syncEvents[1], 0);   // join: the context's stream ends after the copies
    if (err == hipSuccess) err = hipStreamSynchronize(c->copyStream);
    if (err == hipSuccess) err = hipStreamSynchronize(c->lane1);
    if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
    if (err != hipSuccess) return finish(hipFail(err, "ft_render host output"));
    if (pinnedHere) { (void)hipHostUnregister(out); pinnedHere = false; }
    return ft_collect_stats(c, st);
}

// ---- tone map: Image.toColors / toBitmap order on the device ---------------------------------
static int checkToneMap(const void* frame, int32_t X, int32_t Y, const ft_tonemap_params* p) {
    if (!frame || !p || X <= 0 || Y <= 0) return setErr(FT_ERR_INVALID, "bad argument");
    if (!(p->gamma > 0.0f)) return setErr(FT_ERR_INVALID, "gamma must be positive");
    if ((uint64_t)X * (uint64_t)Y >= (1ull << 31)) return setErr(FT_ERR_UNSUPPORTED, "more than 2^31 pixels");
    return FT_OK;
}

int ft_tone_map_device(ft_ctx* c, const void* d_frame, int32_t X, int32_t Y, const ft_tonemap_params* p, void* d_out) {
    int rc = requireDevice(c); if (rc) return rc;
    if ((rc = checkToneMap(d_frame, X, Y, p))) return rc;
    if (!d_out) return setErr(FT_ERR_INVALID, "null output");
    if (reinterpret_cast<uintptr_t>(d_frame) & 15u) return setErr(FT_ERR_INVALID, "the device frame must be 16-byte aligned");
    if ((rc = ensureAux(c, 256))) return rc;
    const float gammaInv = 1.0f / p->gamma;                            // Image.fs:38
    HIP_TRY(ft_launch_tonemap(static_cast<const float*>(d_frame), (uint32_t)X, (uint32_t)Y, static_cast<uint32_t*>(c->aux), gammaInv,
                              p->dither ? 1u : 0u, p->seed, p->bmp_order != 0, static_cast<unsigned char*>(d_out), (unsigned)c->numCUs, c->optMath, c->stream));
    return FT_OK;
}

int ft_tone_map(ft_ctx* c, const void* d_frame, int32_t X, int32_t Y, const ft_tonemap_params* p, uint8_t* out, float* max_out) {
    int rc = requireDevice(c); if (rc) return rc;
    if ((rc = checkToneMap(d_frame, X, Y, p))) return rc;
    if (!out) return setErr(FT_ERR_INVALID, "null output");
    const size_t bytes = (size_t)X * Y * 3;
    if ((rc = ensureAux(c, 256 + bytes))) return rc;
    unsigned char* dBytes = static_cast<unsigned char*>(c->aux) + 256;
    if ((rc = ft_tone_map_device(c, d_frame, X, Y, p, dBytes))) return rc;
    HIP_TRY(hipMemcpyAsync(out, dBytes, bytes, hipMemcpyDeviceToHost, c->stream));
    if (max_out) HIP_TRY(hipMemcpyAsync(max_out, c->aux, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FT_OK;
}

int ft_tone_map_host(ft_ctx* c, const float* frame, int32_t X, int32_t Y, const ft_tonemap_params* p, uint8_t* out, float* max_out) {
    int rc = requireDevice(c); if (rc) return rc;
    if ((rc = checkToneMap(frame, X, Y, p))) return rc;
    const size_t bytes = (size_t)X * Y * 12;
    if ((rc = ensureScratch(c, bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(c->scratch, frame, bytes, hipMemcpyHostToDevice, c->stream));
    return ft_tone_map(c, c->scratch, X, Y, p, out, max_out);
}

int ft_render_colors(ft_ctx* c, const ft_scene* s, const ft_camera* cam, const ft_render_params* p, const ft_tonemap_params* tm,
                     uint8_t* out, float* max_out, ft_stats* st) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!out || !tm) return setErr(FT_ERR_INVALID, "null argument");
    if ((rc = checkParams(p))) return rc;
    if (p->x0 != 0 || p->n_columns != p->width || p->stripe_ranks != 1)
        return setErr(FT_ERR_INVALID, "the tone map needs the whole frame (its normalisation is the global maximum, Image.fs:40-43)");
    if ((rc = checkToneMap(out, p->width, p->height, tm))) return rc;
    // the frame is rendered like ft_render's (column chunks on two lanes); the tone map follows on the context's stream once both
    // lanes are done — its normalisation needs every pixel — and only the bytes are copied out
    ChunkPlan plan;
    const size_t outBytes = (size_t)p->width * p->height * 3;
    bool pinnedHere = false;
    auto finish = [&](int code) { if (c->copyStream) drainPipeline(c); if (pinnedHere) (void)hipHostUnregister(out); return code; };
    if ((rc = launchChunks(c, s, cam, p, plan))) return finish(rc);
    pinnedHere = pinForCall(c, out, outBytes);
    hipError_t err = hipSuccess;
    for (int i = 0; i < plan.n && err == hipSuccess; ++i) err = hipStreamWaitEvent(c->stream, c->syncEvents[2 + i], 0);
    if (err != hipSuccess) return finish(hipFail(err, "ft_render_colors"));
    if ((rc = ft_tone_map(c, c->scratch, p->width, p->height, tm, out, max_out))) return finish(rc);
    if (pinnedHere) { (void)hipHostUnregister(out); pinnedHere = false; }
    return ft_collect_stats(c, st);
}

// mode 1: SdfScene.trace (3 floats per ray); 2: SdfForm.tryTrace (10 dwords); 3: SdfObject.tryTrace (16 dwords)
static int traceRayBuffer(ft_ctx* c, const ft_scene* s, const ft_ray* rays, int64_t n, void* out, uint32_t mode, ft_stats* st) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!s || s->ctx != c || !rays || !out || n < 0) return setErr(FT_ERR_INVALID, "bad argument");
    if (n == 0) { if (st) memset(st, 0, sizeof(*st)); return FT_OK; }
    if (n >= 0xFFFF0000ll) return setErr(FT_ERR_UNSUPPORTED, "more than 2^32 rays in one call");
    const size_t perRay = mode == 1 ? 3 : mode == 2 ? 10 : 16;
    const size_t rayBytes = align256((size_t)n * sizeof(ft_ray)), outBytes = (size_t)n * perRay * sizeof(float);
    if ((rc = ensureScratch(c, rayBytes + outBytes))) return rc;
    unsigned char* base = static_cast<unsigned char*>(c->scratch);
    HIP_TRY(hipMemcpyAsync(base, rays, (size_t)n * sizeof(ft_ray), hipMemcpyHostToDevice, c->stream));
    FtRenderArgs a{};
    a.mode = mode; a.rays = reinterpret_cast<const ft_ray*>(base); a.out = reinterpret_cast<float*>(base + rayBytes);
    a.nJobs = (uint32_t)n; a.stripeW = 1; a.stripeRanks = 1; a.tilesY = 1; a.H = 1; a.W = 1; a.nCols = 1; a.maxSize = 1.0f;
    a.spp = 1; a.sppN = 1; a.jobsPerPlane = a.nJobs; a.planePixels = a.nJobs;
    a.ext = mode >= 2 ? 1u : 0u;                           // the tryTrace outputs exist in the EXTENSION builds of the kernel only
    if ((rc = launchTrace(c, s, a))) return rc;
    HIP_TRY(hipMemcpyAsync(out, base + rayBytes, outBytes, hipMemcpyDeviceToHost, c->stream));
    return ft_collect_stats(c, st);
}

int ft_trace_rays(ft_ctx* c, const ft_scene* s, const ft_ray* rays, int64_t n, float* out, ft_stats* st) {
    return traceRayBuffer(c, s, rays, n, out, 1, st);
}
int ft_form_try_trace(ft_ctx* c, const ft_scene* s, const ft_ray* rays, int64_t n, ft_form_trace_result* out, ft_stats* st) {
    static_assert(sizeof(ft_form_trace_result) == 40, "layout");
    return traceRayBuffer(c, s, rays, n, out, 2, st);
}
int ft_object_try_trace(ft_ctx* c, const ft_scene* s, const ft_ray* rays, int64_t n, ft_object_trace_result* out, ft_stats* st) {
    static_assert(sizeof(ft_object_trace_result) == 64, "layout");
    return traceRayBuffer(c, s, rays, n, out, 3, st);
}

int ft_eval_distance(ft_ctx* c, const ft_scene* s, const ft_vec3* pts, int64_t n, float* outD, int32_t* outM) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!s || s->ctx != c || !pts || !outD || n < 0) return setErr(FT_ERR_INVALID, "bad argument");
    if (n == 0) return FT_OK;
    const size_t pBytes = align256((size_t)n * 12), dBytes = align256((size_t)n * 4), mBytes = (size_t)n * 4;
    if ((rc = ensureScratch(c, pBytes + dBytes + mBytes))) return rc;
    unsigned char* base = static_cast<unsigned char*>(c->scratch);
    HIP_TRY(hipMemcpyAsync(base, pts, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
    const unsigned blocks = (unsigned)std::min<int64_t>((n + FT_BLOCK - 1) / FT_BLOCK, (int64_t)c->numCUs * 8);
    const bool libm = libmLaunch(c, s);
    FtSceneDev dev = s->dev;
    dev.mathFma = c->optMath == FT_MATH_GLIBC_FMA ? 1u : 0u;
    HIP_TRY(ft_launch_eval_points(&dev, libm ? 1 : 0, reinterpret_cast<const float*>(base), n, reinterpret_cast<float*>(base + pBytes),
                                  reinterpret_cast<int*>(base + pBytes + dBytes), blocks, ldsBytes(s, libm), c->stream));
    HIP_TRY(hipMemcpyAsync(outD, base + pBytes, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (outM) HIP_TRY(hipMemcpyAsync(outM, base + pBytes + dBytes, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FT_OK;
}

int ft_math_eval(ft_ctx* c, int32_t op, const float* x, const float* y, int64_t n, float* out) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!x || !out || n < 0 || op < 0 || op > 14 || ((op == 3 || op >= 10) && !y)) return setErr(FT_ERR_INVALID, "bad argument");
    if (n == 0) return FT_OK;
    const size_t b = align256((size_t)n * 4);
    if ((rc = ensureScratch(c, 3 * b))) return rc;
    unsigned char* base = static_cast<unsigned char*>(c->scratch);
    HIP_TRY(hipMemcpyAsync(base, x, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (y) HIP_TRY(hipMemcpyAsync(base + b, y, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(ft_launch_math(op, reinterpret_cast<const float*>(base), reinterpret_cast<const float*>(base + b), n,
                           reinterpret_cast<float*>(base + 2 * b), c->stream));
    HIP_TRY(hipMemcpyAsync(out, base + 2 * b, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FT_OK;
}

int ft_selftest_fastmath(ft_ctx* c, uint64_t mismatches[3]) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!mismatches) return setErr(FT_ERR_INVALID, "null output");
    if ((rc = ensureScratch(c, 256))) return rc;
    unsigned long long* d = static_cast<unsigned long long*>(c->scratch);
    HIP_TRY(hipMemsetAsync(d, 0, 24, c->stream));
    // sqrt: every float in [2^-96, 2^100]; exp: every float in [-2.9e6, -0] and [+0, 88]
    HIP_TRY(ft_launch_selftest(0, 0x0F800000u, 0x71800000u, d, c->stream));
    HIP_TRY(ft_launch_selftest(5, 0x27800000u, 0x48000000u, d, c->stream));    // strength -2 / -4 / -1/2 through the subtraction's output modifier: every root in [2^-48, 2^17] x four radii
    HIP_TRY(ft_launch_selftest(3, 0x0F800000u, 0x71800000u, d, c->stream));    // the 4-instruction form (output modifiers) under the NEAR loop's mode: same range, same counter
    HIP_TRY(ft_launch_selftest(1, 0x80000000u, 0xCA310080u, d + 1, c->stream));
    HIP_TRY(ft_launch_selftest(1, 0x00000000u, 0x42B00000u, d + 1, c->stream));
    // exponent-add form of exp ("near" regime): every float in [-87, -0] and [+0, 88]
    HIP_TRY(ft_launch_selftest(2, 0x80000000u, 0xC2AE0000u, d + 2, c->stream));
    HIP_TRY(ft_launch_selftest(2, 0x00000000u, 0x42B00000u, d + 2, c->stream));
    HIP_TRY(ft_launch_selftest(4, 0x80000000u, 0xC2AE0000u, d + 2, c->stream));  // the same form under the NEAR loop's mode (IEEE off, f32 denormals flushed)
    HIP_TRY(ft_launch_selftest(4, 0x00000000u, 0x42B00000u, d + 2, c->stream));
    unsigned long long h[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    mismatches[0] = h[0]; mismatches[1] = h[1]; mismatches[2] = h[2];
    return FT_OK;
}

int ft_selftest_libm(ft_ctx* c, int32_t op, int32_t variant, float y, uint32_t lo_bits, int32_t n_chunks, uint64_t* sums) {
    int rc = requireDevice(c); if (rc) return rc;
    if (!sums || op < 0 || op > 2 || (variant != FT_MATH_GLIBC_FMA && variant != FT_MATH_GLIBC_SSE2) || n_chunks < 1 || n_chunks > 256 ||
        (uint64_t)lo_bits + ((uint64_t)n_chunks << 24) > (1ull << 32)) return setErr(FT_ERR_INVALID, "bad argument");
    if ((rc = ensureScratch(c, (size_t)n_chunks * 8))) return rc;
    unsigned long long* d = static_cast<unsigned long long*>(c->scratch);
    HIP_TRY(hipMemsetAsync(d, 0, (size_t)n_chunks * 8, c->stream));
    HIP_TRY(ft_launch_libm_checksum(op, variant, y, lo_bits, (uint32_t)n_chunks, d, c->stream));
    HIP_TRY(hipMemcpyAsync(sums, d, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FT_OK;
}

// ---- internal hooks for multi.cpp (not part of the public ABI) ------------------------------------------
int ft_ctx_device_(const ft_ctx* c) { return (c && c->hasDevice) ? c->device : -1; }
void* ft_ctx_stream_(const ft_ctx* c) { return c ? (void*)c->stream : nullptr; }
void ft_set_error_(int, const char* msg) { g_err = msg ? msg : ""; }

// ---- introspection ---------------------------------------------------------------------------------
int ft_scene_info_get(const ft_scene* s, ft_scene_info* o) {
    if (!s || !o) return setErr(FT_ERR_INVALID, "null argument");
    const ft::FlatScene& f = s->flat;
    o->n_instr = (int32_t)f.instr.size(); o->n_slots = (int32_t)f.nSlots; o->n_consts = (int32_t)f.consts.size();
    o->n_grids = (int32_t)f.grids.size(); o->n_children = (int32_t)f.children.size(); o->n_cells = (int32_t)(f.cellCenters.size() / 3);
    o->n_items = (int32_t)f.items.size(); o->n_lights = (int32_t)f.lights.size(); o->n_materials = (int32_t)(f.materials.size() / 3);
    o->fast_path = (int32_t)f.fastPath;
    o->cull_pc = (int32_t)f.cullPc;
    return FT_OK;
}

int ft_scene_grid_shape(const ft_scene* s, int32_t g, float info[6], int32_t counts[3], int32_t* nCells, int32_t* nItems) {
    if (!s || g < 0 || (size_t)g >= s->flat.grids.size()) return setErr(FT_ERR_INVALID, "bad grid index");
    if (!info || !counts) return setErr(FT_ERR_INVALID, "null output");
    const FtGrid& G = s->flat.grids[g];
    for (int i = 0; i < 3; ++i) { info[i] = G.aabbMin[i]; info[3 + i] = G.cellSizeInv[i]; counts[i] = G.count[i]; }
    const int32_t nc = G.count[0] * G.count[1] * G.count[2];
    if (nCells) *nCells = nc;
    if (nItems) *nItems = (int32_t)(s->flat.cellStart[G.cellBase + nc] - s->flat.cellStart[G.cellBase]);
    return FT_OK;
}

int ft_scene_support_sphere(const ft_scene* s, float cr[4]) {
    if (!s || !cr) return setErr(FT_ERR_INVALID, "null argument");
    cr[0] = s->flat.escC[0]; cr[1] = s->flat.escC[1]; cr[2] = s->flat.escC[2]; cr[3] = s->flat.escR;
    return FT_OK;
}

int ft_scene_grid_dump(const ft_scene* s, int32_t g, uint32_t* cellStart, float* centers, float* lower, int32_t* child) {
    if (!s || g < 0 || (size_t)g >= s->flat.grids.size()) return setErr(FT_ERR_INVALID, "bad grid index");
    if (!cellStart || !centers || !lower || !child) return setErr(FT_ERR_INVALID, "null output");
    const ft::FlatScene& f = s->flat;
    const FtGrid& G = f.grids[g];
    const uint32_t nc = (uint32_t)(G.count[0] * G.count[1] * G.count[2]);
    const uint32_t first = f.cellStart[G.cellBase];
    for (uint32_t c = 0; c <= nc; ++c) cellStart[c] = f.cellStart[G.cellBase + c] - first;
    memcpy(centers, f.cellCenters.data() + 3 * (size_t)G.cellBase, (size_t)nc * 12);
    const uint32_t ni = f.cellStart[G.cellBase + nc] - first;
    for (uint32_t i = 0; i < ni; ++i) { lower[i] = f.items[first + i].lowerBound; child[i] = (int32_t)f.items[first + i].child; }
    return FT_OK;
}

}  // extern "C"
