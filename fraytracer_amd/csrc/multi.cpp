// multi.cpp — single-process multi-GPU render: column stripes per device + ONE ncclGather (RCCL
// over xGMI) of the per-device FColor slabs to the first device.
//
// Pixels are independent (immutable scene, Image.fs:28-35), so the only communication is the
// collection of the finished slabs.  Columns are dealt in stripes of `stripe_width` so that the
// expensive centre of the image is spread over all devices; because the output is column-major
// (Array2D.fs:30-38) every stripe is one contiguous run of stripe_width*height*3 floats in both
// the gathered buffer and the final image: one strided device copy (ft_deinterleave_kernel) puts the gathered slabs
// in frame order on the first device and ONE device->host copy delivers the frame.
//
// Threading and RCCL: the communicators come from ncclCommInitAll (one per device, cached), and every device has its own
// host thread, which renders on its context's stream and then calls ncclGather on ITS communicator and stream.  That is
// RCCL's "one thread per device" mode, in which the per-rank calls of one collective are issued concurrently by
// construction; ncclGroupStart / ncclGroupEnd are for the other mode — one thread issuing the calls of several
// communicators in a loop, where the first call would otherwise block waiting for its peers — and are not used here.
// Slabs, the gather buffer and the frame buffer are cached between calls (same devices, same size).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>      // types and prototypes only: the library itself is loaded on first use

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fraytracer_hip.h"
#include "ft_kernels.h"

// small internal hooks exported by capi.cpp
extern "C" {
int ft_ctx_device_(const ft_ctx*);
void* ft_ctx_stream_(const ft_ctx*);
void ft_set_error_(int code, const char* msg);
}

namespace {

struct CommCache {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};
std::mutex g_mu;
CommCache g_cache;

// device buffers of ft_render_multi, kept between calls: one slab per device, and on the first device the gather
// buffer [n x slab] and the de-interleaved frame [n x slab]
struct BufCache {
    std::vector<int> devices;
    size_t slabFloats = 0;
    std::vector<float*> send;
    float* recv = nullptr;
    float* frame = nullptr;
    void release() {
        for (size_t r = 0; r < send.size(); ++r) if (send[r]) { (void)hipSetDevice(devices[r]); (void)hipFree(send[r]); }
        if (!devices.empty()) (void)hipSetDevice(devices[0]);
        if (recv) (void)hipFree(recv);
        if (frame) (void)hipFree(frame);
        *this = BufCache{};
    }
    bool ensure(const std::vector<int>& devs, size_t slab) {
        if (devices == devs && slabFloats == slab) return true;
        release();
        devices = devs; slabFloats = slab; send.assign(devs.size(), nullptr);
        for (size_t r = 0; r < devs.size(); ++r)
            if (hipSetDevice(devs[r]) != hipSuccess || hipMalloc((void**)&send[r], slab * sizeof(float)) != hipSuccess) { release(); return false; }
        if (devs.size() > 1) {
            if (hipSetDevice(devs[0]) != hipSuccess || hipMalloc((void**)&recv, slab * devs.size() * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&frame, slab * devs.size() * sizeof(float)) != hipSuccess) { release(); return false; }
        }
        return true;
    }
};
BufCache g_bufs;

int fail(int code, const std::string& m) { ft_set_error_(code, m.c_str()); return code; }

// RCCL is bound lazily (dlopen) so that single-GPU hosts never load it and a process that already has
// an RCCL (e.g. PyTorch's bundled copy) keeps exactly one: an already-loaded librccl.so.1 is reused.
struct Rccl {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl g_rccl;

bool loadRccl(std::string& err) {
    if (g_rccl.ok) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.Gather = reinterpret_cast<decltype(g_rccl.Gather)>(dlsym(h, "ncclGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.Gather || !g_rccl.GetErrorString) { err = "RCCL lacks ncclGather / ncclCommInitAll"; return false; }
    g_rccl.ok = true;
    return true;
}

bool getComms(const std::vector<int>& devs, std::vector<ncclComm_t>& out, std::string& err) {
    if (!loadRccl(err)) return false;
    if (g_cache.devices == devs) { out = g_cache.comms; return true; }
    for (ncclComm_t c : g_cache.comms) g_rccl.CommDestroy(c);
    g_cache = CommCache{};
    std::vector<ncclComm_t> comms(devs.size());
    ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
    // a failed initialisation leaves nothing cached (g_cache was cleared above), so a later call starts from scratch
    if (r != ncclSuccess) { err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r); return false; }
    g_cache.devices = devs; g_cache.comms = comms;
    out = comms;
    return true;
}

}  // namespace

extern "C" int ft_render_multi(ft_ctx* const* ctxs, const ft_scene* const* scenes, int32_t n, const ft_camera* cam,
                               const ft_render_params* full, float* out, ft_stats* stats) {
    if (!ctxs || !scenes || n <= 0 || !cam || !full || !out) return fail(FT_ERR_INVALID, "bad argument");
    const int W = full->width, H = full->height;
    if (W <= 0 || H <= 0) return fail(FT_ERR_INVALID, "empty image");
    if (full->stripe_width < 0) return fail(FT_ERR_INVALID, "negative stripe_width");
    const int S = full->stripe_width > 0 ? full->stripe_width : W / n;
    if (S <= 0 || W % (S * n) != 0) return fail(FT_ERR_UNSUPPORTED, "width must be a multiple of stripe_width * n_devices");
    const int cols = W / n;                                   // columns per device
    const size_t slab = (size_t)cols * H * 3;                 // floats per device

    std::vector<int> devs(n);
    for (int r = 0; r < n; ++r) if (!ctxs[r] || !scenes[r]) return fail(FT_ERR_INVALID, "null context / scene");
    for (int r = 0; r < n; ++r) {
        devs[r] = ft_ctx_device_(ctxs[r]);
        if (devs[r] < 0) return fail(FT_ERR_NO_DEVICE, "context has no GPU: libfraytracer_hip has no CPU fallback");
    }
    std::lock_guard<std::mutex> lock(g_mu);
    std::vector<ncclComm_t> comms;
    std::string err;
    // several contexts on ONE device (rehearsal on a single-GPU machine; RCCL refuses duplicate GPUs): the
    // slabs are collected with device-to-device copies instead of the gather; everything else is the same path
    bool sameDevice = n > 1;
    for (int r = 1; r < n; ++r) sameDevice = sameDevice && devs[r] == devs[0];
    if (n > 1 && !sameDevice && !getComms(devs, comms, err)) return fail(FT_ERR_COMM, err);

    if (!g_bufs.ensure(devs, slab)) return fail(FT_ERR_HIP, "hipMalloc of the stripe slabs / gather buffers failed");
    const std::vector<float*>& send = g_bufs.send;
    float* recv = g_bufs.recv;
    std::vector<int> rcs(n, FT_OK);
    std::vector<std::string> errs(n);
    std::vector<ft_stats> sts(n);

    auto worker = [&](int r) {
        // RCCL's one-thread-per-device mode wants the calling thread ON the communicator's device; the HIP device is per thread
        if (hipSetDevice(devs[r]) != hipSuccess) { rcs[r] = FT_ERR_HIP; errs[r] = "hipSetDevice failed in a device thread"; return; }
        ft_render_params p = *full;
        p.x0 = 0; p.n_columns = cols; p.stripe_width = S; p.stripe_ranks = n; p.stripe_rank = r;
        int rc = ft_render_device(ctxs[r], scenes[r], cam, &p, send[r]);
        if (rc == FT_OK && n > 1 && sameDevice) {
            if (hipMemcpyAsync(recv + (size_t)r * slab, send[r], slab * sizeof(float), hipMemcpyDeviceToDevice,
                               (hipStream_t)ft_ctx_stream_(ctxs[r])) != hipSuccess) { rc = FT_ERR_HIP; errs[r] = "device-to-device slab copy failed"; }
        } else if (rc == FT_OK && n > 1) {
            ncclResult_t nr = g_rccl.Gather(send[r], recv, slab, ncclFloat, 0, comms[r], (hipStream_t)ft_ctx_stream_(ctxs[r]));
            if (nr != ncclSuccess) { rc = FT_ERR_COMM; errs[r] = std::string("ncclGather: ") + g_rccl.GetErrorString(nr); }
        } else if (rc != FT_OK) errs[r] = ft_last_error();
        if (rc == FT_OK) { rc = ft_collect_stats(ctxs[r], &sts[r]); if (rc) errs[r] = ft_last_error(); }
        rcs[r] = rc;
    };
    if (n == 1) worker(0);
    else {
        std::vector<std::thread> ts;
        for (int r = 0; r < n; ++r) ts.emplace_back(worker, r);
        for (auto& t : ts) t.join();
    }
    for (int r = 0; r < n; ++r) if (rcs[r] != FT_OK) return fail(rcs[r], errs[r]);

    // stripe j of device r is columns [(j*n + r)*S, +S) of the frame: one strided copy on the first device, one copy out
    // (every worker has synchronised its stream in ft_collect_stats, so the gathered data is complete)
    (void)hipSetDevice(devs[0]);
    const float* src = send[0];
    hipError_t he = hipSuccess;
    if (n > 1) {
        hipStream_t st0 = (hipStream_t)ft_ctx_stream_(ctxs[0]);
        he = ft_launch_deinterleave(recv, g_bufs.frame, (unsigned long long)S * H * 3, (uint32_t)(cols / S), (uint32_t)n, st0);
        if (he == hipSuccess) he = hipStreamSynchronize(st0);
        src = g_bufs.frame;
    }
    if (he == hipSuccess) he = hipMemcpy(out, src, slab * n * sizeof(float), hipMemcpyDeviceToHost);
    if (he != hipSuccess) return fail(FT_ERR_HIP, std::string("de-interleave / copy of the gathered image: ") + hipGetErrorString(he));

    if (stats) {
        *stats = ft_stats{};
        for (int r = 0; r < n; ++r) {
            stats->rays_primary += sts[r].rays_primary; stats->rays_shadow += sts[r].rays_shadow; stats->rays_ext += sts[r].rays_ext;
            stats->hits_primary += sts[r].hits_primary; stats->hits_shadow += sts[r].hits_shadow; stats->sdf_evals += sts[r].sdf_evals;
            stats->flags |= sts[r].flags; stats->wave_evals += sts[r].wave_evals;
            if (sts[r].kernel_ms > stats->kernel_ms) stats->kernel_ms = sts[r].kernel_ms;   // devices run concurrently
            if (r == 0 || sts[r].shader_mhz < stats->shader_mhz) stats->shader_mhz = sts[r].shader_mhz;   // the slowest clock
        }
    }
    return FT_OK;
}
