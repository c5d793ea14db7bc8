#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + shadow rays) of the FrayTracer hot path on MI355X.

A "step" renders one synthetic frame of BASELINE.json configs[2]: the 256-sphere smooth-union scene
at 4096x4096 (SURVEY.md §8d "C3"; 1 sample per pixel = the reference's own sampling — spp > 1 is an
extension and is not what `value` is measured on).  With --gpus N the SAME frame is split into
interleaved column stripes over N ranks (one process per GPU, launched by torch.distributed.run),
each rank renders its stripes with the HIP kernel into HBM, and ONE RCCL gather per frame brings
the slabs to rank 0, which de-interleaves them: total work is fixed, so scaling is "strong".  Frames are
pipelined (fraytracer_amd.distributed.FramePipeline): the gather of frame k runs on a side stream while
frame k+1 renders; all K frames are complete on rank 0 when the timed region ends.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the trace kernel against the f32 VALU peak
(256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s; no FMA contraction is
allowed by the parity requirement).  `roofline.achieved / frac` are the HARDWARE fraction: the algorithmic lane-ops of the work the timed
launches EXECUTED (their own exact counters: evaluations after the escape shortcut, children after the culling pass) per second over
the peak; `valu_busy_pmc` is the counter view of the same thing.  `achieved_reference_work / frac_reference_work` price the REFERENCE's
work as SURVEY.md §8d defines it — every evaluation of every ray marched until its Length is used up, all children in each (counters of
one untimed launch with the escape shortcut off = the oracle's) — over the same time: an algorithmic speed-up figure, not a hardware
fraction (rounds 1-3 printed it under `frac`).  `config.arithmetic` names the arithmetic `value` was measured in: "fixed" (the default:
one fixed exp / log algorithm, same bits everywhere) or, with --math glibc, "glibc_fma" / "glibc_sse2" (MathF.Exp / Log as this
host's C runtime computes them — what the reference's CPU path returns under .NET on Linux).  `cpu_baseline` times the CPU oracle (a
C++ restatement of the F# CPU path — NOT the F# program) on a bounded sample of the same frame on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_PEAK_TLANEOPS = 78.6          # SURVEY.md §8d / MI355X_MICROARCH.md chip table
STRIPE = 16                        # columns per stripe when N > 1


def algorithmic_flops(stats, n_children, flops_per_child=13, flops_per_eval=3):
    """per SDF eval of the C3 scene 256 x (sphere 10 + smooth-union 3) + 3 = 3331;
    + 8 per march step, 20 per normal, 25 per shaded light.  Counts are the kernel's own exact
    counters (deterministic for a scene; equal to the oracle's for rays/hits)."""
    evals = stats["sdf_evals"]
    normals = stats["hits_primary"]
    steps = evals - 4 * normals
    return evals * (n_children * flops_per_child + flops_per_eval) + 8 * steps + 20 * normals + 25 * stats["rays_shadow"]


def union_algorithmic_flops(cnt):
    """SURVEY.md section 8d pricing of a grid-union scene from the oracle's counters (the reference's own work: it scans a
    cell's whole candidate list): primitives sphere 10 / capsule 21 / torus 27 / triangle 87 / box 20, union 13 per candidate
    + 30 per lookup (one per scene evaluation), intersect 12 + subtract 2 per evaluation of the Program.fs structure,
    8 per march step, 20 per normal, 25 per shadow ray."""
    prim = cnt["prim"]
    return (10 * prim[0] + 21 * prim[1] + 27 * prim[2] + 87 * prim[3] + 20 * prim[4] + 13 * cnt["union_candidates"]
            + (30 + 12 + 2) * cnt["root_evals"] + 8 * cnt["march_steps"] + 20 * cnt["hits_primary"] + 25 * cnt["rays_shadow"])


def profile_figures(build_src, kernel):
    """PMC figures of `kernel` from a COMMITTED rocprofv3 summary (separate --pmc passes of tools/profile.sh / profile_scene.sh) — but only
    from a summary that was captured on the very sources the running library was built from: tools/summarize_prof.py stamps every
    summary with ft_build_info()'s source hash, and a summary with another stamp (or none) is not quoted.  -> (traffic bytes per launch
    or None, file name or None, VALU-busy or None).  FETCH_SIZE is doubled as the MI355X guide prescribes for gfx950; both counters are in
    KiB.  VALU-busy = SQ_ACTIVE_INST_VALU x 2 issue cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.txt")))
    for f in reversed(files):
        txt = open(f).read()
        stamp = re.search(r"^# build src=([0-9a-f]+)", txt, re.M)
        if not stamp or stamp.group(1) != build_src or kernel not in txt:
            continue
        fe, wr = re.search(r"FETCH_SIZE\s+([0-9.e+]+)", txt), re.search(r"WRITE_SIZE\s+([0-9.e+]+)", txt)
        av, ga = re.search(r"SQ_ACTIVE_INST_VALU\s+([0-9.e+]+)", txt), re.search(r"GRBM_GUI_ACTIVE\s+([0-9.e+]+)", txt)
        busy = round(float(av.group(1)) * 2.0 / (1024.0 * float(ga.group(1)) / 8.0), 3) if av and ga else None
        traffic = int((2.0 * float(fe.group(1)) + float(wr.group(1))) * 1024) if fe and wr else None
        if busy is not None or traffic is not None:
            return traffic, os.path.basename(f), busy
    return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096, help="frame is size x size (default: the metric's 4096)")
    ap.add_argument("--spheres", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", "--no-spp4", dest="no_spp4", action="store_true",
                    help="skip the side measurements (4 spp, two lanes, host output, primary + AO target, Program.fs scene)")
    ap.add_argument("--no-reference-launch", action="store_true",
                    help="profiling passes (tools/profile.sh): skip the untimed launch that counts the reference's evaluations, so that every launch of the "
                         "trace kernel in the profile is a timed one; the roofline then prices the executed evaluation count")
    ap.add_argument("--no-c4", action="store_true", help="N > 1: skip the side block that renders BASELINE.json config 4 (the same scene at 8192x8192, sharded the same way)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: initialise RCCL and run the gather path in a 1-rank group")
    ap.add_argument("--math", choices=("fixed", "glibc"), default="fixed",
                    help="arithmetic of MathF.Exp / Log in the timed launches (FT_OPT_MATH): fixed = the library's one fixed algorithm (default), "
                         "glibc = this host's glibc expf / logf restated on the GPU — then `value`, `roofline` and the oracle check are all in that arithmetic")
    ap.add_argument("--cpu-columns", type=int, default=0,
                    help="columns of the frame the CPU oracle is timed on (default: 32 per usable host CPU, at least 32)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import fraytracer_amd as ft
    from fraytracer_amd import synthetic as syn
    from fraytracer_amd import distributed as ftd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs through torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: fraytracer_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner to stdout when its first communicator comes up: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    W = H = args.size
    if W % (STRIPE * world) != 0:
        raise SystemExit(f"size must be a multiple of {STRIPE * world}")
    scene, _ = syn.config3(n=args.spheres, size=W)
    cam = syn.default_camera()
    dev = ft.Device(local_rank)
    math_variant = ft.glibc_build_of_this_host() if args.math == "glibc" else 0       # FT_MATH_GLIBC_FMA = 1, _SSE2 = 2
    arithmetic = {0: "fixed", 1: "glibc_fma", 2: "glibc_sse2"}[math_variant]
    dev.set_option("math", math_variant)
    stream = torch.cuda.current_stream()
    dev.set_stream(stream.cuda_stream)            # kernel, its HIP events and the collective share one stream
    ds = dev.scene(scene)
    size = ft.ImageSize(W, H)
    cols = W // world
    tiling = ftd.tiling(W, world, rank, STRIPE)
    slab = torch.empty((cols, H, 3), dtype=torch.float32, device="cuda")

    def render(dst):
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, dst.data_ptr(), **tiling)

    # N > 1: frames are pipelined (fraytracer_amd.distributed.FramePipeline).  Two contexts on two streams render
    # alternate frames, so frame k+1 fills the GPU while the few long rays of frame k drain (the tail is 2 % of
    # a whole 4096^2 frame but 20 % of an eighth of it), and the slabs of frame k travel to rank 0 in the path's
    # ONE RCCL gather over xGMI and are de-interleaved there on a third stream.
    pipe, lanes = None, [ds]
    if use_dist:
        lane_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        dev.set_stream(lane_streams[0].cuda_stream)
        dev2 = ft.Device(local_rank)
        dev2.set_option("math", math_variant)
        dev2.set_stream(lane_streams[1].cuda_stream)
        lanes = [ds, dev2.scene(scene)]

        def lane(d):
            return lambda dst: d.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, dst.data_ptr(), **tiling)

        pipe = ftd.FramePipeline([lane(d) for d in lanes], cols, H, world, rank, STRIPE, torch.device("cuda", local_rank),
                                 streams=lane_streams, force=args.force_dist, timed=True)

    def collect():
        """exact counters + HIP-event kernel time since the last call, summed over the render lanes"""
        tot = None
        for d in lanes:
            st_ = d.collect_stats()
            tot = st_ if tot is None else {k: (min(tot[k], st_[k]) if k == "shader_mhz" else tot[k] + st_[k]) for k in tot}
        for k in ("tail_fraction", "culled_fraction"):
            tot[k] /= len(lanes)
        return tot

    def step():
        if pipe is not None:
            pipe.submit()
        else:
            render(slab)

    def fence():
        if pipe is not None:
            pipe.drain()
            dist.barrier()
        torch.cuda.synchronize()

    # The REFERENCE's work for this rank's share of the frame: one untimed launch with the escape shortcut off marches every ray to its end as
    # the reference does (its counters are the oracle's); the roofline prices that work, the timed launches run the product's defaults
    ref = None
    if not args.no_reference_launch:
        dev.set_option("escape", 0); dev.set_option("reuse", 0)
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr(), **tiling)
        torch.cuda.synchronize()
        ref = ds.collect_stats()
        dev.set_option("escape", 1); dev.set_option("reuse", 1)

    if pipe is not None:                          # both render lanes settle (lean-kernel placement is timed per context) before anything counts
        for d_ in lanes:
            d_.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr(), **tiling)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    collect()                                     # drop warm-up counters / events
    if pipe is not None:
        pipe.timings()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    st = collect()                                # exact counters + HIP-event kernel time of the K timed launches

    # BASELINE.json words configs[2] with "4 spp".  The reference samples once per pixel (Image.fs:28-34) and
    # `value` is measured on that; the 4-sample EXTENSION of the same frame is timed beside it (N = 1 only).
    spp4 = None
    if world == 1 and pipe is None and not args.no_spp4:
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr(), spp=4)
        torch.cuda.synchronize(); ds.collect_stats()
        t4 = time.perf_counter()
        for _ in range(args.steps):
            ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr(), spp=4)
        torch.cuda.synchronize()
        d4 = time.perf_counter() - t4
        s4 = ds.collect_stats()
        spp4 = {"value": round((s4["rays_primary"] + s4["rays_shadow"]) / d4 / 1e6, 3), "unit": "Mrays/s",
                "ms_per_step": round(d4 / args.steps * 1e3, 3), "rays_per_frame": (s4["rays_primary"] + s4["rays_shadow"]) // args.steps,
                "kernel": "ft_trace_kernel_smooth_spheres_ext + ft_resolve_kernel",
                "note": "EXTENSION: 2x2 corner-offset samples per pixel, fixed-order resolve; sample 0 is the reference's sample"}
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr())          # the 1-spp frame again, for the oracle check
        torch.cuda.synchronize(); ds.collect_stats()

    # The same K frames as a stream: two contexts on two streams render alternate frames, so the few long rays a
    # frame ends on drain while the next frame already fills the GPU (what the N > 1 pipeline does on every rank).
    # Reported beside `value`, which stays the one-stream figure that `roofline` describes.
    streamed = None
    if world == 1 and pipe is None and not args.no_spp4:
        lane_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        devs2 = [ft.Device(local_rank), ft.Device(local_rank)]
        for d_, s_ in zip(devs2, lane_streams):
            d_.set_option("math", math_variant)
            d_.set_stream(s_.cuda_stream)
        dss2 = [d_.scene(scene) for d_ in devs2]
        slabs2 = [slab, torch.empty_like(slab)]
        for i in range(2):
            dss2[i].render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slabs2[i].data_ptr())
        torch.cuda.synchronize()
        for d_ in dss2: d_.collect_stats()
        ts = time.perf_counter()
        for k in range(args.steps):
            dss2[k & 1].render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slabs2[k & 1].data_ptr())
        torch.cuda.synchronize()
        dts = time.perf_counter() - ts
        rays_s = 0
        for d_ in dss2:
            st_ = d_.collect_stats()
            rays_s += st_["rays_primary"] + st_["rays_shadow"]
        streamed = {"value": round(rays_s / dts / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dts / args.steps * 1e3, 3),
                    "note": "frames alternate between two contexts on two streams; the drain of frame k overlaps frame k+1"}
        for d_ in devs2: d_.close()
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr())          # slab = the 1-spp frame of `ds` again
        torch.cuda.synchronize(); ds.collect_stats()


    # Image.render returns a HOST FColor[,] (Image.fs:26-35): the same K frames through ft_render into a page-locked host
    # array (ft_host_register) — four column chunks on two streams, every finished chunk copied behind the rendering.
    host_out = None
    if world == 1 and pipe is None and not args.no_spp4:
        host = np.zeros((W, H, 3), np.float32)
        dev.host_register(host)
        ds.render(syn.EPSILON, syn.RAY_LENGTH, size, cam, out=host)
        th = time.perf_counter()
        for _ in range(args.steps):
            _, hst = ds.render(syn.EPSILON, syn.RAY_LENGTH, size, cam, out=host)
        dth = time.perf_counter() - th
        same = bool(np.array_equal(host.view(np.uint32), slab.cpu().numpy().view(np.uint32)))
        dev.host_unregister(host)
        host_out = {"value": round((hst["rays_primary"] + hst["rays_shadow"]) * args.steps / dth / 1e6, 3), "unit": "Mrays/s",
                    "ms_per_step": round(dth / args.steps * 1e3, 3), "identical_to_device_frame": same,
                    "note": "ft_render: frame delivered in host memory (201 MB over PCIe per 4096^2 frame), destination page-locked "
                            "once; 4 column chunks on 2 streams, copies overlap the rendering"}
        del host

    # The north star's target: primary + ambient-occlusion rays at 4096^2 on the 256-op scene (EXTENSION: the reference has
    # no AO; 4 rays of radius 0.5 per primary hit), and the reference's own workload, Program.fs:14-83, at 4000^2.
    target = None
    console = None
    if world == 1 and pipe is None and not args.no_spp4 and W == 4096:
        kw = dict(ao_samples=4, ao_radius=0.5)
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr(), **kw)
        torch.cuda.synchronize(); ds.collect_stats()
        ta = time.perf_counter()
        for _ in range(args.steps):
            ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr(), **kw)
        torch.cuda.synchronize()
        dta = time.perf_counter() - ta
        sa = ds.collect_stats()
        rays_a = sa["rays_primary"] + sa["rays_shadow"] + sa["rays_ext"]
        target = {"value": round(rays_a / dta / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dta / args.steps * 1e3, 3),
                  "primary": sa["rays_primary"] // args.steps, "ao": sa["rays_ext"] // args.steps, "shadow": sa["rays_shadow"] // args.steps,
                  "kernel": "ft_trace_kernel_smooth_spheres_ext", "target": ">= 100 Mrays/s primary+AO at 4096^2, 256 SDF ops, >= 40 % VALU-busy",
                  "note": "EXTENSION (no reference counterpart): C3 scene, 1 spp, 4 AO rays of radius 0.5 per primary hit; rays = primary + AO + shadow"}
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr())
        torch.cuda.synchronize(); ds.collect_stats()

        cscene, _ = syn.console_scene()
        cds = dev.scene(cscene)
        CW = 4000
        csize = ft.ImageSize(CW, CW)
        cbuf = torch.empty((CW, CW, 3), dtype=torch.float32, device="cuda")
        dev.set_option("escape", 0); dev.set_option("lazy_union", 0); dev.set_option("reuse", 0)   # the reference's evaluation count: every ray marched to its end, every union walk run to its end (second launch: warm)
        cds.render_device(syn.EPSILON, syn.RAY_LENGTH, csize, cam, cbuf.data_ptr())
        torch.cuda.synchronize(); cds.collect_stats()
        cds.render_device(syn.EPSILON, syn.RAY_LENGTH, csize, cam, cbuf.data_ptr())
        torch.cuda.synchronize(); cref = cds.collect_stats()
        dev.set_option("escape", 1); dev.set_option("lazy_union", 1); dev.set_option("reuse", 1)
        cds.render_device(syn.EPSILON, syn.RAY_LENGTH, csize, cam, cbuf.data_ptr())
        torch.cuda.synchronize(); cds.collect_stats()
        for _ in range(args.steps):
            cds.render_device(syn.EPSILON, syn.RAY_LENGTH, csize, cam, cbuf.data_ptr())
        torch.cuda.synchronize()
        cst = cds.collect_stats()
        console = {"stats": cst, "frame": cbuf, "scene": cscene, "W": CW, "ref_evals": cref["sdf_evals"], "ref_kernel_ms": cref["kernel_ms"]}

    # FT_OPT_MATH = glibc: the same K frames with MathF.Exp / Log as this host's C runtime computes them (glibc's expf / logf restated on
    # the GPU, csrc/ft_libm.h) — the arithmetic the reference's CPU path would use on this very machine.  Checked further down against the
    # oracle switched to the real libm; `value` stays the default (fixed) arithmetic.
    libm = None
    if world == 1 and pipe is None and not args.no_spp4 and math_variant == 0:
        variant = ft.glibc_build_of_this_host()
        fixed_frame = slab.clone()
        dev.set_option("math", variant)
        ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr())
        torch.cuda.synchronize(); ds.collect_stats()
        for _ in range(args.steps):
            ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size, cam, slab.data_ptr())
        torch.cuda.synchronize()
        sl_ = ds.collect_stats()
        dev.set_option("math", 0)
        lib_frame = slab.clone()
        a64, b64 = fixed_frame.double(), lib_frame.double()
        rel = ((a64 - b64).abs() / b64.abs().clamp_min(1e-3)).amax(dim=2)
        same = (fixed_frame.view(torch.int32) == lib_frame.view(torch.int32)).all(dim=2)
        rays_l = (sl_["rays_primary"] + sl_["rays_shadow"]) / args.steps
        kms_l = sl_["kernel_ms"] / args.steps
        flops_l = algorithmic_flops({"sdf_evals": sl_["sdf_evals"] // args.steps, "hits_primary": sl_["hits_primary"] // args.steps,
                                     "rays_shadow": sl_["rays_shadow"] // args.steps}, args.spheres * (1.0 - sl_["culled_fraction"]))
        libm = {"value": round(rays_l / kms_l / 1e3, 1), "unit": "Mrays/s", "kernel_ms": round(kms_l, 3), "arithmetic": "glibc_fma" if variant == 1 else "glibc_sse2",
                "kernel": "ft_trace_kernel_smooth_spheres_libm", "glibc_build": "FMA" if variant == 1 else "SSE2",
                "roofline": {"bound": "valu", "peak": VALU_PEAK_TLANEOPS, "unit": "TFLOP/s", "achieved": round(flops_l / (kms_l / 1e3) / 1e12, 3),
                             "frac": round(flops_l / (kms_l / 1e3) / 1e12 / VALU_PEAK_TLANEOPS, 4),
                             "note": "executed work priced with the same flop counts as the headline (exp = 1 flop); this arithmetic spends ~10 double-precision "
                                     "operations per exponential where the fixed one spends 11 single-precision ones"},
                "fixed_vs_glibc": {"pixels_identical": round(float(same.double().mean()), 4), "pixels_over_1e-4_relative": round(float((rel > 1e-4).double().mean()), 6),
                                   "max_relative": float(rel.max())},
                "frame": lib_frame,
                "note": "MathF.Exp / Log = glibc 2.35 expf / logf, restated in double precision on the GPU and proved equal to this host's libm on "
                        "every float (tests: test_glibc_restatement_on_the_device); fixed_vs_glibc is the distance the default arithmetic keeps "
                        "from it on this frame — what DESIGN.md section 2 calls the unpinnable residual"}
        slab.copy_(fixed_frame)
        del fixed_frame, a64, b64

    # BASELINE.json config 4 beside the metric's config (N > 1 only): the same scene at 8192x8192 in the same interleaved column stripes, the slabs
    # brought to rank 0 by the path's ONE gather — frame by frame, not pipelined, so that the per-rank kernel time and the gather time stand alone
    c4 = None
    if use_dist and not args.no_c4 and W == 4096:
        W8 = 8192
        size8, tiling8 = ft.ImageSize(W8, W8), ftd.tiling(W8, world, rank, STRIPE)
        slab8 = torch.empty((W8 // world, W8, 3), dtype=torch.float32, device="cuda")
        recv8, frame8 = ftd.gather_buffers(slab8, world, rank, args.force_dist)
        torch.cuda.synchronize(); ds.collect_stats()
        k8, wall8, gath8 = 2, 0.0, 0.0
        for i in range(k8 + 1):                                       # first frame untimed
            dist.barrier(); torch.cuda.synchronize()
            if i == 1:
                ds.collect_stats()
            t8 = time.perf_counter()
            ds.render_device(syn.EPSILON, syn.RAY_LENGTH, size8, cam, slab8.data_ptr(), **tiling8)
            torch.cuda.synchronize()
            tg = time.perf_counter()
            out8 = ftd.gather_frame(slab8, world, rank, STRIPE, frame=frame8, recv=recv8, force=args.force_dist)
            torch.cuda.synchronize()
            if i >= 1:
                wall8 += time.perf_counter() - t8; gath8 += time.perf_counter() - tg
        st8 = ds.collect_stats()
        mine8 = torch.tensor([st8["kernel_ms"] / k8, gath8 / k8 * 1e3, wall8 / k8 * 1e3, float(st8["rays_primary"] + st8["rays_shadow"]) / k8],
                             dtype=torch.float64, device="cuda")
        all8 = [torch.empty_like(mine8) for _ in range(world)]
        dist.all_gather(all8, mine8)
        if rank == 0:
            ms8 = max(float(v[2]) for v in all8)
            c4 = {"workload": f"C4: the same scene at {W8}x{W8}, 1 spp, column stripes of {STRIPE} over {world} GPU(s) + 1 RCCL gather; frame by frame (not pipelined)",
                  "value": round(sum(float(v[3]) for v in all8) / ms8 / 1e3, 3), "unit": "Mrays/s", "ms_per_frame": round(ms8, 3), "frames": k8,
                  "per_rank": [{"rank": r, "kernel_ms": round(float(v[0]), 3), "gather_and_deinterleave_ms": round(float(v[1]), 3)} for r, v in enumerate(all8)],
                  "gather_bytes_per_rank": int(slab8.numel() * 4), "frame": out8,
                  "note": "gather time of a rank includes waiting for the slowest rank; rank 0's also the strided de-interleave copy on the device"}
        del slab8

    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    cnt = torch.tensor([st["rays_primary"], st["rays_shadow"], st["sdf_evals"], st["hits_primary"], st["flags"],
                        *((ref["sdf_evals"], ref["hits_primary"], ref["rays_shadow"]) if ref is not None else
                          (st["sdf_evals"] // args.steps, st["hits_primary"] // args.steps, st["rays_shadow"] // args.steps))], dtype=torch.int64, device="cuda")
    kms = torch.tensor([st["kernel_ms"]], dtype=torch.float64, device="cuda")
    per_rank = None
    if use_dist:
        # a SCALE record should explain itself: every rank's kernel time, shader clock, gather and de-interleave time per frame
        g_ms, d_ms = pipe.timings()
        mine = torch.tensor([st["kernel_ms"] / args.steps, st["shader_mhz"], g_ms / args.steps, d_ms / args.steps, dt / args.steps * 1e3],
                            dtype=torch.float64, device="cuda")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "kernel_ms": round(float(v[0]), 3), "shader_mhz": round(float(v[1]), 1), "gather_ms": round(float(v[2]), 3),
                     "deinterleave_ms": round(float(v[3]), 3), "ms_per_step": round(float(v[4]), 3)} for r, v in enumerate(allr)]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    rays_primary, rays_shadow, evals, hits, flags, ref_evals, ref_hits, ref_shadow = (int(v) for v in cnt.tolist())

    if rank == 0:
        rays = rays_primary + rays_shadow
        mrays = rays / dt / 1e6
        # roofline of the dominant (only) kernel, from this rank's launches: algorithmic lane-ops per
        # launch / mean HIP-event duration of a launch
        build = ft.build_info()
        kernel_name = "ft_trace_kernel_smooth_spheres" + ("_libm" if math_variant else "")
        traffic, traffic_src, valu_busy = profile_figures(build["src"], "'" + kernel_name + "'") if (W == 4096 and world == 1) else (None, None, None)
        if traffic_src is None:
            traffic_src = "no profiles/*_summary.txt carries the stamp of this build (src=%s): traffic / valu_busy_pmc not quoted" % build["src"]
        # algorithmic = the reference's work: every evaluation of every ray marched to its end, every child in every evaluation (whole job; at
        # N > 1 the kernel time is the slowest rank's, so the figure is priced per rank share)
        flops_launch = algorithmic_flops({"sdf_evals": ref_evals, "hits_primary": ref_hits, "rays_shadow": ref_shadow}, args.spheres) / world
        launch_s = float(kms.item()) / 1e3 / args.steps
        achieved = flops_launch / launch_s / 1e12
        # executed = what the kernel did: fewer evaluations (escape shortcut), and in each only the children that were not culled
        culled = st["culled_fraction"]
        flops_exec = algorithmic_flops({"sdf_evals": evals // args.steps, "hits_primary": hits // args.steps, "rays_shadow": rays_shadow // args.steps},
                                       args.spheres * (1.0 - culled)) / world
        achieved_exec = flops_exec / launch_s / 1e12
        out = {
            "metric": "Mrays/s (primary+secondary) at 4096x4096; max per-pixel |delta| vs F# ref "
                      "(delta is measured against the CPU oracle: F# parity is unpinned, DESIGN.md section 2)",
            "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "build": build,
            "config": {"workload": f"C3: unionSmooth(0.25) of {args.spheres} spheres, {W}x{H}, 1 spp, 1 directional light "
                                   "(shadow rays = secondary rays), eps 0.01, ray length 30",
                       "arithmetic": arithmetic,
                       "rays_per_frame": rays // args.steps, "primary": rays_primary // args.steps,
                       "shadow": rays_shadow // args.steps, "sdf_evals_per_frame": ref_evals,
                       "sdf_evals_executed_per_frame": evals // args.steps, "children_culled_fraction": round(culled, 4),
                       "parallelism": f"column stripes of {STRIPE} over {world} GPU(s) + 1 RCCL gather" if world > 1 else "1 GPU",
                       "nan_or_cap_flags": flags,
                       "lane_utilisation": round(st["sdf_evals"] / (64.0 * max(1, st["wave_evals"])), 4),
                       "shader_mhz": round(st["shader_mhz"], 1)},
            "roofline": {"bound": "valu", "achieved": round(achieved_exec, 3), "peak": VALU_PEAK_TLANEOPS, "unit": "TFLOP/s",
                         "frac": round(achieved_exec / VALU_PEAK_TLANEOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "achieved_executed": round(achieved_exec, 3), "frac_executed": round(achieved_exec / VALU_PEAK_TLANEOPS, 4),
                         "achieved_reference_work": round(achieved, 3), "frac_reference_work": round(achieved / VALU_PEAK_TLANEOPS, 4),
                         "valu_busy_pmc": valu_busy,
                         "kernel": kernel_name,
                         "kernel_ms": round(launch_s * 1e3, 3),
                         "algorithmic_flops_per_launch": int(flops_exec),
                         "algorithmic_flops_per_launch_reference_work": int(flops_launch),
                         "shader_mhz": round(st["shader_mhz"], 1),
                         "shader_Gcycles_per_launch": round(launch_s * st["shader_mhz"] * 1e6 / 1e9, 4),
                         "work_note": "achieved / frac (= achieved_executed / frac_executed) price what the timed launches EXECUTED — the kernel ends rays that can no "
                                      "longer reach the scene's support sphere (sdf_evals_executed_per_frame) and drops, per wave and round, the children whose terms are "
                                      "below half an ulp of the running sum (children_culled_fraction); both are exact (DESIGN.md section 4) — i.e. the hardware fraction, "
                                      "next to the counter view valu_busy_pmc.  achieved_reference_work / frac_reference_work price the REFERENCE's work over the same "
                                      "time (SURVEY.md section 8d: every SDF evaluation of every ray marched to its end, all children in each; counters of an untimed "
                                      "launch with the escape shortcut off = the oracle's): an algorithmic speed-up figure, not a hardware fraction",
                         "note": "f32 lane-ops (FMA counted once; contraction is forbidden by parity); sqrt and exp count as 1 "
                                 "flop each although a correctly rounded sqrt / reproducible exp need 4 / 11 instructions: 26 VALU instructions "
                                 "per child against 13 algorithmic flops (the near loop's root and strength product use output modifiers inside a "
                                 "verified MODE region, DESIGN.md section 4), issued at ~2.4 cycles each "
                                 "(DESIGN.md section 5: at the instruction-issue floor of this mix). peak is priced at 2.4 GHz; the chip "
                                 "sustains shader_mhz under this load (power management), which is the box-to-box spread. "
                                 "HBM traffic = 12 B/pixel output."},
        }
        if per_rank is not None:
            out["config"]["per_rank"] = per_rank
            out["config"]["per_rank_note"] = ("per frame: kernel_ms = mean start-to-end time of this rank's launches (two overlapping lanes), gather_ms = "
                                              "the RCCL gather on the side stream incl. waiting for the slowest rank, deinterleave_ms = rank 0's strided copy")
        if pipe is not None:
            out["roofline"]["launch_overlap"] = ("consecutive frames run on two streams and overlap: kernel_ms is the mean start-to-end "
                                                 "time of a launch, not its exclusive time; frames per second come from ms_per_step")
        if spp4 is not None:
            out["config"]["same_frame_at_4_spp"] = spp4
        if streamed is not None:
            out["config"]["frames_streamed_on_two_lanes"] = streamed
        if host_out is not None:
            out["config"]["host_output"] = host_out
        if target is not None:
            out["config"]["north_star_target"] = target
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], check = cpu_baseline(scene, cam, W, H, args.cpu_columns, slab if pipe is None else pipe.frame, libm=math_variant != 0)
            out["config"]["max_abs_delta_vs_oracle"] = check["max_abs_delta"]      # second half of the metric: 0.0 = bit-exact
            out["config"]["pixels_compared_with_oracle"] = check["pixels"]
            if console is not None:
                out["config"]["program_fs_scene"] = program_fs_block(console, cam, args.steps, build["src"])
            if libm is not None:
                out["config"]["glibc_math_mode"] = libm_block(libm, scene, cam, W, H)
        elif world > 1:
            # N > 1: the gathered frame gets its own parity flag — a few columns against the oracle, after the timed region
            out["config"].update(oracle_check_columns(scene, cam, W, H, pipe.frame, 16, libm=math_variant != 0))
        if c4 is not None:
            frame8 = c4.pop("frame")
            c4.update(oracle_check_columns(scene, cam, 8192, 8192, frame8, 8, libm=math_variant != 0))
            out["config"]["c4_8192"] = c4
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def oracle_check_columns(scene, cam, W, H, frame, ncols, libm=False):
    """max |delta| of `ncols` evenly spaced columns of a device frame against the CPU oracle (the checker, never the thing measured);
    libm: the oracle calls this host's real expf / logf (--math glibc)"""
    import numpy as np
    from oracle import binding as ob
    xstep = max(1, W // ncols)
    ob.lib.orc_set_libm(1 if libm else 0)
    try:
        img, _ = ob.Oracle().scene(scene).render(0.01, 30.0, W, H, cam.as_array(), xstep=xstep, nthreads=host_cpu_share())
    finally:
        ob.lib.orc_set_libm(0)
    got = frame[::xstep].cpu().numpy()
    delta = float(np.max(np.abs(got.astype(np.float64) - img.astype(np.float64)))) if got.shape == img.shape else float("nan")
    return {"max_abs_delta_vs_oracle": delta, "pixels_compared_with_oracle": int(img.shape[0] * img.shape[1])}


def program_fs_block(console, cam, steps, build_src):
    """Side block for the reference's own workload (src/FrayTracer.Console/Program.fs:14-83: System.Random(19), 1000 tori,
    subtract(intersect(union, sphere), sphere), 2 lights) at 4000^2 on the carved-union kernel of its shape (kernels.hip ft_eval_carved): Mrays/s, and its own
    roofline entry priced as SURVEY.md section 8d prescribes for unions, with the oracle's counters of a column sample of the same
    frame (the per-evaluation flop count is scaled by the kernel's exact evaluation count); the sampled columns are compared."""
    import numpy as np
    from oracle import binding as ob
    cst, CW = console["stats"], console["W"]
    xstep = 40
    osc = ob.Oracle().scene(console["scene"])
    img, cnt = osc.render(0.01, 30.0, CW, CW, cam.as_array(), xstep=xstep, nthreads=host_cpu_share())
    got = console["frame"][::xstep].cpu().numpy()
    delta = float(np.max(np.abs(got.astype(np.float64) - img.astype(np.float64))))
    flops_per_eval = union_algorithmic_flops(cnt) / max(1, cnt["root_evals"])
    # what the kernel executes: it leaves a cell's sorted list at the first failing LowerBound test (exact), so it looks at `union_tested`
    # candidates where the reference scans `union_candidates` (both from the oracle's counters of the sampled columns)
    executed = dict(cnt); executed["union_candidates"] = cnt["union_tested"]
    flops_per_eval_executed = union_algorithmic_flops(executed) / max(1, cnt["root_evals"])
    evals = cst["sdf_evals"] / steps                      # executed: rays that can no longer reach the scene's support sphere end at once (FT_OPT_ESCAPE)
    evals_ref = console["ref_evals"]                      # the reference's: every ray marched to its end (one launch with the shortcut off)
    kernel_s = cst["kernel_ms"] / 1e3 / steps
    rays = (cst["rays_primary"] + cst["rays_shadow"]) / steps
    achieved = flops_per_eval * evals_ref / kernel_s / 1e12
    achieved_executed = flops_per_eval_executed * evals / kernel_s / 1e12
    _, prof_src, valu_busy = profile_figures(build_src, "# case: Program.fs scene 4000")
    return {"workload": f"Program.fs scene, {CW}x{CW}, 1 spp, directional + point light", "value": round(rays / kernel_s / 1e6, 1), "unit": "Mrays/s",
            "kernel_ms": round(kernel_s * 1e3, 3), "kernel": "ft_trace_kernel_carved_tori", "rays_per_frame": int(rays), "sdf_evals_per_frame": int(evals_ref),
            "sdf_evals_executed_per_frame": int(evals), "kernel_ms_with_every_ray_marched_and_every_walk_run_to_its_end": round(console["ref_kernel_ms"], 3),
            "lane_utilisation": round(cst["sdf_evals"] / (64.0 * max(1, cst["wave_evals"])), 4), "shader_mhz": round(cst["shader_mhz"], 1),
            "max_abs_delta_vs_oracle": delta, "pixels_compared_with_oracle": int(img.shape[0] * img.shape[1]),
            "roofline": {"bound": "valu", "peak": VALU_PEAK_TLANEOPS, "unit": "TFLOP/s",
                         "achieved": round(achieved_executed, 3), "frac": round(achieved_executed / VALU_PEAK_TLANEOPS, 4),
                         "achieved_reference_work": round(achieved, 3), "frac_reference_work": round(achieved / VALU_PEAK_TLANEOPS, 4),
                         "achieved_executed": round(achieved_executed, 3), "frac_executed": round(achieved_executed / VALU_PEAK_TLANEOPS, 4),
                         "valu_busy_pmc": valu_busy, "valu_busy_source": prof_src,
                         "algorithmic_flops_per_eval_reference": round(flops_per_eval, 1), "algorithmic_flops_per_eval_executed": round(flops_per_eval_executed, 1),
                         "candidates_per_eval_reference": round(cnt["union_candidates"] / max(1, cnt["root_evals"]), 2),
                         "candidates_per_eval_executed": round(cnt["union_tested"] / max(1, cnt["root_evals"]), 2),
                         "primitive_evals_per_eval": round(sum(cnt["prim"]) / max(1, cnt["root_evals"]), 2),
                         "note": "frac_reference_work prices the reference's own work (every ray marched to its end, a cell's whole candidate list scanned, 13 flops per "
                                 "candidate): an algorithmic speed-up figure, not a hardware fraction.  frac_executed prices what the kernel executes: rays that can no "
                                 "longer reach the scene's support sphere end at once (sdf_evals_executed_per_frame; exact) and the walk leaves the sorted list "
                                 "at the first failing LowerBound test (exact), i.e. candidates_per_eval_executed of them — an upper bound since the lazy union "
                                 "under the scene's intersect ends part of the walks at their first candidate.  valu_busy_pmc: PMC of a committed "
                                 "profile of this same build and size, or null"}}


def libm_block(libm, scene, cam, W, H):
    """the glibc-mode frame against the oracle calling this host's real expf / logf (orc_set_libm) on a column sample"""
    import numpy as np
    from oracle import binding as ob
    frame = libm.pop("frame")
    xstep = max(1, W // 64)
    osc = ob.Oracle().scene(scene)
    ob.lib.orc_set_libm(1)
    try:
        img, _ = osc.render(0.01, 30.0, W, H, cam.as_array(), xstep=xstep, nthreads=host_cpu_share())
    finally:
        ob.lib.orc_set_libm(0)
    got = frame[::xstep].cpu().numpy()
    libm["max_abs_delta_vs_oracle_with_libm"] = float(np.max(np.abs(got.astype(np.float64) - img.astype(np.float64))))
    libm["pixels_compared_with_oracle"] = int(img.shape[0] * img.shape[1])
    return libm


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes
    give a job 16 of the host's 256 hardware threads through cpu.max, which os.cpu_count() does not see)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(scene, cam, W, H, ncols, gpu_frame, libm=False):
    """Time the CPU oracle on every (W/ncols)-th column of the same frame, all host threads, and use its
    output as the checker of the GPU frame on those columns (the oracle is never the thing measured as
    `value`)."""
    from oracle import binding as ob
    threads = host_cpu_share()
    osc = ob.Oracle().scene(scene)
    if ncols <= 0:
        ncols = max(32, 32 * threads)         # the oracle's work queue hands out whole columns (Array2D.fs:32)
    xstep = max(1, W // ncols)
    ob.lib.orc_set_libm(1 if libm else 0)       # --math glibc: the oracle calls this host's real expf / logf, as the reference's CPU path does
    try:
        t0 = time.perf_counter()
        img, cnt = osc.render(0.01, 30.0, W, H, cam.as_array(), xstep=xstep, nthreads=threads)
        dt = time.perf_counter() - t0
    finally:
        ob.lib.orc_set_libm(0)
    rays = cnt["rays_primary"] + cnt["rays_shadow"]
    import numpy as np
    got = gpu_frame[::xstep].cpu().numpy()
    delta = float(np.max(np.abs(got.astype(np.float64) - img.astype(np.float64)))) if got.shape == img.shape else float("nan")
    check = {"max_abs_delta": delta, "pixels": int(img.shape[0] * img.shape[1])}
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"every {xstep}th column of the same {W}x{H} frame ({(W + xstep - 1) // xstep} columns, {rays} rays, {dt:.1f} s); "
                      "oracle = C++ restatement of the F# CPU path, std::function closures, x-column work queue",
            "seconds": round(dt, 2)}, check


if __name__ == "__main__":
    main()
