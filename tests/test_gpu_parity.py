"""Parity of the HIP path (through the C ABI) with the CPU oracle: bit-exact float32."""
import numpy as np
import pytest

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from helpers import assert_bit_equal

pytestmark = pytest.mark.gpu

EPS, LEN = syn.EPSILON, syn.RAY_LENGTH


def both(gpu, oracle, scene):
    O = oracle.Oracle()
    return gpu.scene(scene), O.scene(scene)


def render_both(gpu, oracle, scene, W, H, **kw):
    cam = syn.default_camera()
    ds, os_ = both(gpu, oracle, scene)
    g, gst = ds.render(EPS, LEN, ft.ImageSize(W, H), cam)
    o, ocnt = os_.render(EPS, LEN, W, H, cam.as_array())
    return g, gst, o, ocnt


def check_counts(gst, ocnt):
    assert gst["rays_primary"] == ocnt["rays_primary"]
    assert gst["rays_shadow"] == ocnt["rays_shadow"]
    assert gst["hits_primary"] == ocnt["hits_primary"]
    assert gst["hits_shadow"] == ocnt["hits_shadow"]
    assert gst["flags"] == ocnt["flags"] == 0


def test_product_build_is_the_one_under_test(gpu):
    """the library this GPU suite runs on is the product build of the sources that travelled with it (ft_build_info against
    csrc/source_hash.py) — round 2 ran one suite on a stray experiment build without noticing"""
    info = ft.build_info()
    assert info["kind"] == "product" and info["src"] == ft.source_hash(), (info, ft.source_hash())


def test_statistics_reset_is_ordered_with_the_next_launch(gpu, oracle):
    """Regression for round 2's null-stream hipMemset of the statistics block (capi.cpp ft_collect_stats): the context's stream is
    non-blocking, so the reset could still be pending when the next — here very short — launch added its counters.  300 tiny
    renders back to back, each with the oracle's exact counters, on the context's own stream and on a caller's stream."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")           # the runtime the library is linked against (already loaded: same instance)
    side = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(side), 1) == 0          # hipStreamNonBlocking
    scene, _ = syn.config2(seed=3, size=64)
    ds, os_ = both(gpu, oracle, scene)
    cam = syn.default_camera()
    want = {}
    for n in (8, 16, 24):
        want[n] = os_.render(EPS, LEN, n, n, cam.as_array())[1]
    try:
        for rep in range(300):
            if rep == 150:
                gpu.set_stream(side.value)
            n = (8, 16, 24)[rep % 3]
            _, st = ds.render(EPS, LEN, ft.ImageSize(n, n), cam)
            for k in ("rays_primary", "rays_shadow", "hits_primary", "hits_shadow"):
                assert st[k] == want[n][k], (rep, n, k, st[k], want[n][k])
    finally:
        gpu.set_stream(0)
        hip.hipStreamDestroy(side)


def test_math_exp_log_sqrt_div(gpu, oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-110, 92, 2_000_000), rng.uniform(-2, 2, 1_000_000),
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 88.72283, 88.72284, -103.97, -103.98, -104.5, 1e-40, -1e-40]]).astype(np.float32)
    assert_bit_equal(gpu.math_eval(0, x), oracle.expf(x), "exp")
    u = rng.integers(0, 0x7F800000, 3_000_000, dtype=np.uint32).view(np.float32)       # all positive finite floats incl. subnormals
    u = np.concatenate([u, np.array([0.0, -0.0, np.inf, -1.0, np.nan, 1.0, 1e-45], np.float32)])
    assert_bit_equal(gpu.math_eval(1, u), oracle.logf(u), "log")
    assert_bit_equal(gpu.math_eval(2, u), oracle.sqrtf(u), "sqrt")
    a = rng.integers(0, 0xFFFFFFFF, 2_000_000, dtype=np.uint32).view(np.float32)
    b = rng.integers(0, 0xFFFFFFFF, 2_000_000, dtype=np.uint32).view(np.float32)
    with np.errstate(all="ignore"):
        assert_bit_equal(gpu.math_eval(3, a, b), a / b, "div")


def test_fast_sqrt_exp_exhaustive(gpu):
    """The guarded fast sqrt / exp of the smooth-union loop equal the exact forms on EVERY float of
    their guarded ranges (2^-96..2^100 for sqrt, -87..88 for exp): proof by exhaustion, on the GPU."""
    assert gpu.selftest_fastmath() == {"sqrt": 0, "exp": 0, "exp_near": 0}


def test_smooth_union_guard_fallbacks(gpu, oracle):
    """Points that force the slow path of the smooth-union loop: far away (t < -87, subnormal and zero
    exponentials, log(0) = -inf -> +inf distance), exactly on a sphere centre (q = 0), huge / non-finite."""
    scene, _ = syn.config3(n=37)        # 37: exercises the 4-wide blocks and the scalar remainder
    ds, os_ = both(gpu, oracle, scene)
    O = oracle.Oracle()
    form = O.object_form(os_.object)
    rng = np.random.default_rng(5)
    centre = np.array(scene.Object.kids[1].kids[0].args[0], np.float32)
    pts = np.concatenate([
        rng.uniform(-6, 6, (4000, 3)), rng.uniform(-60, 60, (4000, 3)), rng.uniform(-3000, 3000, (2000, 3)),
        [centre, centre + np.float32(1e-30), centre + np.float32(1e-12)],
        [[1e18, 0, 0], [3e38, 3e38, 3e38], [np.inf, 0, 0], [np.nan, 0, 0], [0, 0, 25.9], [0, 0, 21.7], [0, 0, 21.8]],
    ]).astype(np.float32)
    d, _ = ds.eval_distance(pts)
    with np.errstate(all="ignore"):
        assert_bit_equal(d, O.form_distance(form, pts), "smooth-union distance incl. guard fallbacks")
    assert np.isinf(d).any() and np.isnan(d).any()


@pytest.fixture
def glibc_mode(gpu, oracle):
    """FT_OPT_MATH = the glibc build this host's libm resolves to, with the oracle switched to the C runtime's expf / logf / powf"""
    variant = ft.glibc_build_of_this_host()
    gpu.set_option("math", variant)
    oracle.lib.orc_set_libm(1)
    try:
        yield variant
    finally:
        oracle.lib.orc_set_libm(0)
        gpu.set_option("math", 0)


def test_glibc_restatement_on_the_device(gpu, oracle):
    """ft_selftest_libm: the device restatement of glibc's expf / logf / powf(x, 1/2.2f) over EVERY float (256 chunks of 2^24 bit
    patterns, one checksum each) against the C runtime of the machine this test runs on — equal sums chunk by chunk.  The variant
    is the build that libm resolves to here (FMA on every MI355X host); the other variant must NOT match for expf (the two builds
    differ on a handful of inputs), which shows the checksum can tell them apart."""
    import os
    variant = ft.glibc_build_of_this_host()
    nthreads = max(1, min(16, len(os.sched_getaffinity(0))))
    y = float(np.float32(1.0) / np.float32(2.2))
    for op, yy in ((0, 0.0), (1, 0.0), (2, y)):
        want = oracle.libm_checksums(op, yy, 0, 256, nthreads)
        got = gpu.selftest_libm(op, variant, yy, 0, 256)
        bad = np.nonzero(want != got)[0]
        assert bad.size == 0, (op, variant, bad[:8])
        if op == 0:
            other = gpu.selftest_libm(op, 3 - variant, yy, 0, 256)
            assert 0 < int((other != want).sum()) <= 8
    # array entry points of the same functions (ft_math_eval ops 6 .. 11), special values included
    x = np.concatenate([np.random.default_rng(2).normal(0, 30, 20000), [0, -0.0, np.inf, -np.inf, np.nan, 88.7, 88.8, -103.9, -104.1, 1e-40, 1.0]]).astype(np.float32)
    with np.errstate(all="ignore"):
        assert_bit_equal(gpu.math_eval(5 + variant, x), oracle.libm_array(0, x), "expf")
        e, o = gpu.math_eval(7 + variant, x), oracle.libm_array(1, x)
        assert np.array_equal(np.isnan(e), np.isnan(o)); assert_bit_equal(np.nan_to_num(e, nan=0.0), np.nan_to_num(o, nan=0.0), "logf")
        yv = np.random.default_rng(3).normal(0, 3, x.size).astype(np.float32)
        e, o = gpu.math_eval(9 + variant, x, yv), oracle.libm_array(2, x, yv)
        assert np.array_equal(np.isnan(e), np.isnan(o)); assert_bit_equal(np.nan_to_num(e, nan=0.0), np.nan_to_num(o, nan=0.0), "powf")


def test_glibc_sse2_build_on_the_device_in_a_child_process():
    """The OTHER glibc build.  A child process started under GLIBC_TUNABLES=glibc.cpu.hwcaps=-FMA,-AVX2 has a libm that resolves
    expf / logf / powf to their SSE2 builds; there FT_MATH_GLIBC_SSE2 must equal it: all 256 checksum chunks (every float) per function,
    and a C3 frame against the oracle calling that libm — so both variants of the device restatement are proved on the GPU, whichever
    CPU the box has."""
    import os, subprocess, sys
    if ft.glibc_build_of_this_host() != 1:
        pytest.skip("this CPU has no FMA / AVX2: the SSE2 build is what the other tests already ran against")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from oracle import binding as ob
assert ft.glibc_build_of_this_host() == 2
dev = ft.Device(0)
y = float(np.float32(1.0) / np.float32(2.2))
nthreads = max(1, min(16, len(os.sched_getaffinity(0))))
for op, yy in ((0, 0.0), (1, 0.0), (2, y)):
    want = ob.libm_checksums(op, yy, 0, 256, nthreads)
    got = dev.selftest_libm(op, 2, yy, 0, 256)
    assert (want == got).all(), (op, np.nonzero(want != got)[0][:8])
assert not (dev.selftest_libm(0, 1, 0.0, 0, 256) == ob.libm_checksums(0, 0.0, 0, 256, nthreads)).all()     # the FMA variant is not this libm
scene, _ = syn.config3(n=256, size=96)
cam = syn.default_camera()
dev.set_option("math", 2); ob.lib.orc_set_libm(1)
g, gst = dev.scene(scene).render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(96, 96), cam)
o, ocnt = ob.Oracle().scene(scene).render(syn.EPSILON, syn.RAY_LENGTH, 96, 96, cam.as_array())
assert np.array_equal(g.view(np.uint32), o.view(np.uint32)) and gst["rays_shadow"] == ocnt["rays_shadow"]
print("SSE2-OK")
""" % root
    env = dict(os.environ, GLIBC_TUNABLES="glibc.cpu.hwcaps=-FMA,-AVX2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SSE2-OK" in out.stdout, out.stderr[-2000:]


def test_glibc_math_mode_renders_what_the_oracle_renders_with_libm(gpu, oracle, glibc_mode):
    """FT_OPT_MATH = glibc: SdfForm.unionSmooth's MathF.Exp / MathF.Log (SdfForm.fs:80,82) as this host's C runtime computes them.
    The oracle calls the real expf / logf (orc_set_libm); the kernels run the restatement: bit-exact images and exact counters for
    the lean kernel (C3), its EXTENSION build, the general kernel (smooth unions next to every other combinator), the call-capable
    kernel and single distances incl. the far / centre / non-finite points.  In the default mode the same frames differ."""
    cam = syn.default_camera()
    scene, _ = syn.config3(n=256, size=256)
    ds, os_ = both(gpu, oracle, scene)
    assert ds.info()["fast_path"] == 1
    g, gst = ds.render(EPS, LEN, ft.ImageSize(256, 256), cam)
    o, ocnt = os_.render(EPS, LEN, 256, 256, cam.as_array())
    assert_bit_equal(g, o, "C3 256^2 in glibc mode")
    check_counts(gst, ocnt)
    gpu.set_option("math", 0)
    g0, _ = ds.render(EPS, LEN, ft.ImageSize(256, 256), cam)
    gpu.set_option("math", glibc_mode)
    assert not np.array_equal(g0.view(np.uint32), g.view(np.uint32))           # the two arithmetics are different functions
    g4, _ = ds.render(EPS, LEN, ft.ImageSize(96, 96), cam, spp=4, ao_samples=3, ao_radius=0.5)
    o4, _ = os_.render(EPS, LEN, 96, 96, cam.as_array(), spp=4, ao_samples=3, ao_radius=0.5)
    assert_bit_equal(g4, o4, "C3 EXTENSION build in glibc mode")
    for name, sc in (("mixed nested", syn.mixed_nested()[0]), ("combinator zoo", syn.combinator_zoo()[0]), ("crowd (call children)", syn.combinator_crowd(n=60)[0])):
        ds2, os2 = both(gpu, oracle, sc)
        g, gst = ds2.render(EPS, LEN, ft.ImageSize(96, 96), cam)
        o, ocnt = os2.render(EPS, LEN, 96, 96, cam.as_array())
        assert_bit_equal(g, o, name + " in glibc mode")
        check_counts(gst, ocnt)
    # single evaluations, guard regimes of the sphere run included
    scene37, _ = syn.config3(n=37)
    ds3, os3 = both(gpu, oracle, scene37)
    O = oracle.Oracle()
    form = O.object_form(os3.object)
    rng = np.random.default_rng(5)
    centre = np.array(scene37.Object.kids[1].kids[0].args[0], np.float32)
    pts = np.concatenate([rng.uniform(-6, 6, (4000, 3)), rng.uniform(-60, 60, (4000, 3)), rng.uniform(-3000, 3000, (2000, 3)),
                          [centre, centre + np.float32(1e-30)], [[1e18, 0, 0], [3e38, 3e38, 3e38], [np.inf, 0, 0], [np.nan, 0, 0], [0, 0, 25.9]]]).astype(np.float32)
    d, _ = ds3.eval_distance(pts)
    with np.errstate(all="ignore"):
        want = O.form_distance(form, pts)
    assert np.array_equal(np.isnan(d), np.isnan(want))
    assert_bit_equal(np.nan_to_num(d, nan=0.0), np.nan_to_num(want, nan=0.0), "smooth-union distances in glibc mode")
    # a scene without a unionSmooth has no exponential: the option changes nothing (and no *_libm kernel is launched)
    c2, _ = syn.config2(seed=4, size=64)
    ds4 = gpu.scene(c2)
    a, _ = ds4.render(EPS, LEN, ft.ImageSize(64, 64), cam)
    gpu.set_option("math", 0)
    b, _ = ds4.render(EPS, LEN, ft.ImageSize(64, 64), cam)
    assert_bit_equal(a, b, "C2 is the same in every math mode")


def test_glibc_math_mode_tone_map(gpu, oracle, glibc_mode):
    """FColor.gammaInverse's MathF.Pow (FColor.fs:50-55) as this host's powf: bytes of the device tone map = the oracle's with libm"""
    rng = np.random.default_rng(8)
    img = (rng.random((120, 77, 3)) ** 3 * 4.0).astype(np.float32)
    img[3, 2] = (np.nan, 0.0, -1.0); img[9, 0] = 0.0; img[10, 5] = (1e-30, 1e-38, 1e-44)
    for gamma in (2.2, 1.8, 1.0):
        for seed, bmp in ((None, False), (19, True)):
            want, wmx = oracle.tone_map(img, gamma=gamma, seed=seed, bmp_order=bmp)
            got = ft.Image.toColors(gamma, seed, img, gpu, bmp_order=bmp)
            assert np.array_equal(got, want), (gamma, seed, bmp)


@pytest.mark.parametrize("k", [0, 3, 64])
def test_latency_mode_is_bit_identical(gpu, oracle, k):
    """FT_OPT_TAIL_K: a wave that holds at most k rays evaluates each of them with all 64 lanes (lean kernel: exponentials across the
    lanes, sum in child order; general kernels: the cell's candidates across the lanes, decisions replayed in list order).  k = 0 is the
    one-ray-per-lane path only, k = 64 sends EVERY evaluation through the cooperative code, k = 3 mixes both: images, counters and flags
    must be those of the oracle in every case — smooth unions (all three regimes of the sphere run, non-multiple-of-4 and > 256
    children), grid unions of every primitive kind, slots, on-demand sub-programs, NaN rays, EXTENSION builds, the glibc arithmetic."""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    cam = syn.default_camera()
    gpu.set_option("tail_k", k)
    try:
        cases = [("C3 n=256", syn.config3(n=256, size=96)[0], 96, {}), ("C3 n=37", syn.config3(n=37, size=64)[0], 64, {}),
                 ("C3 n=300 (two segments)", syn.config3(n=300, size=48)[0], 48, {}), ("C3 ext", syn.config3(n=64, size=64)[0], 64, dict(spp=4, ao_samples=3, ao_radius=0.5)),
                 ("C2", syn.config2(seed=6, size=96)[0], 96, {}), ("C2 boxes + AO", syn.config2(boxes=True, size=64)[0], 64, dict(ao_samples=4, ao_radius=0.5)),
                 ("console-like 300", syn.console_like(n=300)[0], 96, {}), ("mixed nested", syn.mixed_nested()[0], 96, {}),
                 ("combinator zoo", syn.combinator_zoo()[0], 80, {}), ("crowd", syn.combinator_crowd(n=80)[0], 80, {}),
                 ("glass", syn.config5()[0], 48, dict(spp=4, spectral=4, max_bounces=4))]
        for name, scene, n, kw in cases:
            ds, os_ = both(gpu, oracle, scene)
            g, gst = ds.render(EPS, LEN, ft.ImageSize(n, n), cam, **kw)
            o, ocnt = os_.render(EPS, LEN, n, n, cam.as_array(), **kw)
            assert_bit_equal(g, o, f"{name}, tail_k = {k}")
            check_counts(gst, ocnt)
            assert (gst["tail_fraction"] == 0.0) if k == 0 else (gst["tail_fraction"] > 0.0), (name, gst["tail_fraction"])
            if k == 64: assert gst["tail_fraction"] == 1.0
        # far / centre / non-finite points through the ray-buffer entry (every regime of the sphere run), and NaN rays through triangles
        scene37, _ = syn.config3(n=37)
        ds3, os3 = both(gpu, oracle, scene37)
        centre = np.array(scene37.Object.kids[1].kids[0].args[0], np.float32)
        rng = np.random.default_rng(9)
        o_ = np.concatenate([rng.uniform(-6, 6, (300, 3)), rng.uniform(-3000, 3000, (100, 3)), [centre, centre + np.float32(1e-30)], [[1e18, 0, 0], [0, 0, 25.9]]]).astype(np.float32)
        d_ = rng.normal(0, 1, o_.shape).astype(np.float32); d_ /= np.linalg.norm(d_, axis=1, keepdims=True)
        rays = np.concatenate([o_, d_, np.full((len(o_), 1), 30, np.float32), np.full((len(o_), 1), 0.01, np.float32)], axis=1)
        with np.errstate(all="ignore"):
            assert_bit_equal(ds3.trace_rays(rays)[0], os3.trace_rays(rays)[0], f"ray buffer, tail_k = {k}")
        tri = lambda t: SdfForm.Primitive.triangle((-1 + t, -1, 0), (1 + t, -1, 0.2), (0 + t, 1, -0.1), 0.2)
        mat = SdfMaterial.createSolid((0.3, 0.6, 0.9))
        nan_rays = np.array([[0, 0, -5, 0, 0, 1, 30, 0.01], [0, 0, -5, np.nan, 0, 1, 30, 0.01], [0.1, 0, -5, 0, np.nan, np.nan, 30, 0.01]], np.float32)
        sc = SdfScene(SdfObject.union([SdfObject.create(mat, tri(0)), SdfObject.create(mat, tri(0.5)), SdfObject.create(mat, tri(-0.5))]), syn.BACKGROUND, syn.program_lights())
        dsn, osn = both(gpu, oracle, sc)
        gr, gst = dsn.trace_rays(nan_rays)
        orr, ocnt = osn.trace_rays(nan_rays)
        assert_bit_equal(gr, orr, "NaN rays"); assert gst["flags"] == ocnt["flags"] == 3
        # the glibc arithmetic through the cooperative lean evaluation
        variant = ft.glibc_build_of_this_host()
        gpu.set_option("math", variant); oracle.lib.orc_set_libm(1)
        try:
            ds, os_ = both(gpu, oracle, syn.config3(n=256, size=64)[0])
            g, _ = ds.render(EPS, LEN, ft.ImageSize(64, 64), cam)
            o, _ = os_.render(EPS, LEN, 64, 64, cam.as_array())
            assert_bit_equal(g, o, f"C3 in glibc mode, tail_k = {k}")
        finally:
            gpu.set_option("math", 0); oracle.lib.orc_set_libm(0)
    finally:
        gpu.set_option("tail_k", -1)


def test_hand_out_options_do_not_change_the_frame(gpu, oracle):
    """FT_OPT_CHUNK (rays per grab: whole, half, quarter tiles) and FT_OPT_GUIDED (shrinking grabs at the end of the queue) only change which
    wave renders which pixel: the frame and the counters stay those of the oracle"""
    cam = syn.default_camera()
    scene, _ = syn.config3(n=64, size=200)
    ds, os_ = both(gpu, oracle, scene)
    want, ocnt = os_.render(EPS, LEN, 200, 136, cam.as_array())
    try:
        for chunk, guided, k in ((64, 1, 32), (32, 0, 32), (16, 0, 32), (32, 1, 0), (16, 0, 2), (64, 0, -1)):
            gpu.set_option("chunk", chunk); gpu.set_option("guided", guided); gpu.set_option("tail_k", k)
            g, gst = ds.render(EPS, LEN, ft.ImageSize(200, 136), cam)
            assert_bit_equal(g, want, f"chunk {chunk}, guided {guided}, tail_k {k}")
            check_counts(gst, ocnt)
        c2, _ = syn.config2(seed=9, size=96)
        ds2, os2 = both(gpu, oracle, c2)
        want2, _ = os2.render(EPS, LEN, 96, 96, cam.as_array())
        for chunk in (32, 16):
            gpu.set_option("chunk", chunk)
            assert_bit_equal(ds2.render(EPS, LEN, ft.ImageSize(96, 96), cam)[0], want2, f"C2, chunk {chunk}")
    finally:
        gpu.set_option("chunk", 64); gpu.set_option("guided", 0); gpu.set_option("tail_k", -1)


def test_child_culling_is_exact_and_happens(gpu, oracle):
    """FT_OPT_CULL (lean kernel): per wave and round, children whose terms are below half an ulp of the running sum in every ray of the
    wave are not evaluated.  The frame and the counters are the oracle's with the pass on and off; with it on a sizeable share of the
    (child, ray) pairs is dropped, with it off none — for several strengths, child counts beyond one culling row, a second (non-culled)
    run's worth of children, both arithmetics, and with the latency mode forced on (which bypasses the pass)."""
    cam = syn.default_camera()
    try:
        # the pass needs the rays of a wave (one 8x8 tile) close together, i.e. a fine pixel pitch = 1 / max(W, H): wide, low frames
        for n, W, H, strength in ((256, 3072, 24, 0.25), (300, 2048, 16, 0.5), (64, 1024, 24, 0.1), (40, 64, 56, 1.0), (256, 96, 88, 0.3)):
            scene, _ = syn.config3(n=n, size=W, strength=strength)
            ds, os_ = both(gpu, oracle, scene)
            want, ocnt = os_.render(EPS, LEN, W, H, cam.as_array())
            for cull in (1, 0):
                gpu.set_option("cull", cull)
                g, gst = ds.render(EPS, LEN, ft.ImageSize(W, H), cam)
                assert_bit_equal(g, want, f"{n} spheres, strength {strength}, cull {cull}")
                check_counts(gst, ocnt)
                if cull == 0: assert gst["culled_fraction"] == 0.0
                elif W >= 1024 and strength <= 0.25: assert gst["culled_fraction"] > 0.05, gst   # (a softer union decays too slowly to drop anything in a ball of radius 4)
        gpu.set_option("cull", 1); gpu.set_option("tail_k", 64)
        scene, _ = syn.config3(n=256, size=2048)
        ds, os_ = both(gpu, oracle, scene)
        g, gst = ds.render(EPS, LEN, ft.ImageSize(2048, 16), cam)
        assert_bit_equal(g, os_.render(EPS, LEN, 2048, 16, cam.as_array())[0], "latency mode forced: no culling pass")
        assert gst["culled_fraction"] == 0.0
        gpu.set_option("tail_k", -1); gpu.set_option("math", ft.glibc_build_of_this_host()); oracle.lib.orc_set_libm(1)
        try:
            g, gst = ds.render(EPS, LEN, ft.ImageSize(2048, 16), cam)
            assert_bit_equal(g, os_.render(EPS, LEN, 2048, 16, cam.as_array())[0], "culling in glibc mode")
            assert gst["culled_fraction"] > 0.05, gst
        finally:
            gpu.set_option("math", 0); oracle.lib.orc_set_libm(0)
    finally:
        gpu.set_option("cull", 1); gpu.set_option("tail_k", -1)


def test_child_culling_in_the_general_kernels(gpu, oracle):
    """Round 4: the culling pass also serves the longest staged sphere run of a general scene's main program (FtSceneDev.cullPc) — a smooth union
    nested in a union, in an intersect, behind children of another kind (the run then continues an accumulator), a glass blob (EXTENSION build),
    a union with on-demand children next to it (calls kernel).  Frames and counters are the oracle's with the pass on and off, and where the rays
    of a wave are close together a share of the (child, ray) pairs is really dropped."""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    P = SdfForm.Primitive
    from fraytracer_amd import Camera, Lens
    # a wide, low frame is the first rows of a square one, i.e. the lower edge of the view: tilt the camera so that this strip crosses the scene's middle
    cam = Camera.lookAt(Position=(0.0, 0.0, -10.0), LookAt=(0.0, -4.9, 0.0), Up=(0.0, 1.0, 0.0), Lens=Lens.create(60.0))   # (Lens.create 60 is sin(30 rad) < 0: row 0 looks up)
    rng = syn.Rng(61)
    mat = [SdfMaterial.createSolid((0.2 + 0.15 * i, 0.8 - 0.1 * i, 0.5)) for i in range(4)]
    balls = lambda n, spread=3.0: [P.sphere(rng.pointInBall(spread), rng.range(0.15, 0.5)) for _ in range(n)]
    blob = SdfForm.unionSmooth(0.2, balls(200))
    in_union = SdfObject.union([SdfObject.create(mat[0], blob), SdfObject.create(mat[1], P.torus((0.0, -3.0, 0.0), (0.0, 1.0, 0.0), 1.5, 0.3)),
                                SdfObject.create(mat[2], P.capsule((-3.5, 0.0, 0.0), (-3.0, 2.0, 1.0), 0.4))])
    in_intersect = SdfObject.create(mat[1], SdfForm.intersect([SdfForm.unionSmooth(0.25, balls(120)), P.sphere((0.0, 0.0, 0.0), 2.6)]))
    behind_others = SdfObject.create(mat[2], SdfForm.unionSmooth(0.25, [P.capsule((-1.0, 0.0, 0.0), (1.0, 0.5, 0.0), 0.3), P.torus((0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 1.0, 0.2)] + balls(90)))
    glass = SdfObject.create(SdfMaterial.createGlass((0.95, 0.9, 0.8), 1.45, 0.03), SdfForm.unionSmooth(0.25, balls(48, 2.0)))
    crowd = SdfObject.union([SdfObject.create(mat[0], SdfForm.unionSmooth(0.2, balls(64, 2.0)))] +
                            [SdfObject.create(mat[i % 4], SdfForm.subtract(P.sphere(c, 0.5), P.sphere(c + rng.pointOnSphere(0.3), 0.3))) for i, c in
                             enumerate(rng.pointOnSphere(3.5) for _ in range(12))])
    cases = [("smooth union in a union", in_union, {}, True), ("smooth union in an intersect", in_intersect, {}, True), ("run behind other kinds", behind_others, {}, True),
             ("glass blob (EXTENSION)", glass, dict(spp=4, spectral=4, max_bounces=4), False), ("smooth union beside on-demand children", crowd, {}, False)]
    try:
        for name, obj, kw, must_cull in cases:
            scene = SdfScene(obj, syn.BACKGROUND, syn.program_lights())
            ds, os_ = both(gpu, oracle, scene)
            assert ds.info()["fast_path"] in ((0, 2) if not kw else (0, 1, 2)), name        # (the glass blob alone is a lean scene: its EXTENSION build culls too)
            for W, H in ((1536, 16), (96, 64)):
                want, ocnt = os_.render(EPS, LEN, W, H, cam.as_array(), **kw)
                for cull in (1, 0):
                    gpu.set_option("cull", cull)
                    g, gst = ds.render(EPS, LEN, ft.ImageSize(W, H), cam, **kw)
                    assert_bit_equal(g, want, f"{name} {W}x{H}, cull {cull}")
                    check_counts(gst, ocnt)
                    if cull == 0: assert gst["culled_fraction"] == 0.0
                    elif must_cull and W >= 1024: assert gst["culled_fraction"] > 0.005, (name, gst["culled_fraction"])   # (how much depends on how the strip cuts the blob: 2 - 30 %)
            ds.close()
    finally:
        gpu.set_option("cull", 1)


def test_escape_shortcut_changes_no_pixel(gpu, oracle):
    """FT_OPT_ESCAPE: a ray that can no longer come within epsilon of the scene's support sphere ends as a miss at once.  Frames, ray and hit
    counters and flags are the oracle's with the shortcut on and off; with it on fewer evaluations are spent.  Scenes of every kernel variant,
    the reference's two light types, explicit rays that start outside, inside, tangent to and pointing away from the sphere, and the
    tryTrace entries (ValueNone on a miss)."""
    cam = syn.default_camera()
    cases = [("C3 lean", syn.config3(n=64, size=256)[0], 256, 64), ("C2", syn.config2(seed=4, size=128)[0], 128, 96),
             ("Program.fs structure", syn.console_scene(n=120, size=160)[0], 160, 120), ("mixed nested", syn.mixed_nested()[0], 96, 96),
             ("combinator zoo (calls)", syn.combinator_zoo()[0], 96, 80)]
    try:
        for name, scene, W, H in cases:
            ds, os_ = both(gpu, oracle, scene)
            want, ocnt = os_.render(EPS, LEN, W, H, cam.as_array())
            evals = {}
            for esc in (1, 0):
                gpu.set_option("escape", esc)
                g, gst = ds.render(EPS, LEN, ft.ImageSize(W, H), cam)
                assert_bit_equal(g, want, f"{name}, escape {esc}")
                check_counts(gst, ocnt)
                evals[esc] = gst["sdf_evals"]
            assert evals[1] < evals[0], (name, evals)
            if name == "C2":                                           # EXTENSIONS on the same scene: AO rays (not shortened) and 4 samples per pixel
                kw = dict(spp=4, ao_samples=3, ao_radius=0.7)
                want_x = os_.render(EPS, LEN, W, H, cam.as_array(), **kw)[0]
                for esc in (1, 0):
                    gpu.set_option("escape", esc)
                    assert_bit_equal(ds.render(EPS, LEN, ft.ImageSize(W, H), cam, **kw)[0], want_x, f"{name} with AO and 4 spp, escape {esc}")
            cx, cy, cz, R = ds.support_sphere()
            assert R > 0
            # explicit rays around the support sphere
            rng = np.random.default_rng(11)
            n = 400
            o = (np.array([cx, cy, cz]) + rng.normal(size=(n, 3)) * R * rng.uniform(0.2, 3.0, (n, 1))).astype(np.float32)
            d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
            d[:50] = (np.array([cx, cy, cz], np.float32) - o[:50]); d[:50] /= np.linalg.norm(d[:50], axis=1, keepdims=True)   # straight at the centre
            d[50:100] = -d[:50]                                                                                               # ... and straight away (origins differ)
            d[100:110] *= np.float32(0.0)                                                                                     # a ray that does not move
            d[110:120] *= np.float32(37.5)                                                                                    # non-unit directions
            rays = np.concatenate([o, d, rng.uniform(0.5, 60.0, (n, 1)).astype(np.float32), np.full((n, 1), 0.01, np.float32)], axis=1).astype(np.float32)
            for esc in (1, 0):
                gpu.set_option("escape", esc)
                with np.errstate(all="ignore"):
                    assert_bit_equal(ds.trace_rays(rays)[0], os_.trace_rays(rays)[0], f"{name}: ray buffer, escape {esc}")
                    gf, of = ds.form_try_trace(rays), os_.form_try_trace(rays)
                    assert_bit_equal(gf[0] if isinstance(gf, tuple) else gf, of[0] if isinstance(of, tuple) else of, f"{name}: SdfForm.tryTrace, escape {esc}")
        # EXTENSION glass: a path inside a body marches on -Distance (below epsilon everywhere outside the support sphere): only paths
        # outside bodies may take the shortcut (a fuzz scene with 7 bounces found the difference)
        scene5 = syn.config5(size=96)[0]
        ds5, os5 = both(gpu, oracle, scene5)
        kw = dict(spp=4, spectral=4, max_bounces=6)
        want5, ocnt5 = os5.render(EPS, LEN, 96, 96, cam.as_array(), **kw)
        for esc in (1, 0):
            gpu.set_option("escape", esc)
            g5, gst5 = ds5.render(EPS, LEN, ft.ImageSize(96, 96), cam, **kw)
            assert_bit_equal(g5, want5, f"glass, escape {esc}")
            assert gst5["rays_ext"] == ocnt5["rays_ext"] and gst5["hits_primary"] == ocnt5["hits_primary"]
    finally:
        gpu.set_option("escape", 1)


def test_lazy_union_under_intersect_changes_no_pixel(gpu, oracle):
    """FT_OPT_LAZY_UNION: a union that is child 0 of an intersect stops its walk at Items.[0] wherever that distance is already <= the next
    child's (which then decides the intersect).  Frames, counters, explicit rays and the materials SdfObject.tryTrace reports are the
    oracle's with the option on and off — for the reference's own structure, for a non-sphere second child, for a union whose objects carry
    different materials (the material of a shortened walk must never be asked for), under subtract and inside another union."""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    cam = syn.default_camera()
    P = SdfForm.Primitive
    rng = syn.Rng(5)
    mats = [SdfMaterial.createSolid((0.2 + 0.1 * i, 0.9 - 0.1 * i, 0.5)) for i in range(6)]
    blobs = [SdfObject.create(mats[i % 6], P.sphere(rng.pointInBall(3.0), rng.range(0.3, 0.9))) for i in range(40)]
    tori = [SdfObject.create(mats[i % 6], P.torus(rng.pointInBall(3.0), rng.pointOnSphere(1.0), rng.range(0.4, 0.8), rng.range(0.1, 0.25))) for i in range(30)]
    clipped = SdfObject.intersect(SdfObject.union(blobs), [P.sphere((0.2, 0.1, 0.0), 2.2)])
    by_torus = SdfObject.intersect(SdfObject.union(tori), [P.torus((0, 0, 0), (0, 1, 0), 2.0, 1.2), P.sphere((0, 0, 0), 3.0)])
    carved = SdfObject.subtract(SdfObject.intersect(SdfObject.union(blobs[:20] + tori[:10]), [P.sphere((0, 0, 0), 2.5)]), P.sphere((-0.5, 1.0, -2.0), 1.5))
    nested = SdfObject.union([SdfObject.intersect(SdfObject.union(blobs[:12]), [P.sphere((-1.5, 0, 0), 1.6)]),
                              SdfObject.intersect(SdfObject.union(tori[:12]), [P.capsule((1.0, -1.0, 0), (2.0, 1.0, 0.5), 1.4)]),
                              SdfObject.create(mats[0], P.sphere((0, -2.5, 0), 0.6))])
    holed = SdfObject.subtract(SdfObject.union(blobs[:25]), P.sphere((0.3, 0.2, -1.0), 1.8))                  # a union directly under subtract
    form_level = SdfObject.create(mats[2], SdfForm.subtract(SdfForm.intersect([SdfForm.union([P.sphere(rng.pointInBall(2.5), rng.range(0.3, 0.7)) for _ in range(30)]),
                                                                                   P.sphere((0, 0, 0), 2.0)]), P.capsule((-1, -1, -2), (1, 1, -2), 1.0)))
    cases = [("Program.fs structure", syn.console_scene(n=150, size=200)[0]), ("clipped blobs", SdfScene(clipped, syn.BACKGROUND, syn.program_lights())),
             ("union under subtract", SdfScene(holed, syn.BACKGROUND, syn.program_lights())), ("form-level subtract(intersect(union, sphere), capsule)", SdfScene(form_level, syn.BACKGROUND, syn.program_lights())),
             ("torus as second child", SdfScene(by_torus, syn.BACKGROUND, syn.program_lights())), ("carved", SdfScene(carved, syn.BACKGROUND, syn.program_lights())),
             ("intersects inside a union", SdfScene(nested, syn.BACKGROUND, syn.program_lights()))]
    try:
        for name, scene in cases:
            ds, os_ = both(gpu, oracle, scene)
            W, H = 200, 136
            want, ocnt = os_.render(EPS, LEN, W, H, cam.as_array())
            rr = np.random.default_rng(3)
            o = (rr.normal(size=(500, 3)) * 3.0).astype(np.float32)
            d = rr.normal(size=(500, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
            rays = np.concatenate([o, d, np.full((500, 1), 30, np.float32), rr.choice([0.01, 0.3, 1.5], (500, 1)).astype(np.float32)], axis=1).astype(np.float32)
            want_obj = os_.object_try_trace(rays)
            want_obj = want_obj[0] if isinstance(want_obj, tuple) else want_obj
            for lazy in (1, 0):
                gpu.set_option("lazy_union", lazy)
                g, gst = ds.render(EPS, LEN, ft.ImageSize(W, H), cam)
                assert_bit_equal(g, want, f"{name}, lazy_union {lazy}")
                check_counts(gst, ocnt)
                got_obj = ds.object_try_trace(rays)
                assert_bit_equal(got_obj[0] if isinstance(got_obj, tuple) else got_obj, want_obj, f"{name}: SdfObject.tryTrace (materials), lazy_union {lazy}")
                with np.errstate(all="ignore"):
                    assert_bit_equal(ds.trace_rays(rays)[0], os_.trace_rays(rays)[0], f"{name}: ray buffer, lazy_union {lazy}")
    finally:
        gpu.set_option("lazy_union", 1)


def test_culling_on_random_smooth_unions(gpu, oracle):
    """a short run of tools/fuzz_cull.py inside the suite: random smooth unions of spheres (counts, strengths, radii, extents over two decades) seen by
    random cameras, from inside the cloud to far outside, on wide low frames where the culling pass drops children — float for float against the oracle"""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene, SdfLight, Camera, Lens
    dropped = 0
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(32, 301))
        spread = float(10.0 ** rng.uniform(-0.5, 1.0))
        strength = float(10.0 ** rng.uniform(-1.5, 0.18))
        rmax = float(spread * 10.0 ** rng.uniform(-1.5, -0.5))
        c = rng.normal(size=(n, 3)) * spread * 0.5
        forms = [SdfForm.Primitive.sphere(tuple(float(v) for v in c[i]), float(rng.uniform(0.2, 1.0) * rmax)) for i in range(n)]
        lights = [SdfLight.directional(tuple(float(v) for v in rng.normal(size=3)), (0.5, 0.5, 0.5))]
        scene = SdfScene(SdfObject.create(SdfMaterial.createSolid((0.9, 0.6, 0.3)), SdfForm.unionSmooth(strength, forms)), syn.BACKGROUND, lights)
        pos = rng.normal(size=3); pos = pos / np.linalg.norm(pos) * spread * float(10.0 ** rng.uniform(-0.7, 0.8))
        cam = Camera.lookAt(Position=tuple(float(v) for v in pos), LookAt=tuple(float(v) for v in rng.normal(size=3) * spread * 0.2), Up=(0.0, 1.0, 0.0),
                            Lens=Lens.create(float(rng.uniform(20.0, 90.0))))
        W, H = int(rng.choice([1024, 2048])), 8
        eps, length = float(10.0 ** rng.uniform(-3.0, -1.5)), float(spread * rng.uniform(2.0, 40.0))
        ds, os_ = both(gpu, oracle, scene)
        g, gst = ds.render(eps, length, ft.ImageSize(W, H), cam)
        o, ocnt = os_.render(eps, length, W, H, cam.as_array())
        assert_bit_equal(g, o, f"culling fuzz seed {1000 + seed}: {n} spheres, strength {strength:.3g}, extent {spread:.3g}")
        check_counts(gst, ocnt)
        dropped += gst["culled_fraction"] > 0.1
    assert dropped >= 5, dropped                                       # the pass really ran on a fair share of them


def test_c1_single_sphere(gpu, oracle):
    scene, size = syn.config1()
    g, gst, o, ocnt = render_both(gpu, oracle, scene, size.X, size.Y)
    assert_bit_equal(g, o, "C1 image")
    check_counts(gst, ocnt)
    # known answers: centre pixel hits after steps 9 -> 0; no lights -> Color * (bg * 1/pi)
    want = np.float32(0.8) * (np.float32(0.1) * (np.float32(1) / np.float32(3.14159274)))
    assert g[128, 128, 0] == want and g[0, 0, 0] == np.float32(0.1)


@pytest.mark.parametrize("boxes", [False, True])
def test_c2_union_of_32(gpu, oracle, boxes):
    scene, _ = syn.config2(boxes=boxes)
    g, gst, o, ocnt = render_both(gpu, oracle, scene, 256, 256)
    assert_bit_equal(g, o, "C2 image")
    check_counts(gst, ocnt)
    assert gst["rays_shadow"] > 0


def test_c3_smooth_union_256(gpu, oracle):
    scene, _ = syn.config3()
    g, gst, o, ocnt = render_both(gpu, oracle, scene, 128, 128)
    assert_bit_equal(g, o, "C3 image")
    check_counts(gst, ocnt)


@pytest.mark.parametrize("strength", [0.25, 0.5, 2.0, 0.3, 0.125])
def test_smooth_union_strength_variants_of_the_near_loop(gpu, oracle, strength):
    """-1/strength = -4, -2, -1/2 take the near loop whose product with the strength rides on the subtraction's output modifier
    (kernels.hip ft_strength_times_diff); 0.3 and 0.125 (-8: no modifier for it) take the general product — all against the oracle,
    on the lean kernel and with 4 samples per pixel on its EXTENSION build"""
    scene, _ = syn.config3(n=67, strength=strength)             # 67 = 16 trips of four and a remainder of three
    g, gst, o, ocnt = render_both(gpu, oracle, scene, 96, 96)
    assert_bit_equal(g, o, f"unionSmooth {strength}")
    check_counts(gst, ocnt)
    cam = syn.default_camera()
    ds, os_ = both(gpu, oracle, scene)
    assert ds.info()["fast_path"] == 1
    g4, _ = ds.render(EPS, LEN, ft.ImageSize(48, 48), cam, spp=4)
    o4, _ = os_.render(EPS, LEN, 48, 48, cam.as_array(), spp=4)
    assert_bit_equal(g4, o4, f"unionSmooth {strength}, 4 spp")


def test_console_like_scene(gpu, oracle):
    scene, _ = syn.console_like(n=300)
    g, gst, o, ocnt = render_both(gpu, oracle, scene, 160, 160)
    assert_bit_equal(g, o, "console-like image")
    check_counts(gst, ocnt)


@pytest.mark.parametrize("factory", [syn.random_sphere, syn.random_capsule, syn.random_torus, syn.random_triangle, syn.random_box])
def test_console_structure_per_primitive(gpu, oracle, factory):
    scene, _ = syn.console_like(seed=5, n=60, factory=factory)
    g, gst, o, ocnt = render_both(gpu, oracle, scene, 96, 96)
    assert_bit_equal(g, o, factory.__name__)
    check_counts(gst, ocnt)


def test_mixed_nested_scene(gpu, oracle):
    scene, size = syn.mixed_nested()
    g, gst, o, ocnt = render_both(gpu, oracle, scene, size.X, size.Y)
    assert_bit_equal(g, o, "nested image")
    check_counts(gst, ocnt)


def test_combinator_zoo(gpu, oracle):
    scene, size = syn.combinator_zoo()
    g, gst, o, ocnt = render_both(gpu, oracle, scene, size.X, size.Y)
    assert_bit_equal(g, o, "zoo image")
    check_counts(gst, ocnt)
    assert gst["hits_primary"] > 500 and gst["rays_shadow"] > 500
    rng = np.random.default_rng(2)
    pts = rng.uniform(-4, 4, (20000, 3)).astype(np.float32)
    ds, os_ = both(gpu, oracle, scene)
    O = oracle.Oracle()
    assert_bit_equal(ds.eval_distance(pts)[0], O.form_distance(O.object_form(os_.object), pts), "zoo distance")


def test_non_square_and_ragged_sizes(gpu, oracle):
    scene, _ = syn.config2(seed=11)
    for W, H in [(37, 101), (130, 19), (1, 1), (8, 9)]:
        g, gst, o, ocnt = render_both(gpu, oracle, scene, W, H)
        assert_bit_equal(g, o, f"{W}x{H}")
        check_counts(gst, ocnt)


def test_eval_distance_and_material(gpu, oracle):
    rng = np.random.default_rng(3)
    pts = rng.uniform(-6, 6, (20000, 3)).astype(np.float32)
    for scene in (syn.config2(seed=4)[0], syn.config3(n=40)[0], syn.console_like(n=120)[0], syn.mixed_nested()[0]):
        ds, os_ = both(gpu, oracle, scene)
        d, m = ds.eval_distance(pts)
        O = oracle.Oracle()
        want = O.form_distance(O.object_form(os_.object), pts)
        assert_bit_equal(d, want, "Form.Distance")


def test_trace_rays_matches_oracle_and_render(gpu, oracle):
    scene, _ = syn.console_like(n=150)
    ds, os_ = both(gpu, oracle, scene)
    cam = syn.default_camera()
    W = H = 64
    rays = np.stack([oracle.pixel_ray(cam.as_array(), W, H, x, y, EPS, LEN) for x in range(W) for y in range(H)])
    g, gst = ds.trace_rays(rays)
    o, ocnt = os_.trace_rays(rays)
    assert_bit_equal(g, o, "trace_rays")
    img, _ = ds.render(EPS, LEN, ft.ImageSize(W, H), cam)
    assert_bit_equal(g.reshape(W, H, 3), img, "ray buffer vs Image.render")
    # arbitrary rays: random origins/directions, per-ray epsilon and length, zero-length ray -> background
    rng = np.random.default_rng(9)
    n = 5000
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = rng.uniform(-8, 8, (n, 3))
    d = rng.normal(size=(n, 3)); r[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    r[:, 6] = rng.uniform(0, 40, n); r[:, 7] = rng.choice([0.01, 0.003, 0.05], n)
    r[0, 6] = 0.0; r[1, 6] = -1.0
    g, _ = ds.trace_rays(r)
    o, _ = os_.trace_rays(r)
    assert_bit_equal(g, o, "arbitrary rays")
    assert tuple(g[0]) == tuple(np.float32(syn.BACKGROUND))


def test_try_trace_entries_match_oracle(gpu, oracle):
    """ft_form_try_trace / ft_object_try_trace = SdfForm.tryTrace / SdfObject.tryTrace (SdfForm.fs:93-104,
    SdfObject.fs:66-78) over a ray buffer: hit rays, distances, normals, colours and misses, bit for bit"""
    rng = np.random.default_rng(13)
    cam = syn.default_camera()
    for scene in (syn.console_like(n=150)[0], syn.config3(n=40)[0], syn.mixed_nested()[0], syn.config5(size=64)[0]):
        ds, os_ = both(gpu, oracle, scene)
        W = H = 48
        rays = [oracle.pixel_ray(cam.as_array(), W, H, x, y, EPS, LEN) for x in range(W) for y in range(H)]
        n = 3000
        r = np.zeros((n, 8), np.float32)
        r[:, 0:3] = rng.uniform(-8, 8, (n, 3))
        d = rng.normal(size=(n, 3)); r[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
        r[:, 6] = rng.uniform(0, 40, n); r[:, 7] = rng.choice([0.01, 0.003, 0.05], n)
        r[0, 6] = 0.0; r[1, 6] = -1.0
        rays = np.concatenate([np.stack(rays), r])
        gf, gst = ds.form_try_trace(rays)
        of, ocnt = os_.form_try_trace(rays)
        assert_bit_equal(gf, of, "SdfForm.tryTrace")
        assert gst["hits_primary"] == int(gf[:, 9].view(np.int32).sum()) > 100 and gst["rays_primary"] == len(rays)
        go, gst = ds.object_try_trace(rays)
        oo, ocnt = os_.object_try_trace(rays)
        assert_bit_equal(go, oo, "SdfObject.tryTrace")
        hit = go[:, 14].view(np.int32) == 1
        assert gst["hits_primary"] == int(hit.sum()) and gst["rays_shadow"] == 0
        assert_bit_equal(go[hit, 3:6], rays[hit, 3:6], "direction passes through")
        assert not go[~hit].any()
    assert ds.form_try_trace(np.zeros((0, 8), np.float32))[0].shape == (0, 10)
    # the mirrors of the F# functions: SdfForm.tryTrace sdf ray / SdfObject.tryTrace object ray
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial
    form = SdfForm.unionSmooth(0.3, [SdfForm.Primitive.sphere((0.5, 0, 0), 1.0), SdfForm.Primitive.torus((0, 0, 0), (0, 1, 0), 2.0, 0.3)])
    obj = SdfObject.create(SdfMaterial.createSolid((0.2, 0.4, 0.6)), form)
    O = oracle.Oracle()
    want_f, _ = O.scene(ft.SdfScene(obj, (0, 0, 0), [])).form_try_trace(rays)
    want_o, _ = O.scene(ft.SdfScene(obj, (0, 0, 0), [])).object_try_trace(rays)
    assert_bit_equal(SdfForm.tryTrace(form, rays, gpu), want_f, "SdfForm.tryTrace mirror")
    assert_bit_equal(SdfObject.tryTrace(obj, rays, gpu), want_o, "SdfObject.tryTrace mirror")
    assert_bit_equal(SdfObject.tryTrace(obj, rays[:10], gpu), want_o[:10], "cached scene")
    # SdfForm.normalFromRay at the hit rays = the Normal of SdfObject.tryTrace; tryDistance = Distance inside the boundary
    hit = want_f[:, 9].view(np.int32) == 1
    assert_bit_equal(SdfForm.normalFromRay(form, want_f[hit, :8], gpu), want_o[hit, 8:11], "SdfForm.normalFromRay mirror")
    pts = rng.uniform(-4, 4, (2000, 3)).astype(np.float32)
    fh = O.object_form(ft.realise(obj, O))
    want_d = O.form_distance(fh, pts)
    assert_bit_equal(SdfForm.distance(form, pts, gpu), want_d, "sdf.Distance mirror")
    b = np.asarray(O.form_boundary(fh), np.float32)
    dd = pts - b[0:3]
    inside = ((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]) < b[3] * b[3]
    got = SdfForm.tryDistance(form, pts, gpu)
    assert 0 < inside.sum() < len(pts) and np.isnan(got[~inside]).all()
    assert_bit_equal(got[inside], want_d[inside], "SdfForm.tryDistance mirror")


def test_column_tiles_concatenate_to_full_frame(gpu):
    scene, _ = syn.config2(seed=6)
    cam = syn.default_camera()
    ds = gpu.scene(scene)
    W, H = 192, 80
    full, _ = ds.render(EPS, LEN, ft.ImageSize(W, H), cam)
    # contiguous tiles
    parts = [ds.render(EPS, LEN, ft.ImageSize(W, H), cam, x0=x0, n_columns=48)[0] for x0 in range(0, W, 48)]
    assert_bit_equal(np.concatenate(parts, 0), full, "contiguous tiles")
    # interleaved stripes of 16 columns over 4 ranks
    S, R = 16, 4
    out = np.empty_like(full)
    for r in range(R):
        slab, _ = ds.render(EPS, LEN, ft.ImageSize(W, H), cam, stripe_width=S, stripe_ranks=R, stripe_rank=r)
        for j in range(slab.shape[0] // S):
            out[(j * R + r) * S:(j * R + r + 1) * S] = slab[j * S:(j + 1) * S]
    assert_bit_equal(out, full, "striped tiles")


def test_host_output_pipeline_is_bit_identical(gpu, oracle):
    """ft_render delivers the frame in host memory (Image.render returns a host FColor[,], Image.fs:26-35): large frames are
    rendered in column chunks on two streams while finished chunks are copied into the page-locked destination.  Registered,
    unregistered (pinned inside the call) and small (single-launch) frames all equal the frame left in HBM, bit for bit."""
    scene, _ = syn.config2(seed=6, size=1536)
    cam = syn.default_camera()
    ds = gpu.scene(scene)
    for (W, H) in [(1536, 1024), (2048, 2048), (300, 5000), (64, 64)]:
        S = ft.ImageSize(W, H)
        gpu.set_option("host_chunks", 1); gpu.set_option("host_pin", 0)            # one launch, one pageable copy after it
        try:
            want, st0 = ds.render(EPS, LEN, S, cam)
        finally:
            gpu.set_option("host_chunks", 0); gpu.set_option("host_pin", 1)
        got, st = ds.render(EPS, LEN, S, cam)                      # fresh pageable array: pinned inside the call
        assert_bit_equal(got, want, f"{W}x{H} pageable")
        for k in ("rays_primary", "rays_shadow", "hits_primary", "hits_shadow", "sdf_evals"):
            assert st[k] == st0[k], k
        out = np.zeros((W, H, 3), np.float32)
        gpu.host_register(out)
        try:
            got2, _ = ds.render(EPS, LEN, S, cam, out=out)
            assert got2 is out
            assert_bit_equal(out, want, f"{W}x{H} registered")
            part, _ = ds.render(EPS, LEN, S, cam, out=out[:W // 2] if W >= 128 else None, x0=0, n_columns=W // 2) if W >= 128 else (None, None)
            if part is not None:
                assert_bit_equal(part, want[:W // 2], "a column range into the front of the registered buffer")
        finally:
            gpu.host_unregister(out)
    with pytest.raises(ft.FrayTracerError):
        gpu.host_unregister(out)                                   # not registered any more
    with pytest.raises(ValueError):
        ds.render(EPS, LEN, ft.ImageSize(64, 64), cam, out=np.zeros((64, 64, 4), np.float32))
    # against the oracle, through the chunked path
    full, _ = ds.render(EPS, LEN, ft.ImageSize(1536, 1536), cam)
    o, _ = oracle.Oracle().scene(scene).render(EPS, LEN, 1536, 1536, cam.as_array(), xstep=64)
    assert_bit_equal(full[::64], o, "chunked host output against the oracle")


def test_render_multi_single_process_path(gpu):
    """ft_render_multi with the one GPU this box has: stripes + (degenerate) gather + de-interleaving copy
    must reproduce the monolithic render; two contexts on the same device exercise ft_scene_clone and the
    thread-per-device path without a second GPU (the RCCL gather itself needs >= 2 devices)."""
    scene, _ = syn.config2(seed=8)
    cam = syn.default_camera()
    W, H = 128, 72
    full, st = gpu.scene(scene).render(EPS, LEN, ft.ImageSize(W, H), cam)
    img, st1 = ft.render_multi([gpu], scene, EPS, LEN, ft.ImageSize(W, H), cam, stripe_width=16)
    assert_bit_equal(img, full, "ft_render_multi n=1")
    assert st1["rays_primary"] == st["rays_primary"] and st1["rays_shadow"] == st["rays_shadow"]
    # n = 2 and 4 contexts on the one GPU: thread-per-context, interleaved stripes, slab collection and
    # de-interleaving all run (the collection itself degenerates to device copies: RCCL refuses duplicate GPUs)
    extra = [ft.Device(0) for _ in range(3)]
    try:
        for devs in ([gpu, extra[0]], [gpu] + extra):
            imgn, stn = ft.render_multi(devs, scene, EPS, LEN, ft.ImageSize(W, H), cam, stripe_width=8)
            assert_bit_equal(imgn, full, f"ft_render_multi n={len(devs)} on one device")
            assert stn["rays_primary"] == st["rays_primary"] and stn["rays_shadow"] == st["rays_shadow"]
    finally:
        for d in extra:
            d.close()
    other = ft.Device(0)
    try:
        p = ft.api.C.c_void_p()
        ft.api.check(ft.api.lib.ft_scene_clone(gpu.scene(scene)._scene, other._ctx, ft.api.C.byref(p)))
        clone = ft.DeviceScene.__new__(ft.DeviceScene); clone.device, clone._scene = other, p
        img2, _ = clone.render(EPS, LEN, ft.ImageSize(W, H), cam)
        assert_bit_equal(img2, full, "cloned scene on a second context")
        clone.close()
    finally:
        other.close()


# ---- EXTENSIONS (no reference counterpart; checked against the oracle's own definition only) ------------
@pytest.mark.parametrize("spp,ao", [(4, 0), (1, 8), (4, 16), (9, 3)])
def test_extension_spp_and_ambient_occlusion(gpu, oracle, spp, ao):
    cam = syn.default_camera()
    for scene, n in ((syn.config2(boxes=True)[0], 96), (syn.config3(n=48)[0], 64)):
        ds, os_ = both(gpu, oracle, scene)
        g, gst = ds.render(EPS, LEN, ft.ImageSize(n, n), cam, spp=spp, ao_samples=ao, ao_radius=0.75)
        o, ocnt = os_.render(EPS, LEN, n, n, cam.as_array(), spp=spp, ao_samples=ao, ao_radius=0.75)
        assert_bit_equal(g, o, f"extension spp={spp} ao={ao}")
        assert gst["rays_primary"] == ocnt["rays_primary"] == n * n * spp
        assert gst["rays_ext"] == ocnt["rays_ext"] and (ao == 0 or gst["rays_ext"] > 0)
        assert gst["rays_shadow"] == ocnt["rays_shadow"]


def test_extension_defaults_are_the_reference_path(gpu):
    scene, _ = syn.config2(seed=12)
    cam = syn.default_camera()
    ds = gpu.scene(scene)
    a, _ = ds.render(EPS, LEN, ft.ImageSize(80, 80), cam)
    b, _ = ds.render(EPS, LEN, ft.ImageSize(80, 80), cam, spp=1, ao_samples=0, ao_radius=3.0)
    assert_bit_equal(a, b, "spp=1, ao=0")
    with pytest.raises(ft.FrayTracerError):
        ds.render(EPS, LEN, ft.ImageSize(8, 8), cam, spp=3)


def glass_blob_scene(n=24, seed=31, disp=0.03):
    """one glass object that is a smooth union of spheres (-> the shape-specialised kernel's EXTENSION build)"""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    rng = syn.Rng(seed)
    kids = [SdfForm.Primitive.sphere(rng.pointInBall(2.5), rng.range(0.4, 0.9)) for _ in range(n)]
    obj = SdfObject.create(SdfMaterial.createGlass((0.95, 0.9, 0.8), 1.45, disp), SdfForm.unionSmooth(0.25, kids))
    return SdfScene(obj, syn.BACKGROUND, syn.program_lights())


@pytest.mark.parametrize("spp,spectral,bounces,ao", [(1, 0, 4, 0), (4, 4, 4, 0), (16, 16, 4, 0), (4, 2, 1, 0), (4, 4, 6, 4), (1, 1, 2, 0)])
def test_extension_glass_paths(gpu, oracle, spp, spectral, bounces, ao):
    """EXTENSION (BASELINE.json config 5): refraction / reflection / wavelength bins, defined by the oracle
    (tests/test_oracle_glass_ext.py pins that definition); the kernel must reproduce it bit for bit, ray for ray."""
    cam = syn.default_camera()
    for scene, n in ((syn.config5(size=72)[0], 72), (glass_blob_scene(), 56)):
        ds, os_ = both(gpu, oracle, scene)
        kw = dict(spp=spp, spectral=spectral, max_bounces=bounces, ao_samples=ao, ao_radius=0.5)
        g, gst = ds.render(EPS, LEN, ft.ImageSize(n, n), cam, **kw)
        o, ocnt = os_.render(EPS, LEN, n, n, cam.as_array(), **kw)
        assert_bit_equal(g, o, f"glass spp={spp} spectral={spectral} bounces={bounces} ao={ao}")
        for k in ("rays_primary", "rays_ext", "rays_shadow", "hits_primary", "hits_shadow"):
            assert gst[k] == ocnt[k], k
        assert gst["rays_ext"] > n * n // 20 and gst["flags"] == ocnt["flags"] == 0


def test_extension_glass_degenerate_cases(gpu, oracle):
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    cam = syn.default_camera()
    scene, _ = syn.config5(size=64)
    ds, os_ = both(gpu, oracle, scene)
    # max_bounces = 0: glass is createSolid(tint) and runs on the reference kernel
    g0, _ = ds.render(EPS, LEN, ft.ImageSize(64, 64), cam)
    o0, _ = os_.render(EPS, LEN, 64, 64, cam.as_array())
    assert_bit_equal(g0, o0, "glass with max_bounces = 0")
    # column stripes of a glass frame (multi-GPU tiling): seeds hang on global pixel coordinates
    full, _ = ds.render(EPS, LEN, ft.ImageSize(64, 64), cam, spp=4, spectral=4, max_bounces=4)
    parts = [ds.render(EPS, LEN, ft.ImageSize(64, 64), cam, spp=4, spectral=4, max_bounces=4,
                       stripe_width=8, stripe_ranks=2, stripe_rank=r)[0] for r in range(2)]
    inter = np.empty_like(full)
    for r in range(2):
        for b in range(4):
            inter[(2 * b + r) * 8:(2 * b + r + 1) * 8] = parts[r][b * 8:(b + 1) * 8]
    assert_bit_equal(inter, full, "striped glass render")
    # bounces on a scene without glass change nothing
    plain, _ = syn.config2(seed=4)
    dp = gpu.scene(plain)
    a, _ = dp.render(EPS, LEN, ft.ImageSize(48, 48), cam)
    b, _ = dp.render(EPS, LEN, ft.ImageSize(48, 48), cam, max_bounces=4)
    assert_bit_equal(a, b, "max_bounces without glass")
    for kw in (dict(spectral=3, spp=4), dict(spectral=17), dict(max_bounces=-1), dict(max_bounces=65)):
        with pytest.raises(ft.FrayTracerError):
            ds.render(EPS, LEN, ft.ImageSize(8, 8), cam, **kw)
    with pytest.raises(ft.FrayTracerError):
        gpu.scene(SdfScene(SdfObject.create(SdfMaterial.createGlass((1, 1, 1), 0.0), SdfForm.Primitive.sphere((0, 0, 0), 1.0)), syn.BACKGROUND, []))


def test_extension_glass_full_size_properties(gpu):
    """BASELINE.json config 5 at its full size (2048^2, 16 spp, 4 bounces, 16 wavelength bins): finite, bounded
    by the brightest solid shading, deterministic (two runs agree bit for bit)."""
    scene, size = syn.config5()
    cam = syn.default_camera()
    ds = gpu.scene(scene)
    kw = dict(spp=16, spectral=16, max_bounces=4)
    a, st = ds.render(EPS, LEN, size, cam, **kw)
    b, _ = ds.render(EPS, LEN, size, cam, **kw)
    assert_bit_equal(a, b, "config 5 twice")
    assert np.isfinite(a).all() and a.min() >= 0.0 and st["flags"] == 0
    assert st["rays_primary"] == size.X * size.Y * 16 and st["rays_ext"] > st["rays_primary"] // 10
    solid, _ = ds.render(EPS, LEN, size, cam)
    assert a.max() <= solid.max() * 2.9          # a wavelength weight is below 2.9 (ft_spectral_table(16))


def test_differential_fuzz(gpu, oracle):
    """random combinator trees / materials / lights / cameras / render parameters (synthetic.fuzz_scene): the HIP
    path and the oracle agree float for float and ray for ray, or both reject the scene (tools/fuzz_parity.py
    runs the same loop over tens of thousands of seeds; profiles/r01f_fuzz.txt)"""
    rendered = 0
    for seed, big in [(s_, False) for s_ in range(5000, 5120)] + [(s_, True) for s_ in range(7000, 7010)]:   # big: 60-400 objects per union
        scene, cam, size, eps, ext = syn.fuzz_scene(seed, big)
        try:
            ds = gpu.scene(scene)
        except ft.FrayTracerError:
            with pytest.raises(oracle.OracleError):
                oracle.Oracle().scene(scene)
            continue
        g, st = ds.render(eps, LEN, size, cam, **ext)
        o, cnt = oracle.Oracle().scene(scene).render(eps, LEN, size.X, size.Y, cam.as_array(), **ext)
        assert_bit_equal(g, o, f"fuzz seed {seed} big={big} {ext}")
        for k in ("rays_primary", "rays_shadow", "rays_ext", "hits_primary", "hits_shadow", "flags"):
            assert st[k] == cnt[k], (seed, k)
        ds.close()
        rendered += 1
    assert rendered > 60


def test_frame_pipeline_over_rccl():
    """the N > 1 host path on the one GPU of the box: 1-rank RCCL group, double-buffered slabs, gather and
    de-interleave on a side stream (tests/rccl_rehearsal.py), and bench.py's own distributed branch"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_rehearsal.py")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "PIPELINE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--size", "512", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=dict(env, MASTER_PORT="29547"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["nan_or_cap_flags"] == 0
    assert "launch_overlap" in line["roofline"]


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver and the judge read (reduced frame so it takes seconds)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--size", "512", "--steps", "2", "--warmup", "1", "--cpu-columns", "32"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert d["config"]["max_abs_delta_vs_oracle"] == 0.0 and d["config"]["same_frame_at_4_spp"]["value"] > 0
    assert d["config"]["frames_streamed_on_two_lanes"]["value"] > 0
    assert abs(d["ms_per_step"] - d["roofline"]["kernel_ms"]) < 0.35 * d["ms_per_step"]       # one kernel per step dominates


def test_maximum_nesting_and_lds_footprint(gpu, oracle):
    """the largest scene the kernel accepts: 48 value slots (right-nested subtract / smooth-union chain) next to a
    2900-sphere smooth union whose constants fill the 48 KB LDS stage — 153 KB of dynamic LDS per workgroup; one
    level deeper is FT_ERR_UNSUPPORTED, not a wrong image"""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    P = SdfForm.Primitive

    def build(depth):
        rng = syn.Rng(77)
        blob = SdfForm.unionSmooth(0.3, [P.sphere(rng.pointInBall(3.0), rng.range(0.05, 0.2)) for _ in range(2900)])
        f = P.sphere(rng.pointInBall(2.0), 1.0)
        for _ in range(depth):
            f = SdfForm.subtract(P.sphere(rng.pointInBall(1.5), rng.range(1.0, 2.0)), SdfForm.unionSmooth(0.2, [P.sphere(rng.pointInBall(2.0), 0.3), f]))
        objs = [SdfObject.create(SdfMaterial.createSolid((0.8, 0.5, 0.3)), blob), SdfObject.create(SdfMaterial.createSolid((0.2, 0.5, 0.9)), f)]
        return SdfScene(SdfObject.union(objs), syn.BACKGROUND, syn.program_lights())

    scene = build(23)
    ds, os_ = both(gpu, oracle, scene)
    assert ds.info()["n_slots"] == 48
    cam = syn.default_camera()
    g, gst = ds.render(EPS, LEN, ft.ImageSize(40, 40), cam)
    o, ocnt = os_.render(EPS, LEN, 40, 40, cam.as_array())
    assert_bit_equal(g, o, "48 slots, full LDS stage")
    check_counts(gst, ocnt)
    with pytest.raises(ft.FrayTracerError, match="FT_MAX_SLOTS"):
        gpu.scene(build(30))


def test_unions_of_many_combinator_children(gpu, oracle):
    """combinator children of a union (without a union inside) are sub-programs the candidate loop runs on demand
    (FT_PR_CALL): hundreds of them need no slots of their own and are culled by the grid like primitives"""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    P = SdfForm.Primitive
    rng = syn.Rng(5)

    def mat(): return SdfMaterial.createSolid((rng.range_01(), rng.range_01(), rng.range_01()))
    objs = []
    for k in range(300):
        c = rng.pointInBall(4.0)
        if k % 3 == 0:
            objs.append(SdfObject.subtract(SdfObject.create(mat(), P.sphere(c, rng.range(0.4, 0.8))), P.sphere(c + rng.pointOnSphere(0.4), 0.35)))
        elif k % 3 == 1:
            objs.append(SdfObject.create(mat(), SdfForm.unionSmooth(0.2, [P.sphere(c + rng.pointOnSphere(0.3), rng.range(0.2, 0.4)) for _ in range(5)])))
        else:
            objs.append(SdfObject.intersect(SdfObject.create(mat(), P.torus(c, rng.pointOnSphere(1.0), 0.6, 0.2)), [P.sphere(c, 0.7), P.box(c, (0.6, 0.5, 0.7))]))
    objs += [syn.random_sphere(rng), syn.random_triangle(rng)]
    inner = SdfObject.union([syn.random_torus(rng), SdfObject.subtract(syn.random_sphere(rng), P.sphere((0, 0, 0), 0.5))])   # a union-bearing child: slot
    scene = SdfScene(SdfObject.union(objs + [inner]), syn.BACKGROUND, syn.program_lights())
    ds, os_ = both(gpu, oracle, scene)
    info = ds.info()
    assert info["fast_path"] == 2 and info["n_slots"] <= 6 and info["n_children"] >= 303
    cam = syn.default_camera()
    g, gst = ds.render(EPS, LEN, ft.ImageSize(96, 96), cam)
    o, ocnt = os_.render(EPS, LEN, 96, 96, cam.as_array())
    assert_bit_equal(g, o, "union of 300 combinator objects")
    check_counts(gst, ocnt)
    assert gst["hits_primary"] > 2000
    g, gst = ds.render(EPS, LEN, ft.ImageSize(64, 64), cam, spp=4, ao_samples=3, ao_radius=0.5)          # EXTENSION build of the variant
    o, ocnt = os_.render(EPS, LEN, 64, 64, cam.as_array(), spp=4, ao_samples=3, ao_radius=0.5)
    assert_bit_equal(g, o, "same, extension kernel")


def test_lean_kernel_at_2_pow_20_jobs_against_oracle(gpu, oracle):
    """the shape-specialised smooth-sphere kernel (whose inner loops the build's layout pass places, csrc/loop_layout.py) on a
    frame of 2^20 jobs, plain and 4-spp EXTENSION build: sampled columns against the oracle, counters exact"""
    scene, _ = syn.config3(n=64, size=1024)
    cam = syn.default_camera()
    ds, os_ = both(gpu, oracle, scene)
    assert ds.info()["fast_path"] == 1
    img, st = ds.render(EPS, LEN, ft.ImageSize(1024, 1024), cam)
    img4, st4 = ds.render(EPS, LEN, ft.ImageSize(1024, 1024), cam, spp=4)
    assert st["rays_primary"] == 1024 * 1024 and st4["rays_primary"] == 4 * 1024 * 1024
    want, _ = os_.render(EPS, LEN, 1024, 1024, cam.as_array(), x0=500, x1=516)
    assert_bit_equal(img[500:516], want, "1 spp against the oracle")
    want4, _ = os_.render(EPS, LEN, 1024, 1024, cam.as_array(), x0=500, x1=516, spp=4)
    assert_bit_equal(img4[500:516], want4, "4 spp against the oracle")
    assert st["shader_mhz"] > 500.0                      # the kernel reports the shader clock it ran at


def test_nan_distances_are_flagged_identically(gpu, oracle):
    """A degenerate capsule (From == To -> dirInv = 0/0) has a NaN distance.  The reference would spin
    forever in SdfForm.tryTrace; oracle and kernel both resolve such rays as misses and raise flag bit 0."""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    bad = SdfObject.create(SdfMaterial.createSolid((1, 0, 0)), SdfForm.Primitive.capsule((0, 0, 0), (0, 0, 0), 0.5))
    scene = SdfScene(bad, syn.BACKGROUND, syn.program_lights())
    g, gst, o, ocnt = render_both(gpu, oracle, scene, 32, 32)
    assert_bit_equal(g, o, "NaN scene")
    assert gst["flags"] & 1 and ocnt["flags"] & 1
    assert gst["hits_primary"] == ocnt["hits_primary"] == 0
    # A ray with a NaN direction through triangles: after the first step the query point is NaN, the triangle's edge
    # tests call MathF.Sign(NaN) (SdfForm.fs:235-237 — throws in .NET): flag bit 1 on both sides, then the NaN distance
    # (bit 0).  Alone, inside a union (Items.[0] is evaluated unconditionally) and inside an intersect.
    tri = lambda k: SdfForm.Primitive.triangle((-1 + k, -1, 0), (1 + k, -1, 0.2), (0 + k, 1, -0.1), 0.2)
    mat = SdfMaterial.createSolid((0.3, 0.6, 0.9))
    rays = np.array([[0, 0, -5, 0, 0, 1, 30, 0.01], [0, 0, -5, np.nan, 0, 1, 30, 0.01], [0.1, 0, -5, 0, np.nan, np.nan, 30, 0.01]], np.float32)
    for obj, want_flags in ((SdfObject.create(mat, tri(0)), 3),
                            (SdfObject.union([SdfObject.create(mat, tri(0)), SdfObject.create(mat, tri(0.5)), SdfObject.create(mat, tri(-0.5))]), 3),
                            (SdfObject.create(mat, SdfForm.intersect([tri(0), tri(0.1), SdfForm.Primitive.sphere((0, 0, 0), 3.0)])), 3)):
        sc = SdfScene(obj, syn.BACKGROUND, syn.program_lights())
        ds, os_ = both(gpu, oracle, sc)
        gg, gs = ds.trace_rays(rays)
        oo, oc = os_.trace_rays(rays)
        assert_bit_equal(gg, oo, "NaN-direction rays")
        assert gs["flags"] == oc["flags"] == want_flags
        g1, s1 = ds.trace_rays(rays[:1])                   # the healthy ray alone raises nothing
        assert s1["flags"] == 0 and tuple(g1[0]) != tuple(np.float32(syn.BACKGROUND))


def test_default_device_paths(oracle):
    """the calls that name no device — `Image.render eps len size camera (SdfScene.trace scene)`, `Image.toColors`, `SdfObject.tryTrace` — run on
    the process-wide default context of GPU 0 (examples/console.py uses exactly these)"""
    scene, _ = syn.config2(seed=3)
    cam = syn.default_camera()
    img = ft.Image.render(EPS, LEN, ft.ImageSize(64, 48), cam, ft.SdfScene.trace(scene))
    want, _ = oracle.Oracle().scene(scene).render(EPS, LEN, 64, 48, cam.as_array())
    assert_bit_equal(img, want, "Image.render on the default device")
    cols = ft.Image.toColors(2.2, 19, img)
    wcols, _ = oracle.tone_map(img, gamma=2.2, seed=19)
    assert np.array_equal(cols, wcols)
    rays = np.stack([oracle.pixel_ray(cam.as_array(), 64, 48, x, 24, EPS, LEN) for x in range(64)])
    assert_bit_equal(ft.SdfScene.trace(scene)(rays[32]), img[32, 24], "SdfScene.trace scene ray")
    assert ft.Device.default(0) is ft.Device.default(0)


def test_empty_and_degenerate_inputs(gpu, oracle):
    scene, _ = syn.config1()
    ds, os_ = both(gpu, oracle, scene)
    out, st = ds.trace_rays(np.zeros((0, 8), np.float32))
    assert out.shape == (0, 3) and st["rays_primary"] == 0
    d, m = ds.eval_distance(np.zeros((0, 3), np.float32))
    assert d.shape == (0,)
    cam = syn.default_camera()
    with pytest.raises(ft.FrayTracerError):
        ds.render(EPS, LEN, ft.ImageSize(0, 8), cam)
    with pytest.raises(ft.FrayTracerError):
        ds.render(EPS, LEN, ft.ImageSize(64, 64), cam, x0=60, n_columns=8)          # column range leaves the image
    # epsilon / length edge cases go through the same code on both sides
    for eps, length in ((0.01, 0.0), (0.5, 30.0), (1e-4, 12.0)):
        g, _ = ds.render(eps, length, ft.ImageSize(24, 24), cam)
        o, _ = os_.render(eps, length, 24, 24, cam.as_array())
        assert_bit_equal(g, o, f"eps={eps} length={length}")


# ---- BASELINE.json sizes -----------------------------------------------------------------------------------
def test_config2_full_size_against_oracle(gpu, oracle):
    """configs[1] at its stated 1024x1024, every pixel against the oracle."""
    scene, size = syn.config2()
    g, gst, o, ocnt = render_both(gpu, oracle, scene, size.X, size.Y)
    assert_bit_equal(g, o, "C2 1024^2")
    check_counts(gst, ocnt)


def test_config3_full_frame_properties_and_sampled_oracle(gpu, oracle):
    """configs[2] at 4096x4096: (a) every 64th column against the oracle, (b) the frame equals the
    concatenation of 8 interleaved-stripe renders (what 8 GPUs would produce), (c) counters add up."""
    scene, size = syn.config3()
    cam = syn.default_camera()
    ds, os_ = both(gpu, oracle, scene)
    W = H = size.X
    full, st = ds.render(EPS, LEN, size, cam)
    want, ocnt = os_.render(EPS, LEN, W, H, cam.as_array(), xstep=64)
    assert_bit_equal(full[::64], want, "C3 4096^2, every 64th column")
    assert st["rays_primary"] == W * H and st["flags"] == 0
    S, R = 16, 8
    shadow = 0
    for r in range(R):
        slab, sst = ds.render(EPS, LEN, size, cam, stripe_width=S, stripe_ranks=R, stripe_rank=r, n_columns=W // R)
        shadow += sst["rays_shadow"]
        got = slab.reshape(W // R // S, S, H, 3)
        ref = full.reshape(W // (R * S), R, S, H, 3)[:, r]
        assert_bit_equal(got, ref, f"stripe rank {r}")
    assert shadow == st["rays_shadow"]


def test_config2_as_worded_boxes_and_ao_full_size(gpu, oracle):
    """configs[1] as BASELINE.json words it: union of 16 spheres + 16 (EXTENSION) boxes, diffuse + 8 ambient-occlusion
    rays per primary hit (EXTENSION), 1024x1024 — every pixel and every ray counter against the oracle."""
    scene, size = syn.config2(boxes=True)
    cam = syn.default_camera()
    ds, os_ = both(gpu, oracle, scene)
    kw = dict(ao_samples=8, ao_radius=1.0)
    g, gst = ds.render(EPS, LEN, size, cam, **kw)
    o, ocnt = os_.render(EPS, LEN, size.X, size.Y, cam.as_array(), **kw)
    assert size.X == size.Y == 1024
    assert_bit_equal(g, o, "C2 (boxes + 8 AO rays) 1024^2")
    check_counts(gst, ocnt)
    assert gst["rays_ext"] == ocnt["rays_ext"] > 8 * 100000


def test_config3_at_4_spp_full_size_sampled_oracle(gpu, oracle):
    """configs[2] with its stated 4 samples per pixel (EXTENSION: 2x2 corner offsets, fixed-order resolve) at
    4096x4096: every 64th column against the oracle's own 4-spp render, all counters of the frame exact."""
    scene, size = syn.config3()
    cam = syn.default_camera()
    ds, os_ = both(gpu, oracle, scene)
    full, st = ds.render(EPS, LEN, size, cam, spp=4)
    want, ocnt = os_.render(EPS, LEN, size.X, size.Y, cam.as_array(), xstep=64, spp=4)
    assert size.X == size.Y == 4096
    assert_bit_equal(full[::64], want, "C3 4096^2 at 4 spp, every 64th column")
    assert st["rays_primary"] == 4 * size.X * size.Y and st["flags"] == 0
    # sample 0 of the extension is the reference's sample: the 1-spp frame of the same columns bounds nothing, but the
    # per-column ray counts of the sampled columns must agree with the oracle's
    part, pst = ds.render(EPS, LEN, size, cam, spp=4, x0=2048, n_columns=1)
    one, ocnt1 = os_.render(EPS, LEN, size.X, size.Y, cam.as_array(), x0=2048, x1=2049, spp=4)
    assert_bit_equal(part, one, "column 2048 on its own")
    for k in ("rays_primary", "rays_shadow", "hits_primary", "hits_shadow"):
        assert pst[k] == ocnt1[k], k


def test_config4_8192_sampled_oracle_and_stripes(gpu, oracle):
    """configs[3]: the C3 scene at 8192x8192.  (a) every 128th column of the one-GPU frame against the oracle,
    (b) the frame equals what 8 ranks' interleaved 16-column stripes concatenate to, (c) shadow-ray counts add up."""
    scene, size = syn.config4()
    cam = syn.default_camera()
    ds, os_ = both(gpu, oracle, scene)
    W = H = size.X
    assert W == 8192
    full, st = ds.render(EPS, LEN, size, cam)
    want, ocnt = os_.render(EPS, LEN, W, H, cam.as_array(), xstep=128)
    assert_bit_equal(full[::128], want, "C4 8192^2, every 128th column")
    assert st["rays_primary"] == W * H and st["flags"] == 0
    S, R = 16, 8
    shadow = 0
    ref = full.reshape(W // (R * S), R, S, H, 3)
    for r in range(R):
        slab, sst = ds.render(EPS, LEN, size, cam, stripe_width=S, stripe_ranks=R, stripe_rank=r, n_columns=W // R)
        shadow += sst["rays_shadow"]
        assert_bit_equal(slab.reshape(W // R // S, S, H, 3), ref[:, r], f"stripe rank {r}")
        del slab
    assert shadow == st["rays_shadow"]


def test_program_fs_scene_full_size_against_oracle(gpu, oracle):
    """The reference's only workload, src/FrayTracer.Console/Program.fs:14-83: System.Random(19), 1000 random tori,
    subtract(intersect(union ..., sphere 3.5), sphere 2.5), directional + point light, 1000x1000 (Program.fs:24-26),
    eps 0.01, length 30 — every pixel and every counter against the oracle."""
    scene, size = syn.console_scene()
    assert (size.X, size.Y) == (1000, 1000)
    g, gst, o, ocnt = render_both(gpu, oracle, scene, size.X, size.Y)
    assert_bit_equal(g, o, "Program.fs scene 1000^2")
    check_counts(gst, ocnt)
    assert gst["hits_primary"] > 100000 and gst["rays_shadow"] > 100000
    # and as the ray-buffer form of the same pixels (SdfScene.trace over Camera.uniformPixelToRay rays), one column
    cam = syn.default_camera()
    rays = np.stack([oracle.pixel_ray(cam.as_array(), 1000, 1000, 500, y, EPS, LEN) for y in range(1000)])
    assert_bit_equal(gpu.scene(scene).trace_rays(rays)[0], o[500], "column 500 through ft_trace_rays")


def test_config5_full_size_against_oracle(gpu, oracle):
    """configs[4] (EXTENSION only): glass, 4 bounces, 16 wavelength bins, 2048x2048 at 16 spp — every 16th column
    against the oracle's definition, counters of those columns exact."""
    scene, size = syn.config5()
    cam = syn.default_camera()
    ds, os_ = both(gpu, oracle, scene)
    kw = dict(spp=16, spectral=16, max_bounces=4)
    full, st = ds.render(EPS, LEN, size, cam, **kw)
    want, ocnt = os_.render(EPS, LEN, size.X, size.Y, cam.as_array(), xstep=16, **kw)
    assert_bit_equal(full[::16], want, "C5 2048^2 x 16 spp, every 16th column")
    assert st["rays_primary"] == 16 * size.X * size.Y and st["flags"] == 0


def test_tone_map_on_the_device_matches_the_oracle(gpu, oracle):
    """Image.toColors (global max, Pow(c / max, 1 / gamma), x 254.5 + noise, half-to-even, min 255) and the
    buffer order of Image.toBitmap as HIP kernels: byte-exact against the oracle's restatement, with and without noise,
    in both orders; ragged sizes, black / NaN / huge pixels; the frame of the Program.fs scene end to end."""
    rng = np.random.default_rng(21)
    for (X, Y) in [(64, 64), (37, 101), (130, 19), (1, 1), (200, 65)]:
        img = (rng.random((X, Y, 3)) ** 3 * 4.0).astype(np.float32)
        if X > 30:
            img[3, 2] = (np.nan, 0.0, -1.0); img[7, 1] = (1e30, 1e-30, 3.0); img[9, 0] = 0.0
            # a pixel with a NaN channel is skipped WHOLE by the max (MathF.Max propagates NaN, Seq.max then keeps `acc`,
            # Math.fs:83 / Array2D.fs:45-50): its 5e30 must not become the normalisation although it is the largest channel
            img[5, 3] = (np.nan, 5e30, 0.0); img[11, 4] = (2.0, np.nan, 7e30)
        for gamma in (2.2, 1.0):
            for seed in (None, 19):
                for bmp in (False, True):
                    want, wmx = oracle.tone_map(img, gamma=gamma, seed=seed, bmp_order=bmp)
                    got = ft.Image.toColors(gamma, seed, img, gpu, bmp_order=bmp)
                    assert got.shape == want.shape and np.array_equal(got, want), (X, Y, gamma, seed, bmp)
                    if X > 30: assert wmx == np.float32(1e30)
    z = ft.Image.toColors(2.2, None, np.zeros((16, 8, 3), np.float32), gpu)
    assert z.max() == 0
    # end to end: Program.fs:90-100 — render + toColors on the device, 3 bytes per pixel come back
    scene, size = syn.console_scene()
    cam = syn.default_camera()
    ds = gpu.scene(scene)
    frame, st = ds.render(EPS, LEN, size, cam)
    for seed, bmp in ((None, False), (19, True)):
        got, mx, st2 = ds.render_colors(EPS, LEN, size, cam, gamma=2.2, seed=seed, bmp_order=bmp)
        want, wmx = oracle.tone_map(frame, gamma=2.2, seed=seed, bmp_order=bmp)
        assert mx == wmx == frame.max() and np.array_equal(got, want)
        assert st2["rays_primary"] == st["rays_primary"] and st2["rays_shadow"] == st["rays_shadow"]
    assert got.shape == (1000, 1000, 3) and got.max() == 255
    with pytest.raises(ft.FrayTracerError):
        ds.render_colors(EPS, LEN, size, cam, x0=8, n_columns=16)          # the normalisation needs the whole frame


def test_tone_map_of_the_4096_frame_leaves_as_bytes(gpu, oracle):
    """configs[2]'s 4096x4096 frame through ft_render_colors: 50 MB of bytes instead of 201 MB of floats cross PCIe, and
    every byte equals the oracle's tone map of the float frame"""
    scene, size = syn.config3()
    cam = syn.default_camera()
    ds = gpu.scene(scene)
    frame, _ = ds.render(EPS, LEN, size, cam)
    got, mx, _ = ds.render_colors(EPS, LEN, size, cam, gamma=2.2, seed=7, bmp_order=True)
    want, wmx = oracle.tone_map(frame, gamma=2.2, seed=7, bmp_order=True)
    assert got.nbytes == 4096 * 4096 * 3 and mx == wmx
    assert np.array_equal(got, want)


def test_device_grid_build_equals_host_build(gpu, oracle):
    """the per-cell part of buildSpatialLookup runs on the GPU for large unions; the resulting
    grid (cell centres, CSR offsets, sorted (LowerBound, item) lists) must equal the host build and the
    oracle's bit for bit."""
    host = ft.Device(-1)
    try:
        for scene in (syn.console_like(n=1000)[0], syn.console_like(seed=4, n=400, factory=syn.random_triangle)[0]):
            gd = gpu.scene(scene).grid(0)
            gh = host.scene(scene).grid(0)
            assert gd["counts"] == gh["counts"]
            for k in ("aabbMin", "cellSizeInv", "centers", "lower"):
                assert np.array_equal(gd[k].view(np.uint32), gh[k].view(np.uint32)), k
            assert np.array_equal(gd["cell_start"], gh["cell_start"]) and np.array_equal(gd["child"], gh["child"])
            assert len(gd["child"]) > 20000
    finally:
        host.close()


def test_union_fast_sqrt_clamp_points(gpu, oracle):
    """The union loop's clamped fast square roots (flatten-time flag fastQ + per-evaluation point test) must
    give the reference's value also where a root's operand is zero or tiny: exactly on sphere / torus centres,
    capsule end points and axis points, triangle vertices and edges; and must fall back to the IEEE path for
    huge / non-finite points."""
    rng = np.random.default_rng(11)
    for factory in (syn.random_sphere, syn.random_capsule, syn.random_torus, syn.random_triangle):
        scene, _ = syn.console_like(seed=6, n=80, factory=factory)
        ds, os_ = both(gpu, oracle, scene)
        assert ds.info()["n_grids"] == 1
        union = scene.Object.kids[0].kids[0]
        special = []
        for obj in union.kids:
            form = obj.kids[1]
            for a in form.args:
                if isinstance(a, tuple):
                    special.append(a)
            if form.kind == "capsule":
                special.append(tuple(0.5 * (np.float32(form.args[0][i]) + np.float32(form.args[1][i])) for i in range(3)))
            if form.kind == "triangle":
                special.append(tuple(0.5 * (np.float32(form.args[0][i]) + np.float32(form.args[2][i])) for i in range(3)))
        special = np.array(special, np.float32)
        pts = np.concatenate([special, special + np.float32(1e-20), special + np.float32(3e-13), rng.uniform(-5, 5, (3000, 3)),
                              [[3e4, 0, 0], [1e30, 1e30, 0], [np.inf, 0, 0], [np.nan, 1, 2]]]).astype(np.float32)
        d, _ = ds.eval_distance(pts)
        O = oracle.Oracle()
        with np.errstate(all="ignore"):
            assert_bit_equal(d, O.form_distance(O.object_form(os_.object), pts), f"union fast-sqrt, {factory.__name__}")


# ---- last on purpose: needs two GPUs, i.e. it never ran on the one-GPU development boxes -------------------------------
def test_render_multi_over_rccl_when_two_devices_exist(gpu):
    """ft_render_multi's real branch — one host thread per device, ncclGather over xGMI into the first device, strided
    de-interleave there, one copy out — needs two distinct GPUs: skipped on a one-GPU box, run wherever the suite meets
    two or more (the one-GPU rehearsal above takes the same code path except for the collective itself)."""
    try:
        second = ft.Device(1)
    except ft.FrayTracerError:
        pytest.skip("needs at least two GPUs")
    try:
        scene, _ = syn.config2(seed=8)
        cam = syn.default_camera()
        W, H = 256, 144
        full, st = gpu.scene(scene).render(EPS, LEN, ft.ImageSize(W, H), cam)
        for stripe in (16, 64):
            img, stn = ft.render_multi([gpu, second], scene, EPS, LEN, ft.ImageSize(W, H), cam, stripe_width=stripe)
            assert_bit_equal(img, full, f"ft_render_multi over 2 GPUs, stripes of {stripe}")
            assert stn["rays_primary"] == st["rays_primary"] and stn["rays_shadow"] == st["rays_shadow"]
    finally:
        second.close()


def _mathf_max(a, b):
    """MathF.Max (IEEE 754-2019 maximum): NaN propagates, -0 < +0 (Math.fs:83; ft_math.h ft_max)"""
    with np.errstate(all="ignore"):
        r = np.where(a > b, a, b)
        z = (a == b)
        r = np.where(z & np.signbit(a), b, np.where(z, a, r))
        return np.where(np.isnan(a) | np.isnan(b), np.float32(np.nan), r).astype(np.float32)


def test_branch_free_min_max_of_the_carved_kernels(gpu):
    """ft_max_dev / ft_vmin (v_max_f32 / v_min_f32 plus an unordered test) against MathF.Max / Min on every pair of a table of special values
    and on random bit patterns: zeros of both signs, infinities, NaNs, subnormals."""
    rng = np.random.default_rng(11)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, -3.4e38, 0.01, -0.01], np.float32)
    a, b = [x.ravel() for x in np.meshgrid(sp, sp)]
    ra = rng.integers(0, 0xFFFFFFFF, 1_000_000, dtype=np.uint32).view(np.float32)
    rb = rng.integers(0, 0xFFFFFFFF, 1_000_000, dtype=np.uint32).view(np.float32)
    a, b = np.concatenate([a, ra, ra]), np.concatenate([b, rb, ra])
    assert_bit_equal(gpu.math_eval(13, a, b), _mathf_max(a, b), "MathF.Max")
    ok = ~np.isnan(a)                                                     # the walk's minimum is never NaN when it calls Min
    want_min = -_mathf_max(-a[ok], -b[ok])
    assert_bit_equal(gpu.math_eval(14, a[ok], b[ok]), want_min, "MathF.Min")


def _carved_cases():
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    P = SdfForm.Primitive
    makers = {"spheres": syn.random_sphere, "capsules": syn.random_capsule, "tori": syn.random_torus, "triangles": syn.random_triangle, "boxes": syn.random_box}
    S1, S2, CAP = P.sphere((0.0, 0.0, 0.0), 3.2), P.sphere((-0.5, 1.0, -2.0), 2.2), P.capsule((-2.0, -1.0, -1.0), (2.0, 1.0, -1.5), 1.3)
    BOX = P.box((0.3, -0.2, 0.0), (2.5, 2.0, 2.8))
    tails = {"plain": lambda u: u,
             "isect": lambda u: SdfObject.intersect(u, [S1]),
             "sub": lambda u: SdfObject.subtract(u, S2),
             "Program.fs": lambda u: SdfObject.subtract(SdfObject.intersect(u, [S1]), S2),
             "sub then isect": lambda u: SdfObject.intersect(SdfObject.subtract(u, S2), [S1]),
             "isect two": lambda u: SdfObject.intersect(u, [S1, CAP]),
             "sub sub": lambda u: SdfObject.subtract(SdfObject.subtract(u, S2), CAP),
             "isect box, sub capsule": lambda u: SdfObject.subtract(SdfObject.intersect(u, [BOX]), CAP)}
    out = []
    for ki, (kind, mk) in enumerate(makers.items()):
        rng = syn.Rng(40 + ki)
        objs = [mk(rng) for _ in range(120)]
        for ti, (tn, tail) in enumerate(tails.items()):
            if kind in ("capsules", "triangles", "boxes") and ti % 3 != ki % 3:      # every kind sees the reference's structure and a third of the rest
                if tn != "Program.fs":
                    continue
            out.append((f"{kind}, {tn}", SdfScene(tail(SdfObject.union(objs)), syn.BACKGROUND, syn.program_lights())))
    rng = syn.Rng(77)
    mixed = [m(rng) for _ in range(30) for m in (syn.random_sphere, syn.random_torus, syn.random_capsule, syn.random_triangle)]
    out.append(("mixed kinds, Program.fs", SdfScene(tails["Program.fs"](SdfObject.union(mixed)), syn.BACKGROUND, syn.program_lights())))
    out.append(("mixed kinds, plain", SdfScene(SdfObject.union(mixed), syn.BACKGROUND, syn.program_lights())))
    return out


def test_carved_union_kernels_against_oracle(gpu, oracle):
    """FT_OPT_CARVED (round 4): scenes that are one union of primitives plus at most two intersect / subtract steps run on kernels specialised
    per primitive kind (kernels.hip ft_eval_carved).  Every kind x every tail: frame, counters and explicit rays (several epsilons, huge and
    NaN origins) are the oracle's, with the kernel on and off, with and without its early exits (FT_OPT_LAZY_UNION) and the escape shortcut."""
    cam = syn.default_camera()
    W, H = 176, 120
    rr = np.random.default_rng(8)
    o = (rr.normal(size=(600, 3)) * 3.0).astype(np.float32)
    d = rr.normal(size=(600, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    eps = rr.choice([0.01, 0.05, 0.3, 2.0, 1e-3], (600, 1)).astype(np.float32)
    rays = np.concatenate([o, d, np.full((600, 1), 30, np.float32), eps], axis=1).astype(np.float32)
    rays[0, 0] = np.nan; rays[1, 1] = 3e4; rays[2, 2] = -1e30; rays[3, 3:6] = 0.0; rays[4, 6] = np.inf; rays[5, 0:3] = 0.0
    try:
        for name, scene in _carved_cases():
            ds, os_ = both(gpu, oracle, scene)
            assert ds.info()["fast_path"] == 3, name
            want, ocnt = os_.render(EPS, LEN, W, H, cam.as_array())
            with np.errstate(all="ignore"):
                want_rays, want_rcnt = os_.trace_rays(rays)
            for carved, lazy, esc in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0)):
                gpu.set_option("carved", carved); gpu.set_option("lazy_union", lazy); gpu.set_option("escape", esc)
                g, gst = ds.render(EPS, LEN, ft.ImageSize(W, H), cam)
                tag = f"{name}: carved {carved}, lazy {lazy}, escape {esc}"
                assert_bit_equal(g, want, tag)
                check_counts(gst, ocnt)
                with np.errstate(all="ignore"):
                    got_rays, rst = ds.trace_rays(rays)
                assert_bit_equal(got_rays, want_rays, tag + " (ray buffer)")
                assert rst["flags"] == want_rcnt["flags"], (tag, rst["flags"], want_rcnt["flags"])
            ds.close()
    finally:
        gpu.set_option("carved", 1); gpu.set_option("lazy_union", 1); gpu.set_option("escape", 1)


def _offset_scenes(off):
    """three small scenes of different kernel families around the point `off`: the reference's structure (carved union), a smooth union of
    spheres (lean kernel) and a union with a combinator child (general kernel)"""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    P = SdfForm.Primitive
    off = np.asarray(off, np.float32)
    rng = syn.Rng(91)
    mat = [SdfMaterial.createSolid((0.3 + 0.1 * i, 0.8 - 0.1 * i, 0.4)) for i in range(4)]
    at = lambda v: tuple((np.asarray(v, np.float32) + off).tolist())
    tori = [SdfObject.create(mat[i % 4], P.torus(at(rng.pointInBall(2.0)), rng.pointOnSphere(1.0), rng.range(0.2, 0.5), rng.range(0.1, 0.2))) for i in range(40)]
    carved = SdfObject.subtract(SdfObject.intersect(SdfObject.union(tori), [P.sphere(at((0, 0, 0)), 1.8)]), P.sphere(at((-0.3, 0.6, -1.2)), 1.2))
    blob = SdfObject.create(mat[0], SdfForm.unionSmooth(0.25, [P.sphere(at(rng.pointInBall(1.5)), rng.range(0.2, 0.5)) for _ in range(24)]))
    nested = SdfObject.union([SdfObject.create(mat[1], SdfForm.subtract(P.sphere(at((0.5, 0, 0)), 1.0), P.sphere(at((0.9, 0.2, -0.4)), 0.6))),
                              SdfObject.create(mat[2], P.capsule(at((-1.5, -0.5, 0)), at((-0.5, 0.8, 0.3)), 0.3)), SdfObject.create(mat[3], P.sphere(at((0, -1.2, 0.4)), 0.5))])
    lights = [SdfLight_dir(), SdfLight_point(at((-0.5, 0.0, -2.0)))]
    return [("carved", SdfScene(carved, syn.BACKGROUND, lights)), ("smooth", SdfScene(blob, syn.BACKGROUND, lights)), ("nested", SdfScene(nested, syn.BACKGROUND, lights))]


def SdfLight_dir():
    from fraytracer_amd import SdfLight
    return SdfLight.directional((-0.5, -1.0, 1.0), (0.5, 0.5, 0.5))


def SdfLight_point(pos):
    from fraytracer_amd import SdfLight
    return SdfLight.point(pos, (10.0, 0.0, 0.0))


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (5000.0, -3000.0, 4000.0)])
def test_escape_shortcut_at_its_edge(gpu, oracle, offset):
    """The escape shortcut reasons about the ideal line while the reference accumulates the origin in float32 (Ray.fs:9-13); its padding is
    proved to cover that drift (scene.cpp "drift of the marched points").  Probe where the proof is thinnest: rays tangent to the support sphere
    within +-2 paddings, tiny epsilons (1e-4, 1e-5: below the float spacing at |c| = 7000), Length 1000, starts from just outside to 50 radii away,
    unit and non-unit directions (a point light's shadow direction is diff / |diff|^2), scenes 7000 away from the origin.  Colours, ray / hit
    counters and flags must be the oracle's with the shortcut on and off — the oracle has no shortcut at all."""
    rng = np.random.default_rng(5)
    try:
        for name, scene in _offset_scenes(offset):
            ds, os_ = both(gpu, oracle, scene)
            cx, cy, cz, R = ds.support_sphere()
            assert R > 0, name
            c = np.array([cx, cy, cz], np.float64)
            n = 360
            u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
            start = c + u * R * rng.choice([1.02, 1.5, 3.0, 10.0, 50.0], (n, 1))
            # aim at a point whose distance from the centre is R + epsilon + k paddings, in a random direction perpendicular to the start direction
            v = np.cross(u, rng.normal(size=(n, 3))); v /= np.linalg.norm(v, axis=1, keepdims=True)
            eps = rng.choice([1e-2, 1e-4, 1e-5], (n, 1))
            pad = max(0.02, 0.01 * R)
            miss = R + eps + rng.uniform(-2.0, 2.0, (n, 1)) * pad
            miss[180:240] = rng.uniform(0.0, 1.0, (60, 1)) * min(R, 2.5)           # ... and a family through the scene itself (hits, shadow rays of both lights)
            target = c + v * miss
            d = target - start; d /= np.linalg.norm(d, axis=1, keepdims=True)
            d[240:300] *= rng.choice([0.2, 0.5, 3.0, 17.0], (60, 1))              # non-unit directions, as SdfLight.point casts them
            d[300:330] *= -1.0                                                     # pointing away
            length = rng.choice([30.0, 1000.0], (n, 1))
            rays = np.concatenate([start, d, length, eps], axis=1).astype(np.float32)
            with np.errstate(all="ignore"):
                want, ocnt = os_.trace_rays(rays)
            evals = {}
            for esc in (1, 0):
                gpu.set_option("escape", esc)
                with np.errstate(all="ignore"):
                    got, gst = ds.trace_rays(rays)
                assert_bit_equal(got, want, f"{name} at {offset}, escape {esc}")
                for k in ("rays_shadow", "hits_primary", "hits_shadow", "flags"):
                    assert gst[k] == ocnt[k], (name, offset, esc, k, gst[k], ocnt[k])
                evals[esc] = gst["sdf_evals"]
            assert evals[1] <= evals[0], (name, evals)
            if name == "carved" and offset == (0.0, 0.0, 0.0):
                assert evals[1] < evals[0], evals                                  # the shortcut is alive (not gated away) where it matters
            ds.close()
    finally:
        gpu.set_option("escape", 1)


def test_a_scene_of_nan_constants_takes_no_escape_shortcut(gpu, oracle):
    """A capsule of length 0 (dirInv = 0 / 0) or a collinear triangle evaluates to NaN everywhere; the reference's march never ends on such a
    value and both sides raise the NaN flag.  Such a scene gets no support sphere (scene.cpp supportOf), so a ray that would have been ended
    early still meets the NaN and reports it: flags are the oracle's with the shortcut on and off."""
    from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene
    P = SdfForm.Primitive
    m = SdfMaterial.createSolid((0.5, 0.5, 0.5))
    ball = SdfObject.create(m, P.sphere((0.0, 0.0, 0.0), 0.5))
    bads = (P.capsule((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.2), P.triangle((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (2.0, 2.0, 2.0), 0.1))
    # the NaN form sits where the support rules never look: the subtrahend of a subtract, a later child of an intersect
    flagged = 0
    for scene_obj in [SdfObject.subtract(ball, bad) for bad in bads] + [SdfObject.intersect(ball, [P.sphere((0.1, 0.0, 0.0), 0.6), bad]) for bad in bads] + [SdfObject.create(m, bads[0])]:
        scene = SdfScene(scene_obj, syn.BACKGROUND, syn.program_lights())
        ds, os_ = both(gpu, oracle, scene)
        assert ds.support_sphere()[3] < 0
        rays = np.array([[0, 0, -5, 0, 0, 1, 30, 0.01], [0, 3, -5, 0, 0, 1, 30, 0.01], [4, 0, 0, 1, 0, 0, 30, 0.01], [0, 0, -5, 0, 0, -1, 30, 0.01]], np.float32)
        with np.errstate(all="ignore"):
            want, ocnt = os_.trace_rays(rays)
        try:
            for esc in (1, 0):
                gpu.set_option("escape", esc)
                with np.errstate(all="ignore"):
                    got, gst = ds.trace_rays(rays)
                assert_bit_equal(got, want, f"NaN constants, escape {esc}")
                assert gst["flags"] == ocnt["flags"], (esc, gst["flags"], ocnt["flags"])
            flagged += ocnt["flags"] != 0
        finally:
            gpu.set_option("escape", 1)
        ds.close()
    assert flagged >= 3            # NaN distance (flag 1) or MathF.Sign(NaN) (flag 2) was really met


def test_reusing_the_centre_probe_as_a_secondary_rays_first_step_changes_no_pixel(gpu, oracle):
    """FT_OPT_REUSE (round 4): every shadow ray and every EXTENSION ambient-occlusion ray starts at the pulled-back hit position, where the fourth
    probe of SdfForm.normal has just evaluated the scene; its first evaluation is that value and is not computed again.  Likewise every primary ray of
    Image.render starts at the camera position, which each wave evaluates once.  Frames, ray / hit counters, flags and explicit rays are the oracle's with
    the option on and off, in every kernel family and with both light types; with it on exactly one evaluation per ray is saved (a camera inside an object —
    a hit at step 0 — included)."""
    cam = syn.default_camera()
    cases = [("C3 lean", syn.config3(n=64, size=128)[0], {}), ("C2 carved mixed", syn.config2(seed=4, size=128)[0], {}),
             ("Program.fs structure, both lights", syn.console_scene(n=120, size=128)[0], {}), ("mixed nested (general)", syn.mixed_nested()[0], {}),
             ("combinator zoo (calls)", syn.combinator_zoo()[0], {}), ("C2 boxes + AO + 4 spp (EXTENSION)", syn.config2(boxes=True, size=96)[0], dict(spp=4, ao_samples=5, ao_radius=0.6)),
             ("glass (EXTENSION)", syn.config5()[0], dict(spp=4, spectral=4, max_bounces=4))]
    rr = np.random.default_rng(12)
    o = (rr.normal(size=(300, 3)) * 3.0).astype(np.float32)
    d = rr.normal(size=(300, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d, np.full((300, 1), 30, np.float32), rr.choice([0.01, 0.2, 1.0], (300, 1)).astype(np.float32)], axis=1).astype(np.float32)
    rays[0, 0] = np.nan
    try:
        for name, scene, kw in cases:
            ds, os_ = both(gpu, oracle, scene)
            W, H = 112, 80
            want, ocnt = os_.render(EPS, LEN, W, H, cam.as_array(), **kw)
            with np.errstate(all="ignore"):
                want_rays, rcnt = os_.trace_rays(rays)
            evals = {}
            for reuse in (1, 0):
                gpu.set_option("reuse", reuse)
                g, gst = ds.render(EPS, LEN, ft.ImageSize(W, H), cam, **kw)
                assert_bit_equal(g, want, f"{name}, reuse {reuse}")
                check_counts(gst, ocnt)
                assert gst["rays_ext"] == ocnt["rays_ext"]
                evals[reuse] = gst["sdf_evals"]
                with np.errstate(all="ignore"):
                    got_rays, rst = ds.trace_rays(rays)
                assert_bit_equal(got_rays, want_rays, f"{name}: ray buffer, reuse {reuse}")
                # bit 1 (MathF.Sign(NaN) inside a triangle) may also be raised by an evaluation the general kernels make up front (a union's slot children) for
                # the NaN ray's query point — the reference, whose running minimum is NaN by then, never reaches that child; bits 0 and 2 are exact
                assert rst["flags"] & ~2 == rcnt["flags"] & ~2 and (rst["flags"] & 1), (name, rst["flags"], rcnt["flags"])
                if ds.info()["fast_path"] in (1, 3) and not kw: assert rst["flags"] == rcnt["flags"], (name, rst["flags"], rcnt["flags"])
            gpu.set_option("escape", 0)                                # without the escape shortcut every secondary ray makes its first step: exactly one evaluation each is saved
            try:
                n = {}
                for reuse in (1, 0):
                    gpu.set_option("reuse", reuse)
                    _, st = ds.render(EPS, LEN, ft.ImageSize(W, H), cam, **{k: v for k, v in kw.items() if k != "max_bounces" and k != "spectral"})
                    n[reuse] = (st["sdf_evals"], st["rays_primary"] + st["rays_shadow"] + (st["rays_ext"] if "ao_samples" in kw else 0))
            finally:
                gpu.set_option("escape", 1)
            if "max_bounces" not in kw:
                assert n[0][0] - n[1][0] == n[1][1], (name, n)
            assert evals[1] <= evals[0]
            ds.close()
        # the camera inside an object (first evaluation is a hit: material and normal come from the shared value) and inside the carved-out sphere of the reference's structure
        from fraytracer_amd import Camera, Lens
        for name, scene, pos in (("camera inside a blob", syn.config3(n=40, size=64)[0], None), ("camera inside the Program.fs structure", syn.console_scene(n=150, size=64)[0], (-0.5, 1.0, -2.0))):
            ds, os_ = both(gpu, oracle, scene)
            if pos is None:
                pos = tuple(np.asarray(scene.Object.kids[1].kids[0].args[0], np.float32).tolist())      # the centre of the first sphere
            cam2 = Camera.lookAt(Position=pos, LookAt=(0.3, 0.2, 5.0), Up=(0.0, 1.0, 0.0), Lens=Lens.create(60.0))
            want, ocnt = os_.render(EPS, LEN, 72, 56, cam2.as_array())
            for reuse in (1, 0):
                gpu.set_option("reuse", reuse)
                g, gst = ds.render(EPS, LEN, ft.ImageSize(72, 56), cam2)
                assert_bit_equal(g, want, f"{name}, reuse {reuse}")
                check_counts(gst, ocnt)
            ds.close()
    finally:
        gpu.set_option("reuse", 1); gpu.set_option("escape", 1)
