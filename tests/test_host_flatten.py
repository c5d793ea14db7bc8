"""libfraytracer_hip's host side (scene construction + flattening, no GPU) against the oracle:
boundaries and the uniform grids must agree bit for bit, the flattened program must have the
documented shape.  Uses a host-only context (device -1); nothing here launches a kernel."""
import numpy as np
import pytest

import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from fraytracer_amd import SdfForm, SdfObject, SdfMaterial, SdfScene


@pytest.fixture(scope="module")
def host():
    d = ft.Device(-1)
    yield d
    d.close()


def all_forms(scene):
    out, seen = [], set()

    def walk(n):
        if id(n) in seen: return
        seen.add(id(n))
        if isinstance(n, ft.api.Form): out.append(n)
        for k in n.kids: walk(k)
    walk(scene.Object)
    return out


@pytest.mark.parametrize("make", [lambda: syn.config2()[0], lambda: syn.config2(boxes=True)[0], lambda: syn.config3(n=50)[0],
                                  lambda: syn.console_like(n=150)[0], lambda: syn.mixed_nested()[0]])
def test_boundaries_match_oracle_bitwise(host, oracle, make):
    scene = make()
    O = oracle.Oracle()
    mh, mo = {}, {}
    ft.realise(scene.Object, host, mh)
    ft.realise(scene.Object, O, mo)
    forms = all_forms(scene)
    assert len(forms) > 3
    for f in forms:
        a = np.array(host.form_boundary(mh[id(f)]), np.float32)
        b = np.array(O.form_boundary(mo[id(f)]), np.float32)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (f, a, b)


@pytest.mark.parametrize("make", [lambda: syn.config2()[0], lambda: syn.console_like(n=400)[0],
                                  lambda: syn.console_like(seed=3, n=80, factory=syn.random_triangle)[0]])
def test_grid_matches_oracle_bitwise(host, oracle, make):
    scene = make()
    ds = host.scene(scene)
    info = ds.info()
    assert info["n_grids"] == 1
    g = ds.grid(0)
    O = oracle.Oracle()
    memo = {}
    obj = ft.realise(scene.Object, O, memo)
    # the (object) union is the innermost object of subtract(intersect(union, ..), ..) or the root itself
    node = scene.Object
    while node.kind != "union":
        node = node.kids[0]
    og = O.grid(O.object_form(memo[id(node)]))
    assert g["counts"] == og["counts"]
    for k in ("aabbMin", "cellSizeInv"):
        assert np.array_equal(g[k].view(np.uint32), og[k].view(np.uint32)), k
    assert np.array_equal(g["cell_start"], og["cell_start"])
    assert np.array_equal(g["centers"].view(np.uint32), og["centers"].view(np.uint32))
    assert np.array_equal(g["lower"].view(np.uint32), og["lower"].view(np.uint32))
    assert np.array_equal(g["child"], og["child"])
    ds.close()


def forms_far(k):
    return SdfForm.Primitive.sphere((10.0 + k, 0.0, 0.0), 0.5)


def test_flattened_program_shapes(host):
    # C3: one staged fast sphere run + FIN + SETLEAF -> lean kernel variant, one value slot
    i = host.scene(syn.config3()[0]).info()
    assert (i["n_instr"], i["n_slots"], i["fast_path"], i["n_grids"]) == (3, 1, 1, 0)
    # console scene: UNION, ISECT_RUN(sphere), PRIM(sphere), SUBTRACT -> "carved union" kernel (3), two slots for the interpreter, 1000 children
    i = host.scene(syn.console_like(n=1000)[0]).info()
    assert (i["n_instr"], i["n_slots"], i["fast_path"], i["n_grids"], i["n_children"], i["n_materials"]) == (4, 2, 3, 1, 1000, 1000)
    # a plain union of primitives is the same shape with an empty tail; a union with a combinator child is not
    assert host.scene(syn.config2()[0]).info()["fast_path"] == 3
    assert host.scene(syn.mixed_nested()[0]).info()["fast_path"] in (0, 2)
    # three steps behind the union: more than the carved kernels' tail holds -> general kernel
    u = SdfObject.union([syn.random_sphere(syn.Rng(1)), syn.random_sphere(syn.Rng(2))])
    s3 = SdfObject.subtract(SdfObject.subtract(SdfObject.subtract(u, forms_far(0)), forms_far(1)), forms_far(2))
    assert host.scene(SdfScene(s3, syn.BACKGROUND)).info()["fast_path"] == 0
    s2 = SdfObject.subtract(SdfObject.subtract(u, forms_far(0)), forms_far(1))
    assert host.scene(SdfScene(s2, syn.BACKGROUND)).info()["fast_path"] == 3
    # strength outside the proven range of the fast exp disables the fast path, not the scene
    forms = [SdfForm.Primitive.sphere((x, 0, 0), 0.5) for x in range(4)]
    tiny = SdfScene(SdfObject.create(SdfMaterial.createSolid((1, 1, 1)), SdfForm.unionSmooth(0.001, forms)), syn.BACKGROUND)
    assert host.scene(tiny).info()["fast_path"] == 0
    # mixed primitive kinds in a smooth union: one run per kind in child order
    mixed = SdfScene(SdfObject.create(SdfMaterial.createSolid((1, 1, 1)), SdfForm.unionSmooth(0.25, [
        forms[0], forms[1], SdfForm.Primitive.capsule((0, 1, 0), (1, 1, 0), 0.2), forms[2]])), syn.BACKGROUND)
    assert host.scene(mixed).info()["n_instr"] == 5          # RUN(2 spheres), RUN(capsule), RUN(sphere), FIN, SETLEAF


def test_nested_unions_use_extra_slots(host):
    i = host.scene(syn.mixed_nested()[0]).info()
    assert i["n_grids"] == 3 and i["n_slots"] >= 3 and i["fast_path"] in (0, 2)      # 2: some combinator children run on demand


def test_errors_are_reported_not_swallowed(host):
    with pytest.raises(ft.FrayTracerError) as e:
        host.form_union([])
    assert e.value.code == ft._lib.FT_ERR_EMPTY and "No SdfObjects given" in str(e.value)       # SdfForm.fs:16
    with pytest.raises(ft.FrayTracerError):
        host.form_subtract(12345, 0)
    with pytest.raises(ft.FrayTracerError) as e:
        host.scene(syn.config1()[0]).render(0.01, 30.0, ft.ImageSize(8, 8), syn.default_camera())
    assert e.value.code == ft._lib.FT_ERR_NO_DEVICE                                             # no CPU fallback
    # every compute entry refuses a context without a GPU: there is no CPU path anywhere in the product
    ds = host.scene(syn.config1()[0])
    rays = np.zeros((4, 8), np.float32)
    for call in (lambda: ds.trace_rays(rays), lambda: ds.form_try_trace(rays), lambda: ds.object_try_trace(rays),
                 lambda: ds.eval_distance(np.zeros((4, 3), np.float32)), lambda: ds.collect_stats(),
                 lambda: host.math_eval(0, np.zeros(4, np.float32)), lambda: host.selftest_fastmath(),
                 lambda: ds.render(0.01, 30.0, ft.ImageSize(8, 8), syn.default_camera(), spp=4, max_bounces=2, spectral=4),
                 lambda: ds.render_colors(0.01, 30.0, ft.ImageSize(8, 8), syn.default_camera()),
                 lambda: ft.Image.toColors(2.2, None, np.zeros((4, 4, 3), np.float32), host),
                 lambda: ft.api.tone_map_device(host, 4096, 4, 4),
                 lambda: host.host_register(np.zeros(1024, np.float32)), lambda: host.host_unregister(np.zeros(4, np.float32))):
        with pytest.raises(ft.FrayTracerError) as e:
            call()
        assert e.value.code == ft._lib.FT_ERR_NO_DEVICE


def test_render_multi_validates_its_arguments_without_a_gpu(host):
    """ft_render_multi (multi.cpp): null arrays, width not a multiple of stripe_width x n, and host-only contexts are refused
    before any device or RCCL call — FT_ERR_INVALID / FT_ERR_UNSUPPORTED / FT_ERR_NO_DEVICE (there is no CPU path)"""
    import ctypes as C
    from fraytracer_amd import _lib
    lib = _lib.lib
    ds = host.scene(syn.config1()[0])
    cam = syn.default_camera()
    out = np.zeros((64, 8, 3), np.float32)
    st = _lib.Stats()

    def call(ctxs, scenes, n, W=64, H=8, stripe=16, outp=out.ctypes.data_as(C.c_void_p), camp=None):
        p = _lib.RenderParams(W, H, 0, W, stripe, 1, 0, 1, 0.01, 30.0, 0, 0.0, 0, 0)
        return lib.ft_render_multi(ctxs, scenes, n, camp if camp is not None else C.byref(cam._c), C.byref(p), outp, C.byref(st))

    two_ctx = (C.c_void_p * 2)(host._ctx, host._ctx)
    two_sc = (C.c_void_p * 2)(ds._scene, ds._scene)
    assert call(None, two_sc, 2) == _lib.FT_ERR_INVALID and call(two_ctx, None, 2) == _lib.FT_ERR_INVALID
    assert call(two_ctx, two_sc, 0) == _lib.FT_ERR_INVALID and call(two_ctx, two_sc, 2, outp=None) == _lib.FT_ERR_INVALID
    assert call(two_ctx, two_sc, 2, W=0) == _lib.FT_ERR_INVALID and call(two_ctx, two_sc, 2, stripe=-4) == _lib.FT_ERR_INVALID
    assert call(two_ctx, two_sc, 2, W=72) == _lib.FT_ERR_UNSUPPORTED and "multiple of stripe_width" in _lib.last_error()
    assert call((C.c_void_p * 2)(host._ctx, None), two_sc, 2) == _lib.FT_ERR_INVALID
    assert call(two_ctx, (C.c_void_p * 2)(ds._scene, None), 2) == _lib.FT_ERR_INVALID
    assert call(two_ctx, two_sc, 2) == _lib.FT_ERR_NO_DEVICE and "no CPU fallback" in _lib.last_error()
    assert call(two_ctx, two_sc, 1) == _lib.FT_ERR_NO_DEVICE


def test_host_side_mirrors_of_the_small_reference_functions(oracle):
    """ImageSize.getUniformPixelPos / Camera.uniformPixelToRay (Image.fs:17-23, Camera.fs:44-54) and Ray.get / move /
    setDirection (Ray.fs:6-15) of the Python mirror against the oracle's restatement, bit for bit"""
    import fraytracer_amd as ft
    cam = syn.default_camera()
    W, H = 96, 64
    pos = ft.ImageSize.getUniformPixelPos(ft.ImageSize(W, H))
    for x, y in ((0, 0), (48, 32), (95, 63), (17, 5), (80, 1)):
        got = ft.Camera.uniformPixelToRay(0.01, 30.0, cam, pos(x, y))
        want = oracle.pixel_ray(cam.as_array(), W, H, x, y, 0.01, 30.0)
        assert got.tobytes() == np.asarray(want, np.float32).tobytes(), (x, y)
    r = np.array([1, 2, 3, 0, 0.6, 0.8, 30, 0.01], np.float32)
    moved = ft.Ray.move(2.5, r)
    np.testing.assert_array_equal(moved, np.array([1, np.float32(2) + np.float32(0.6) * np.float32(2.5), np.float32(3) + np.float32(0.8) * np.float32(2.5), 0, 0.6, 0.8, 27.5, 0.01], np.float32))
    np.testing.assert_array_equal(ft.Ray.get(-0.01, r), r[0:3] + r[3:6] * np.float32(-0.01))
    np.testing.assert_array_equal(ft.Ray.setDirection((1, 0, 0), r)[3:6], [1, 0, 0])


def _weird_scenes():
    """scenes that stress the support-sphere rules: intersections whose later children are far bigger or far away, subtractions, nested smooth
    unions, thin and obtuse triangles, boxes, a smooth union of non-sphere children"""
    P = SdfForm.Primitive
    mat = SdfMaterial.createSolid((0.5, 0.5, 0.5))
    rng = syn.Rng(21)
    mk = lambda form: SdfScene(SdfObject.create(mat, form), syn.BACKGROUND, syn.program_lights())
    out = [mk(SdfForm.intersect([P.sphere((3, 0, 0), 1.0), P.sphere((0, 0, 0), 30.0), P.torus((40, 0, 0), (0, 1, 0), 5.0, 1.0)])),
           mk(SdfForm.subtract(P.capsule((-2, 0, 0), (2, 1, 0), 0.4), P.sphere((50, 0, 0), 45.0))),
           mk(SdfForm.unionSmooth(0.7, [SdfForm.unionSmooth(0.2, [P.sphere(rng.pointInBall(3.0), 0.3) for _ in range(6)]),
                                        P.torus((1, 2, 0), (0, 0, 1), 1.5, 0.2), P.triangle((0, 0, 0), (5, 0.01, 0), (-4, 0.02, 0.1), 0.05),
                                        P.box((-3, 0, 1), (0.5, 2.0, 0.1))])),
           mk(SdfForm.union([P.triangle((0, 0, 0), (1, 0, 0), (0.5, 1e-3, 0), 0.01), P.sphere((10, 10, 10), 0.1), P.capsule((0, -5, 0), (0, -5.001, 0), 0.2)])),
           # round 4: the sphere children of an intersect lend their own spheres — a small one far from child 0's centre, a second even smaller one later in the list
           mk(SdfForm.intersect([P.torus((0, 0, 0), (0, 1, 0), 6.0, 2.0), P.sphere((5.5, 0.5, 0), 1.5), P.capsule((5, 0, 0), (6, 1, 0), 3.0), P.sphere((6.0, 0, 0.2), 0.9)])),
           mk(SdfForm.subtract(SdfForm.intersect([SdfForm.union([P.torus(rng.pointInBall(3.0), rng.pointOnSphere(1.0), 0.5, 0.2) for _ in range(12)]), P.sphere((0.5, 0, 0), 2.0)]),
                               P.sphere((0, 1, -1), 1.2)))]
    return out


@pytest.mark.parametrize("make", [lambda: syn.config1()[0], lambda: syn.config2(boxes=True)[0], lambda: syn.config3(n=40)[0], lambda: syn.config5()[0],
                                  lambda: syn.console_scene(n=60)[0], lambda: syn.mixed_nested()[0], lambda: syn.combinator_zoo()[0]]
                         + [(lambda s=s: s) for s in _weird_scenes()])
def test_support_sphere_holds_every_small_distance(host, oracle, make):
    """FT_OPT_ESCAPE ends a ray as a miss once it can no longer come within epsilon of the scene's support sphere (scene.cpp supportOf).  That
    is exact iff  Distance(p) < tau  implies  dist(p, sphere) < tau: checked here against the ORACLE's Distance on 60 000 points outside the
    sphere, from just beyond it to 100 radii away (where the inequality is tightest: points near the sphere in every direction)."""
    scene = make()
    ds = host.scene(scene)
    cx, cy, cz, R = ds.support_sphere()
    assert R > 0 and np.isfinite(R)
    O = oracle.Oracle()
    obj = ft.realise(scene.Object, O, {})
    rng = np.random.default_rng(5)
    u = rng.normal(size=(60000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    beyond = np.concatenate([rng.uniform(0, 0.05, 20000), rng.uniform(0, 2.0, 20000), 10.0 ** rng.uniform(-2, 2, 20000)]) * R
    pts = (np.array([cx, cy, cz]) + u * (R + beyond)[:, None]).astype(np.float32)
    d = O.form_distance(O.object_form(obj), pts).astype(np.float64)
    outside = np.linalg.norm(pts.astype(np.float64) - np.array([cx, cy, cz]), axis=1) - R
    ok = np.isnan(d) | (d >= outside)                                  # NaN never compares below epsilon
    assert ok.all(), (pts[~ok][:5], d[~ok][:5], outside[~ok][:5])


def test_support_sphere_of_an_intersect_and_its_paddings(host):
    """Round 4: an intersect's support sphere is the smallest of child 0's and its sphere children's own spheres (Program.fs: the 3.5 sphere, not
    the tori's 4.7); the radius carries the evaluation padding and the drift padding of scene.cpp; a non-finite primitive constant ANYWHERE in
    the tree (also where the support rules never look) means no sphere at all."""
    P = SdfForm.Primitive
    mat = SdfMaterial.createSolid((0.5, 0.5, 0.5))
    cx, cy, cz, R = host.scene(syn.console_scene(n=200)[0]).support_sphere()
    assert (cx, cy, cz) == (0.0, 0.0, 0.0) and 3.5 + 0.0135 < R < 3.6          # r + padEval < escR (padDrift on top), far below the tori's sphere
    rng = syn.Rng(4)
    blobs = SdfObject.union([SdfObject.create(mat, P.sphere(rng.pointInBall(3.0), 0.5)) for _ in range(20)])
    two = SdfScene(SdfObject.intersect(blobs, [P.sphere((0.0, 0.0, 0.0), 9.0), P.sphere((0.5, 0.0, 0.0), 2.0)]), syn.BACKGROUND, [])
    cx, cy, cz, R = host.scene(two).support_sphere()
    assert (cx, cy, cz) == (0.5, 0.0, 0.0) and 2.0 < R < 2.1                     # the smallest sphere child wins, whatever its position in the list
    capsule_child = SdfScene(SdfObject.intersect(blobs, [P.capsule((0.0, 0.0, 0.0), (0.1, 0.0, 0.0), 0.3)]), syn.BACKGROUND, [])
    assert host.scene(capsule_child).support_sphere()[3] > 3.0                   # only sphere children are used (their pruning bound is the sphere itself)
    ball = SdfObject.create(mat, P.sphere((0.0, 0.0, 0.0), 0.5))
    for bad in (P.capsule((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.2), P.triangle((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (2.0, 2.0, 2.0), 0.1)):
        for obj in (SdfObject.subtract(ball, bad), SdfObject.intersect(ball, [P.sphere((0.1, 0.0, 0.0), 0.6), bad])):
            assert host.scene(SdfScene(obj, syn.BACKGROUND, [])).support_sphere()[3] < 0
    # far from the origin the paddings grow with the coordinates, the sphere stays usable
    far = SdfScene(SdfObject.create(mat, P.sphere((5000.0, -3000.0, 4000.0), 2.0)), syn.BACKGROUND, [])
    assert 2.0 + 12.0 < host.scene(far).support_sphere()[3] < 2.0 + 20.0


def test_support_sphere_radius_follows_the_stated_padding_rule(host):
    """escR = r + padEval + padDrift with padEval = 0.1 % r + 0.01 + 0.1 % |c|_1 and padDrift the root of
    (40 Rp / padDrift + 1) * 3 * 2^-23 * (|c|inf + 5 Rp) = padDrift / 4, Rp = escR — recomputed here independently (fixed-point iteration in double) for spheres of
    several sizes and positions: the flattener's radius must carry at least that much and at most 5 % more."""
    P = SdfForm.Primitive
    mat = SdfMaterial.createSolid((0.5, 0.5, 0.5))
    u3 = 3.0 * 2.0 ** -23
    for c, r in (((0.0, 0.0, 0.0), 1.0), ((0.0, 0.0, 0.0), 3.5), ((1.0, -2.0, 0.5), 0.05), ((300.0, 200.0, -100.0), 4.0), ((5000.0, -3000.0, 4000.0), 2.0), ((0.0, 0.0, 0.0), 900.0)):
        R = host.scene(SdfScene(SdfObject.create(mat, P.sphere(c, r)), syn.BACKGROUND, [])).support_sphere()[3]
        cinf, c1 = max(abs(v) for v in c), sum(abs(v) for v in c)
        pad_eval = 0.001 * r + 0.01 + 0.001 * c1
        pd = 0.0
        for _ in range(60):
            Rp = r + pad_eval + pd
            k = 4.0 * u3 * (cinf + 5.0 * Rp)
            pd = 0.5 * (k + (k * k + 160.0 * k * Rp) ** 0.5)
        need = r + pad_eval + pd
        assert need * (1 - 1e-6) <= R <= need * 1.05 + 1e-6, (c, r, R, need)
        Rp = R
        assert (40.0 * Rp / (R - r - pad_eval) + 1.0) * u3 * (cinf + 5.0 * Rp) <= (R - r - pad_eval) / 4.0 * (1 + 1e-9)      # the near-step budget really holds at the shipped radius


def test_support_sphere_is_refused_where_it_cannot_be_proved(host):
    P = SdfForm.Primitive
    mat = SdfMaterial.createSolid((0.5, 0.5, 0.5))
    weird = SdfScene(SdfObject.create(mat, SdfForm.unionSmooth(-0.5, [P.sphere((0, 0, 0), 1.0), P.sphere((2, 0, 0), 1.0)])), syn.BACKGROUND, [])
    assert host.scene(weird).support_sphere()[3] < 0
    big = SdfScene(SdfObject.create(mat, P.sphere((0, 0, 0), 3e20)), syn.BACKGROUND, [])
    assert host.scene(big).support_sphere()[3] < 0


def test_fuzz_scene_streams():
    """synthetic._fuzz_stream: Rng(s) and Rng(s + 1) read ONE splitmix stream one draw apart, so below FUZZ_INDEPENDENT_FROM neighbouring fuzz seeds build
    their scenes from overlapping draws (kept: the committed fuzz records name those seeds); from there on seeds are scrambled into unrelated streams."""
    a, b = syn.Rng(1000), syn.Rng(1001)
    a._next()
    assert [a._next() for _ in range(4)] == [b._next() for _ in range(4)]                  # the overlap the scramble removes
    base = syn.FUZZ_INDEPENDENT_FROM
    assert syn._fuzz_stream(base - 1) == base - 1 and syn._fuzz_stream(12345) == 12345     # recorded seeds keep their scenes
    s = [syn._fuzz_stream(base + i) for i in range(2000)]
    assert len(set(s)) == len(s) and all(0 <= v < 2 ** 64 for v in s)
    pos = sorted(s)                                                                        # Rng(seed) starts the shared stream at draw number `seed` (mod 2^64)
    assert min(q - p for p, q in zip(pos, pos[1:])) > 10 ** 6, "two scrambled seeds start within a scene's worth of draws of each other"
    late = lambda seed: (lambda r: (r[2].X, r[2].Y, r[3]))(syn.fuzz_scene(seed))            # image size and epsilon: drawn late in the stream
    assert late(base + 7) == late(base + 7)
