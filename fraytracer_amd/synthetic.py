"""Synthetic scenes of the BASELINE.json configs, built with the mirrored
scene-composition API.  Shared conventions come from the reference's console program:
camera src/FrayTracer.Console/Program.fs:16-22 (incl. `Lens.create 60.0f`, radians), epsilon 0.01
and ray length 30 (Program.fs:85,93), background 0.1 (Program.fs:78), the directional light of
Program.fs:80 and the random factories of Program.fs:32-65.

The scene RNG is a portable splitmix64 (NOT System.Random), narrowed to float32 like
Random.fs:9; pointInBall / pointOnSphere / range follow Random.fs:11, 27-40.
"""
import numpy as np

from .api import SdfForm, SdfMaterial, SdfObject, SdfLight, SdfScene, Lens, Camera, ImageSize

F = np.float32
EPSILON = 0.01
RAY_LENGTH = 30.0
BACKGROUND = (0.1, 0.1, 0.1)


class Rng:
    def __init__(self, seed):
        self.s = (int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF

    def _next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def range_01(self):                                   # Random.fs:9
        return F((self._next() >> 40) * (1.0 / (1 << 24)))

    def range(self, lo, hi):                              # Random.fs:11
        lo, hi = F(lo), F(hi)
        return F(lo + F(self.range_01() * F(hi - lo)))

    def _vec(self):
        return np.array([self.range(-1, 1), self.range(-1, 1), self.range(-1, 1)], F)

    def pointInBall(self, radius):                        # Random.fs:27-32
        while True:
            v = self._vec()
            if F(F(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) <= F(1.0):
                return v * F(radius)

    def pointOnSphere(self, radius):                      # Random.fs:34-40
        while True:
            v = self._vec()
            l = F(F(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            if F(0.01) <= l <= F(1.0):
                return v / np.sqrt(l) * F(radius)


def default_camera():
    """Program.fs:16-22"""
    return Camera.lookAt(Position=(0.0, 0.0, -10.0), LookAt=(0.0, 0.0, 0.0), Up=(0.0, 1.0, 0.0), Lens=Lens.create(60.0))


def program_lights():
    """Program.fs:79-82"""
    return [SdfLight.directional((-0.5, -1.0, 1.0), (0.5, 0.5, 0.5)),
            SdfLight.point((-0.5, 0.0, -2.0), (10.0, 0.0, 0.0))]


def _material(rng):                                        # Program.fs:28-30
    return SdfMaterial.createSolid((rng.range_01(), rng.range_01(), rng.range_01()))


def random_sphere(rng):                                    # Program.fs:32-37
    f = SdfForm.Primitive.sphere(Center=rng.pointInBall(4.0), Radius=rng.range(0.3, 1.0))
    return SdfObject.create(_material(rng), f)


def random_capsule(rng):                                   # Program.fs:39-46
    c = rng.pointInBall(4.0)
    f = SdfForm.Primitive.capsule(From=c, To=c + rng.pointOnSphere(rng.range(0.5, 2.0)), Radius=rng.range(0.1, 0.3))
    return SdfObject.create(_material(rng), f)


def random_torus(rng):                                     # Program.fs:48-55
    f = SdfForm.Primitive.torus(Center=rng.pointInBall(4.0), Normal=rng.pointOnSphere(1.0),
                                MajorRadius=rng.range(0.1, 0.4), MinorRadius=rng.range(0.1, 0.3))
    return SdfObject.create(_material(rng), f)


def random_triangle(rng):                                  # Program.fs:57-65
    v1 = rng.pointInBall(4.0)
    f = SdfForm.Primitive.triangle(V1=v1, V2=v1 + rng.pointOnSphere(rng.range(0.2, 0.6)),
                                   V3=v1 + rng.pointOnSphere(rng.range(0.2, 0.6)), Radius=rng.range(0.1, 0.3))
    return SdfObject.create(_material(rng), f)


def random_box(rng):                                       # EXTENSION
    f = SdfForm.Primitive.box(Center=rng.pointInBall(4.0), HalfExtent=(rng.range(0.2, 0.7), rng.range(0.2, 0.7), rng.range(0.2, 0.7)))
    return SdfObject.create(_material(rng), f)


# ---- BASELINE.json configs -------------------------------------------------------------------------
def config1():
    """C1: single SDF sphere, 256x256, primary rays only."""
    obj = SdfObject.create(SdfMaterial.createSolid((0.8, 0.8, 0.8)), SdfForm.Primitive.sphere((0.0, 0.0, 0.0), 1.0))
    return SdfScene(obj, BACKGROUND, []), ImageSize(256, 256)


def config2(seed=2, boxes=False, size=1024):
    """C2: SdfObject.union of 32 leaves (16 spheres + 16 capsules; boxes=True swaps the capsules for
    EXTENSION boxes), one directional light (its shadow rays are the secondary rays)."""
    rng = Rng(seed)
    objs = []
    for _ in range(16):
        objs.append(random_sphere(rng))
        objs.append(random_box(rng) if boxes else random_capsule(rng))
    lights = [SdfLight.directional((-0.5, -1.0, 1.0), (0.5, 0.5, 0.5))]
    return SdfScene(SdfObject.union(objs), BACKGROUND, lights), ImageSize(size, size)


def config3(seed=3, n=256, size=4096, strength=0.25):
    """C3 (the metric's config): unionSmooth 0.25 of 256 spheres, one solid colour, directional light."""
    rng = Rng(seed)
    forms = [SdfForm.Primitive.sphere(Center=rng.pointInBall(4.0), Radius=rng.range(0.1, 0.5)) for _ in range(n)]
    obj = SdfObject.create(SdfMaterial.createSolid((0.9, 0.6, 0.3)), SdfForm.unionSmooth(strength, forms))
    lights = [SdfLight.directional((-0.5, -1.0, 1.0), (0.5, 0.5, 0.5))]
    return SdfScene(obj, BACKGROUND, lights), ImageSize(size, size)


def config4(seed=3):
    """C4: the C3 scene at 8192x8192 (column-tiled over 1/2/4/8 GPUs)."""
    scene, _ = config3(seed)
    return scene, ImageSize(8192, 8192)


def config5(seed=5, size=2048, n_glass=6, n_solid=10, ior=1.5, dispersion=0.02):
    """C5 (EXTENSION only, BASELINE.json config 5): glass spheres, tori and one smooth glass blob in front of
    solid spheres; render with max_bounces = 4, spectral = 16, spp = 16."""
    rng = Rng(seed)
    objs = []
    for i in range(n_glass):
        tint = (rng.range(0.85, 1.0), rng.range(0.85, 1.0), rng.range(0.85, 1.0))
        glass = SdfMaterial.createGlass(tint, ior, dispersion)
        c = rng.pointInBall(3.0)
        c = (c[0], c[1], c[2] * 0.5 - 1.5)                   # glass sits in front
        if i % 3 == 0:
            form = SdfForm.Primitive.torus(c, rng.pointOnSphere(1.0), rng.range(0.7, 1.1), rng.range(0.2, 0.35))
        elif i % 3 == 1:
            form = SdfForm.Primitive.sphere(c, rng.range(0.6, 1.1))
        else:
            kids = [SdfForm.Primitive.sphere((c[0] + rng.range(-0.6, 0.6), c[1] + rng.range(-0.6, 0.6), c[2] + rng.range(-0.3, 0.3)),
                                             rng.range(0.35, 0.6)) for _ in range(4)]
            form = SdfForm.unionSmooth(0.2, kids)
        objs.append(SdfObject.create(glass, form))
    for _ in range(n_solid):
        c = rng.pointInBall(4.0)
        objs.append(SdfObject.create(_material(rng), SdfForm.Primitive.sphere((c[0], c[1], abs(c[2]) + 1.0), rng.range(0.4, 1.0))))
    return SdfScene(SdfObject.union(objs), BACKGROUND, program_lights()), ImageSize(size, size)


def console_like(seed=19, n=1000, size=1000, factory=random_torus):
    """The structure of Program.fs:67-83 — subtract(intersect(union of n random tori, sphere r3.5),
    sphere r2.5) with the two lights — on the portable RNG (the System.Random-seeded original is
    reproduced separately, see fraytracer_amd/dotnet_random.py)."""
    rng = Rng(seed)
    union = SdfObject.union([factory(rng) for _ in range(n)])
    obj = SdfObject.subtract(
        SdfObject.intersect(union, [SdfForm.Primitive.sphere((0.0, 0.0, 0.0), 3.5)]),
        SdfForm.Primitive.sphere((-0.5, 1.0, -2.0), 2.5))
    return SdfScene(obj, BACKGROUND, program_lights()), ImageSize(size, size)


def console_scene(seed=19, n=1000, size=1000):
    """The reference's own scene: Program.fs:14-83 with `System.Random(19)` (fraytracer_amd/dotnet_random.py,
    re-implemented from memory — best effort, see that module), 1000 random tori, draw order of
    Program.fs:28-30, 48-55: form fields first, then the three material draws."""
    from .dotnet_random import Random
    rng = Random(seed)
    union = SdfObject.union([random_torus(rng) for _ in range(n)])
    obj = SdfObject.subtract(
        SdfObject.intersect(union, [SdfForm.Primitive.sphere((0.0, 0.0, 0.0), 3.5)]),
        SdfForm.Primitive.sphere((-0.5, 1.0, -2.0), 2.5))
    return SdfScene(obj, BACKGROUND, program_lights()), ImageSize(size, size)


def mixed_nested(seed=7):
    """A small scene touching every combinator and primitive, including combinator children of a
    union (material resolution through nested object unions)."""
    rng = Rng(seed)
    blob = SdfObject.create(_material(rng), SdfForm.unionSmooth(0.3, [
        SdfForm.Primitive.sphere(rng.pointInBall(2.0), rng.range(0.3, 0.8)) for _ in range(5)]))
    inner_union = SdfObject.union([random_sphere(rng), random_torus(rng), random_capsule(rng)])
    carved = SdfObject.subtract(random_sphere(rng), SdfForm.Primitive.sphere(rng.pointInBall(3.0), 0.8))
    clipped = SdfObject.intersect(random_torus(rng), [SdfForm.Primitive.sphere((0, 0, 0), 4.0),
                                                      SdfForm.union([SdfForm.Primitive.sphere((0, 0, 0), 3.9), SdfForm.Primitive.sphere((1, 0, 0), 3.9)])])
    objs = [blob, inner_union, carved, clipped] + [random_triangle(rng) for _ in range(4)] + [random_capsule(rng) for _ in range(3)]
    return SdfScene(SdfObject.union(objs), BACKGROUND, program_lights()), ImageSize(96, 96)


def combinator_zoo(seed=11):
    """Every flattener path the other scenes miss: smooth unions with non-sphere runs and with combinator
    children (SMOOTH_ADD), intersect runs of several primitives of one kind, a union whose first-sorted
    candidates are combinators, form-level unions under create, three lights."""
    rng = Rng(seed)
    P = SdfForm.Primitive

    def sph(r=2.5, lo=0.3, hi=0.8): return P.sphere(rng.pointInBall(r), rng.range(lo, hi))
    def cap():
        c = rng.pointInBall(2.5); return P.capsule(c, c + rng.pointOnSphere(rng.range(0.5, 1.5)), rng.range(0.1, 0.3))
    def tor(): return P.torus(rng.pointInBall(2.5), rng.pointOnSphere(1.0), rng.range(0.3, 0.6), rng.range(0.1, 0.2))
    def tri():
        v = rng.pointInBall(2.5); return P.triangle(v, v + rng.pointOnSphere(0.8), v + rng.pointOnSphere(0.8), rng.range(0.05, 0.2))

    blob_mixed = SdfForm.unionSmooth(0.2, [sph(), sph(), cap(), cap(), cap(), tor(), SdfForm.subtract(sph(2.0, 0.8, 1.0), sph(2.0, 0.4, 0.6)),
                                           tri(), tri(), sph(), SdfForm.intersect([sph(1.0, 1.0, 1.2), sph(1.0, 1.0, 1.2)]), P.box(rng.pointInBall(2.0), (0.3, 0.4, 0.2))])
    clipped = SdfForm.intersect([SdfForm.union([tor(), tor(), cap(), tri()]), P.sphere((0, 0, 0), 3.0), P.sphere((0.2, 0, 0), 3.1),
                                 P.box((0, 0, 0), (2.5, 2.5, 2.5)), P.box((0.1, 0, 0), (2.6, 2.4, 2.5)), cap()])
    objs = [SdfObject.create(_material(rng), blob_mixed), SdfObject.create(_material(rng), clipped),
            SdfObject.union([SdfObject.create(_material(rng), SdfForm.unionSmooth(0.3, [sph(), sph(), sph()])), random_sphere(rng)]),
            random_torus(rng), random_triangle(rng)]
    lights = program_lights() + [SdfLight.point((3.0, 4.0, -6.0), (0.0, 20.0, 30.0))]
    return SdfScene(SdfObject.union(objs), BACKGROUND, lights), ImageSize(96, 96)


def combinator_crowd(seed=5, n=300, size=1024):
    """A union of `n` combinator objects (subtracted spheres, small smooth blobs, clipped tori): every child is a
    sub-program the candidate walk runs on demand (ft_device.h FT_PR_CALL)."""
    rng = Rng(seed)
    P = SdfForm.Primitive
    objs = []
    for k in range(n):
        c = rng.pointInBall(4.0)
        if k % 3 == 0:
            objs.append(SdfObject.subtract(SdfObject.create(_material(rng), P.sphere(c, rng.range(0.4, 0.8))), P.sphere(c + rng.pointOnSphere(0.4), 0.35)))
        elif k % 3 == 1:
            objs.append(SdfObject.create(_material(rng), SdfForm.unionSmooth(0.2, [P.sphere(c + rng.pointOnSphere(0.3), rng.range(0.2, 0.4)) for _ in range(5)])))
        else:
            objs.append(SdfObject.intersect(SdfObject.create(_material(rng), P.torus(c, rng.pointOnSphere(1.0), 0.6, 0.2)), [P.sphere(c, 0.7), P.box(c, (0.6, 0.5, 0.7))]))
    return SdfScene(SdfObject.union(objs), BACKGROUND, program_lights()), ImageSize(size, size)


def fuzz_scene(seed, big=False):
    """A random scene for differential testing (HIP path against the CPU oracle): random combinator trees over every primitive,
    solid and (EXTENSION) glass materials, 0-3 lights, a random camera and random render parameters.
    `big`: unions of 60-400 objects (large grids, the device-side grid build) and larger images.
    Returns (scene, camera, size, epsilon, extension kwargs)."""
    scene, cam, size, eps, _, ext = _fuzz_scene(seed, big, False)
    return scene, cam, size, eps, ext


def fuzz_scene_edge(seed):
    """fuzz_scene at the edge of the exact shortcuts' arguments (round 4): the whole scene (camera and lights with it) shifted by up to 7000
    from the origin, epsilon down to 1e-5 (below the float spacing out there) and ray length up to 1000.
    Returns (scene, camera, size, epsilon, length, extension kwargs)."""
    return _fuzz_scene(seed, False, True)


FUZZ_INDEPENDENT_FROM = 10_000_000


def _fuzz_stream(seed):
    """Rng(s) and Rng(s + 1) are ONE splitmix stream read one draw apart, so scenes of neighbouring seeds are built from overlapping draws (an object
    that makes the reference throw sits in every scene of a few thousand consecutive seeds).  Seeds from FUZZ_INDEPENDENT_FROM on are scrambled first:
    unrelated streams.  Smaller seeds keep the scenes the committed fuzz records (profiles/) name."""
    if seed < FUZZ_INDEPENDENT_FROM: return seed
    z = (seed * 0xD6E8FEB86659FD93 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 32)) * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 32)) * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 32)


def _fuzz_scene(seed, big, edge):
    rng = Rng((0xB16 if big else 0xF00D) + _fuzz_stream(seed))
    P0 = SdfForm.Primitive
    off = np.zeros(3, F)
    edge_eps, edge_len = None, RAY_LENGTH
    if edge:
        er = Rng(0xED6E + _fuzz_stream(seed))
        off = np.array(((0.0, 0.0, 0.0), (5000.0, -3000.0, 4000.0), (300.0, 200.0, -100.0), (-40.0, 0.0, 25.0))[int(er.range_01() * 4) % 4], F)
        edge_eps = (1e-2, 1e-3, 1e-4, 1e-5)[int(er.range_01() * 4) % 4]
        edge_len = (30.0, 1000.0)[int(er.range_01() * 2) % 2]

    def sh(v): return np.asarray(v, F) + off                      # every position of the scene goes through the offset

    class P:                                                      # the primitive constructors, shifted
        sphere = staticmethod(lambda c, r: P0.sphere(sh(c), r))
        capsule = staticmethod(lambda a, b, r: P0.capsule(sh(a), sh(b), r))
        torus = staticmethod(lambda c, n, R, r: P0.torus(sh(c), n, R, r))
        triangle = staticmethod(lambda a, b, c, r: P0.triangle(sh(a), sh(b), sh(c), r))
        box = staticmethod(lambda c, h: P0.box(sh(c), h))

    def pick(n): return int(rng.range_01() * n) % n

    def prim(spread=3.0):
        k = pick(5)
        c = rng.pointInBall(spread)
        if k == 0: return P.sphere(c, rng.range(0.2, 1.2))
        if k == 1: return P.capsule(c, c + rng.pointOnSphere(rng.range(0.3, 2.0)), rng.range(0.05, 0.5))
        if k == 2: return P.torus(c, rng.pointOnSphere(1.0), rng.range(0.3, 1.2), rng.range(0.05, 0.3))
        if k == 3: return P.triangle(c, c + rng.pointOnSphere(rng.range(0.4, 1.5)), c + rng.pointOnSphere(rng.range(0.4, 1.5)), rng.range(0.03, 0.3))
        return P.box(c, (rng.range(0.1, 0.9), rng.range(0.1, 0.9), rng.range(0.1, 0.9)))

    def form(depth):
        k = pick(7) if depth > 0 else 0
        if k <= 1: return prim()
        if k == 2:                                              # smooth union: all spheres (fast runs) or mixed
            n = 2 + pick(9)
            if pick(2): return SdfForm.unionSmooth(rng.range(0.05, 0.6), [P.sphere(rng.pointInBall(2.5), rng.range(0.2, 0.9)) for _ in range(n)])
            return SdfForm.unionSmooth(rng.range(0.05, 0.6), [form(depth - 1) for _ in range(n)])
        if k == 3: return SdfForm.union([form(depth - 1) for _ in range(2 + pick(5))])
        if k == 4: return SdfForm.subtract(form(depth - 1), prim(2.0))
        if k == 5: return SdfForm.intersect([form(depth - 1)] + [P.sphere(rng.pointInBall(1.0), rng.range(2.0, 4.0)) for _ in range(1 + pick(3))])
        return SdfForm.intersect([form(depth - 1), prim(1.5), prim(1.5)])

    def material():
        if pick(5) == 0:
            return SdfMaterial.createGlass((rng.range(0.7, 1.0), rng.range(0.7, 1.0), rng.range(0.7, 1.0)), rng.range(1.1, 2.2), rng.range(0.0, 0.06))
        return _material(rng)

    def obj(depth):
        k = pick(6) if depth > 0 else 0
        if k <= 2: return SdfObject.create(material(), form(2))
        if k == 3: return SdfObject.union([obj(depth - 1) for _ in range(2 + pick(4))])
        if k == 4: return SdfObject.subtract(obj(depth - 1), prim(2.0))
        return SdfObject.intersect(obj(depth - 1), [P.sphere(rng.pointInBall(1.0), rng.range(2.5, 4.5))])

    top = pick(4)
    if big: root = SdfObject.union([SdfObject.create(material(), prim(5.0)) if pick(8) else obj(1) for _ in range(60 + pick(341))])
    elif top == 0: root = obj(1)
    elif top == 1:
        strength = rng.range(0.1, 0.5)
        if seed % 2 == 0: strength = (0.25, 0.5, 2.0, 0.125)[(seed // 2) % 4]     # -1/strength a power of two: the near loop's output-modifier variants (no extra draw: odd seeds keep their scenes)
        root = SdfObject.create(material(), SdfForm.unionSmooth(strength, [P.sphere(rng.pointInBall(3.0), rng.range(0.2, 0.8)) for _ in range(3 + pick(40))]))
    else: root = SdfObject.union([obj(2) for _ in range(2 + pick(14))])
    lights = []
    for _ in range(pick(4)):
        if pick(2): lights.append(SdfLight.directional(rng.pointOnSphere(1.0), (rng.range(0.1, 1.0), rng.range(0.1, 1.0), rng.range(0.1, 1.0))))
        else: lights.append(SdfLight.point(sh(rng.pointOnSphere(rng.range(5.0, 9.0))), (rng.range(5, 40), rng.range(5, 40), rng.range(5, 40))))
    scene = SdfScene(root, (rng.range(0.0, 0.3), rng.range(0.0, 0.3), rng.range(0.0, 0.3)), lights)
    if pick(3) == 0: cam = default_camera() if not edge else Camera.lookAt(Position=sh((0.0, 0.0, -10.0)), LookAt=sh((0.0, 0.0, 0.0)), Up=(0.0, 1.0, 0.0), Lens=Lens.create(60.0))
    else: cam = Camera.lookAt(Position=sh(rng.pointOnSphere(rng.range(7.0, 12.0))), LookAt=sh(rng.pointInBall(1.0)), Up=(0.0, 1.0, 0.0), Lens=Lens.create(rng.range(59.0, 61.5)))
    size = ImageSize(24 + 8 * pick(5), 24 + 8 * pick(5)) if not big else ImageSize(64 + 16 * pick(5), 64 + 16 * pick(5))
    eps = (0.01, 0.003, 0.03)[pick(3)]
    if edge_eps is not None: eps = edge_eps
    ext = {}
    if pick(2):
        spp = (1, 4)[pick(2)]
        ext = dict(spp=spp, ao_samples=(0, 0, 3, 7)[pick(4)], ao_radius=float(rng.range(0.3, 1.5)), max_bounces=(0, 2, 4, 7)[pick(4)],
                   spectral=(0, 1, 2, 4)[pick(4)] if spp == 4 else (0, 1)[pick(2)])
    return scene, cam, size, eps, edge_len, ext
