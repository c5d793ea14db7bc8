// scene.cpp — scene construction and flattening (host, float32, reference operation order).
// Compile with -ffp-contract=off.
#include "scene.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>

namespace ft {

// ------------------------------------------------------------------------------------------------
// SdfBoundary.fs:7-22 / 29-49 / 62-67
// ------------------------------------------------------------------------------------------------
Boundary boundaryUnion(Boundary a, Boundary b) {
    const f3 diff = b.center - a.center;
    const float distance = ft_length(diff);
    if (distance + b.radius <= a.radius) return a;
    if (distance + a.radius <= b.radius) return b;
    const f3 dir = diff / distance;
    const f3 pa = a.center - dir * a.radius;
    const f3 pb = b.center + dir * b.radius;
    return Boundary{(pa + pb) * 0.5f, ft_distance(pa, pb) * 0.5f};
}

Boundary boundaryIntersection(Boundary a, Boundary b) {
    const f3 diff = b.center - a.center;
    const float distance = ft_length(diff);
    if (distance + b.radius <= a.radius) return b;
    if (distance + a.radius <= b.radius) return a;
    const f3 dir = diff / distance;
    const f3 pa = a.center + dir * a.radius;
    const f3 pb = b.center - dir * b.radius;
    const float d2 = distance * distance;
    const float aR2 = a.radius * a.radius;
    const float bR2 = b.radius * b.radius;
    // SdfBoundary.fs:48: (d2 - bR2 + aR2) is NOT squared in the reference; reproduced, not fixed.
    const float radius = sqrtf(4.0f * d2 * aR2 - (d2 - bR2 + aR2)) / (2.0f * distance);
    return Boundary{(pa + pb) * 0.5f, radius};
}

static inline float minDistance(const Boundary& x, f3 p) { return ft_distance(x.center, p) - x.radius; }  // :62
static inline float maxDistance(const Boundary& x, f3 p) { return ft_distance(x.center, p) + x.radius; }  // :63

// ------------------------------------------------------------------------------------------------
// SdfBoundary.buildSpatialLookup (SdfBoundary.fs:225-274).  Quirks kept: all three counts are
// derived from aabbSize.X (:237-239); upperBound adds half a cell diagonal once (:253).
// ------------------------------------------------------------------------------------------------
std::shared_ptr<HostGrid> buildSpatialLookup(const std::vector<Boundary>& bounds, std::string& err, GridFiller* filler) {
    auto g = std::make_shared<HostGrid>();
    const size_t n = bounds.size();
    f3 aabbMin = bounds[0].center - splat3(bounds[0].radius);
    f3 aabbMax = bounds[0].center + splat3(bounds[0].radius);
    float sum = bounds[0].radius;
    for (size_t i = 1; i < n; ++i) {
        aabbMin = ft_vmin(aabbMin, bounds[i].center - splat3(bounds[i].radius));
        aabbMax = ft_vmax(aabbMax, bounds[i].center + splat3(bounds[i].radius));
        sum = sum + bounds[i].radius;
    }
    const float countSize = 1.5f * (sum / (float)(int)n);
    const f3 aabbSize = aabbMax - aabbMin;
    const int c = std::max(1, ft_ceiling_i(aabbSize.x / countSize));
    if (c > 1024 || c < 1) { err = "union: grid resolution out of range (degenerate boundaries?)"; return nullptr; }
    g->count[0] = c; g->count[1] = c; g->count[2] = c;
    g->aabbMin = aabbMin;
    g->cellSize = aabbSize / mk3((float)c, (float)c, (float)c);
    g->cellSizeInv = splat3(1.0f) / g->cellSize;
    const size_t ncells = (size_t)c * c * c;
    if (ncells * 1 > (size_t)64 << 20) { err = "union: grid too large"; return nullptr; }
    g->centers.resize(ncells);
    g->cellStart.assign(ncells + 1, 0);
    const float halfDiag = ft_length(g->cellSize * 0.5f);
    if (filler) {                                                       // device-side per-cell build (same arithmetic)
        std::string ferr;
        if (filler->fill(*g, bounds, halfDiag, ferr)) return g;
        if (!ferr.empty()) { err = ferr; return nullptr; }              // a real error (NaN boundary, empty cell), not a decline
        g->centers.assign(ncells, mk3(0, 0, 0)); g->cellStart.assign(ncells + 1, 0); g->items.clear();
    }

    // one x-slab of cells per task (the reference parallelises the same build over y, Array3D.fs:4-14);
    // Distance(center_i, cellCenter) is computed once and reused for getMaxDistance / getMinDistance —
    // the same two operations the reference performs, so the values are bit-identical.
    std::vector<std::vector<FtItem>> slabItems(c);
    std::vector<std::vector<uint32_t>> slabCounts(c);
    std::vector<std::string> slabErr(c);
    auto buildSlab = [&](int x) {
        std::vector<float> dist(n);
        std::vector<FtItem> cellItems;
        slabCounts[x].reserve((size_t)c * c);
        for (int y = 0; y < c; ++y)
        for (int z = 0; z < c; ++z) {
            const size_t ci = ((size_t)x * c + y) * c + z;
            const f3 center = aabbMin + g->cellSize * 0.5f + g->cellSize * mk3((float)x, (float)y, (float)z);
            g->centers[ci] = center;
            for (size_t i = 0; i < n; ++i) dist[i] = ft_distance(bounds[i].center, center);
            float m = dist[0] + bounds[0].radius;                                   // Seq.min of getMaxDistance (:249-252)
            for (size_t i = 1; i < n; ++i) { const float v = dist[i] + bounds[i].radius; if (v < m) m = v; }
            const float upperBound = m + halfDiag;                                  // :253
            cellItems.clear();
            for (size_t i = 0; i < n; ++i) {
                const float lo = dist[i] - bounds[i].radius;                        // getMinDistance (:257)
                if (lo < upperBound) cellItems.push_back(FtItem{lo, (uint32_t)i});
            }
            if (cellItems.empty()) { slabErr[x] = "union: a lookup cell has no candidates (the reference would throw at Items.[0])"; return; }
            for (const FtItem& it : cellItems)          // the device loop's early exit relies on a totally ordered list
                if (it.lowerBound != it.lowerBound) { slabErr[x] = "union: NaN boundary"; return; }
            // Array.sortInPlaceBy (:267-268) is unstable in .NET; ties resolved by input order here.
            std::stable_sort(cellItems.begin(), cellItems.end(),
                             [](const FtItem& a, const FtItem& b) { return a.lowerBound < b.lowerBound; });
            slabCounts[x].push_back((uint32_t)cellItems.size());
            slabItems[x].insert(slabItems[x].end(), cellItems.begin(), cellItems.end());
        }
    };
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (hw > 1 && ncells * n > 200000 && c > 1) {
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < std::min<unsigned>(hw, (unsigned)c); ++t)
            pool.emplace_back([&]() { for (int x; (x = next.fetch_add(1)) < c;) buildSlab(x); });
        for (auto& t : pool) t.join();
    } else {
        for (int x = 0; x < c; ++x) buildSlab(x);
    }
    // stitch the slabs together in x order: CSR offsets, then the items behind one another
    size_t ci = 0, total = 0;
    for (int x = 0; x < c; ++x) {
        if (!slabErr[x].empty()) { err = slabErr[x]; return nullptr; }
        for (uint32_t cnt : slabCounts[x]) { g->cellStart[ci++] = (uint32_t)total; total += cnt; }
    }
    g->items.reserve(total);
    for (int x = 0; x < c; ++x) g->items.insert(g->items.end(), slabItems[x].begin(), slabItems[x].end());
    g->cellStart[ncells] = (uint32_t)g->items.size();
    return g;
}

// ------------------------------------------------------------------------------------------------
// Primitives: host precomputes what each closure captures (SdfForm.fs:125-268)
// ------------------------------------------------------------------------------------------------
static void put3(std::vector<float>& v, f3 a, float w = 0.0f) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); v.push_back(w); }

int Builder::sphere(f3 c, float r) {                                   // SdfForm.fs:125-135
    HostForm f; f.kind = HostForm::SPHERE;
    put3(f.params, c, r);
    f.boundary = Boundary{c, r};
    forms.push_back(f); return (int)forms.size() - 1;
}

int Builder::capsule(f3 from, f3 to, float r) {                        // SdfForm.fs:145-170
    HostForm f; f.kind = HostForm::CAPSULE;
    const f3 dir = to - from;                                          // :148
    const f3 dirInv = dir / ft_length2(dir);                           // :149, Math.fs:69
    put3(f.params, from, r); put3(f.params, dir); put3(f.params, dirInv);
    f.boundary = Boundary{ft_lerp(from, to, 0.5f), r + ft_distance(from, to) * 0.5f};   // :166-169
    forms.push_back(f); return (int)forms.size() - 1;
}

int Builder::torus(f3 c, f3 nIn, float R, float r) {                   // SdfForm.fs:181-203
    HostForm f; f.kind = HostForm::TORUS;
    const f3 n = ft_normalize(nIn);                                    // :182-185
    const float planeD = -(ft_dot(c, n));                              // :188
    put3(f.params, c, R); put3(f.params, n, r); put3(f.params, mk3(planeD, 0, 0));
    f.boundary = Boundary{c, R + r};                                   // :199-202
    forms.push_back(f); return (int)forms.size() - 1;
}

int Builder::triangle(f3 v1, f3 v2, f3 v3, float r) {                  // SdfForm.fs:214-268
    HostForm f; f.kind = HostForm::TRIANGLE;
    const f3 v21 = v2 - v1, v32 = v3 - v2, v13 = v1 - v3;              // :216-220
    const f3 v21i = v21 / ft_length2(v21), v32i = v32 / ft_length2(v32), v13i = v13 / ft_length2(v13);
    const f3 nor = ft_normalize(ft_cross(v21, v13));                   // :222
    const f3 n21 = ft_normalize(ft_cross(v21, nor));                   // :223
    const f3 n32 = ft_normalize(ft_cross(v32, nor));                   // :224
    const f3 n13 = ft_normalize(ft_cross(v13, nor));                   // :225
    put3(f.params, v1, r); put3(f.params, v2); put3(f.params, v3);
    put3(f.params, v21); put3(f.params, v32); put3(f.params, v13);
    put3(f.params, v21i); put3(f.params, v32i); put3(f.params, v13i);
    put3(f.params, nor); put3(f.params, n21); put3(f.params, n32); put3(f.params, n13);
    // boundary = circumsphere + Radius (:252-263)
    const float areaInv = 0.5f / ft_length2(ft_cross(v1 - v2, v2 - v3));
    const float w1 = ft_length2(v2 - v3) * ft_dot(v1 - v2, v1 - v3) * areaInv;
    const float w2 = ft_length2(v1 - v3) * ft_dot(v2 - v1, v2 - v3) * areaInv;
    const float w3 = 1.0f - w1 - w2;
    const f3 center = w1 * v1 + w2 * v2 + w3 * v3;
    const float radius = ft_length(v21) * ft_length(v32) * ft_length(v13) / 2.0f / ft_length(ft_cross(v21, v32)) + r;
    f.boundary = Boundary{center, radius};
    forms.push_back(f); return (int)forms.size() - 1;
}

int Builder::box(f3 c, f3 half) {                                      // EXTENSION (not in the reference)
    HostForm f; f.kind = HostForm::BOX;
    put3(f.params, c); put3(f.params, half);
    f.boundary = Boundary{c, ft_length(half)};
    forms.push_back(f); return (int)forms.size() - 1;
}

// ------------------------------------------------------------------------------------------------
// Combinators (SdfForm.fs:14-91): a single child is returned unchanged, as in the reference.
// ------------------------------------------------------------------------------------------------
static bool checkKids(const Builder& b, const int* kids, int n, std::string& err) {
    if (n <= 0 || !kids) { err = "No SdfObjects given."; return false; }
    for (int i = 0; i < n; ++i) if (!b.okForm(kids[i])) { err = "invalid form handle"; return false; }
    return true;
}

int Builder::formUnion(const int* kids, int n) {                       // SdfForm.fs:14-40
    if (!checkKids(*this, kids, n, err)) return n <= 0 ? FT_ERR_EMPTY_ : FT_ERR_INVALID_;
    if (n == 1) return kids[0];
    HostForm f; f.kind = HostForm::UNION; f.kids.assign(kids, kids + n);
    std::vector<Boundary> bs; for (int i = 0; i < n; ++i) bs.push_back(forms[kids[i]].boundary);
    f.grid = buildSpatialLookup(bs, err, gridFiller);
    if (!f.grid) return FT_ERR_UNSUPPORTED_;
    Boundary acc = bs[0];                                              // Seq.reduce union (:36-39)
    for (int i = 1; i < n; ++i) acc = boundaryUnion(acc, bs[i]);
    f.boundary = acc;
    forms.push_back(f); return (int)forms.size() - 1;
}

int Builder::formSubtract(int a, int b) {                              // SdfForm.fs:42-49
    if (!okForm(a) || !okForm(b)) { err = "invalid form handle"; return FT_ERR_INVALID_; }
    HostForm f; f.kind = HostForm::SUBTRACT; f.kids = {a, b};
    f.boundary = forms[a].boundary;
    forms.push_back(f); return (int)forms.size() - 1;
}

int Builder::formIntersect(const int* kids, int n) {                   // SdfForm.fs:51-67
    if (!checkKids(*this, kids, n, err)) return n <= 0 ? FT_ERR_EMPTY_ : FT_ERR_INVALID_;
    if (n == 1) return kids[0];
    HostForm f; f.kind = HostForm::INTERSECT; f.kids.assign(kids, kids + n);
    Boundary acc = forms[kids[0]].boundary;                            // Seq.reduce intersection (:66)
    for (int i = 1; i < n; ++i) acc = boundaryIntersection(acc, forms[kids[i]].boundary);
    f.boundary = acc;
    forms.push_back(f); return (int)forms.size() - 1;
}

int Builder::formUnionSmooth(float strength, const int* kids, int n) { // SdfForm.fs:69-91
    if (!checkKids(*this, kids, n, err)) return n <= 0 ? FT_ERR_EMPTY_ : FT_ERR_INVALID_;
    if (n == 1) return kids[0];
    HostForm f; f.kind = HostForm::SMOOTH; f.kids.assign(kids, kids + n); f.strength = strength;
    Boundary acc = forms[kids[0]].boundary;                            // :87-90
    for (int i = 1; i < n; ++i) acc = boundaryUnion(acc, forms[kids[i]].boundary);
    f.boundary = acc;
    forms.push_back(f); return (int)forms.size() - 1;
}

// ------------------------------------------------------------------------------------------------
// SdfMaterial.fs:4-7, SdfObject.fs:6-64, SdfLight.fs:6-42
// ------------------------------------------------------------------------------------------------
int Builder::materialSolid(f3 rgb) { materials.push_back(rgb); materialExt.push_back(MatExt{}); return (int)materials.size() - 1; }
int Builder::materialGlass(f3 tint, float ior, float dispersion) {     // EXTENSION (no counterpart in the reference)
    if (!(ior > 0.0f)) { err = "glass: ior must be positive"; return FT_ERR_INVALID_; }
    materials.push_back(tint);
    MatExt e; e.glass = 1; e.ior = ior; e.dispersion = dispersion;
    materialExt.push_back(e);
    return (int)materials.size() - 1;
}

int Builder::objectCreate(int material, int form) {                    // SdfObject.fs:6-10
    if (material < 0 || (size_t)material >= materials.size() || !okForm(form)) { err = "invalid handle"; return FT_ERR_INVALID_; }
    HostObject o; o.kind = HostObject::CREATE; o.material = material; o.form = form;
    objects.push_back(o); return (int)objects.size() - 1;
}

int Builder::objectUnion(const int* objs, int n) {                     // SdfObject.fs:12-48
    if (n <= 0 || !objs) { err = "No SdfObjects given."; return FT_ERR_EMPTY_; }
    for (int i = 0; i < n; ++i) if (!okObject(objs[i])) { err = "invalid object handle"; return FT_ERR_INVALID_; }
    if (n == 1) return objs[0];
    std::vector<int> fs; for (int i = 0; i < n; ++i) fs.push_back(objects[objs[i]].form);
    // Form = SdfForm.union of the children's forms (:16-19).  The material closure builds a second
    // lookup from the same boundaries (:26); it is identical, so one grid serves both.
    const int form = formUnion(fs.data(), n);
    if (form < 0) return form;
    HostObject o; o.kind = HostObject::UNION; o.form = form; o.kidObjects.assign(objs, objs + n);
    objects.push_back(o); return (int)objects.size() - 1;
}

int Builder::objectSubtract(int obj, int form) {                       // SdfObject.fs:50-54
    if (!okObject(obj) || !okForm(form)) { err = "invalid handle"; return FT_ERR_INVALID_; }
    const int f = formSubtract(objects[obj].form, form);
    if (f < 0) return f;
    HostObject o; o.kind = HostObject::SUBTRACT; o.form = f; o.inner = obj; o.forms = {form};
    objects.push_back(o); return (int)objects.size() - 1;
}

int Builder::objectIntersect(int obj, const int* fs, int n) {          // SdfObject.fs:56-64
    if (!okObject(obj)) { err = "invalid object handle"; return FT_ERR_INVALID_; }
    std::vector<int> all; all.push_back(objects[obj].form);
    for (int i = 0; i < n; ++i) { if (!okForm(fs[i])) { err = "invalid form handle"; return FT_ERR_INVALID_; } all.push_back(fs[i]); }
    const int f = formIntersect(all.data(), (int)all.size());
    if (f < 0) return f;
    HostObject o; o.kind = HostObject::INTERSECT; o.form = f; o.inner = obj; o.forms.assign(fs, fs + n);
    objects.push_back(o); return (int)objects.size() - 1;
}

int Builder::lightDirectional(f3 dir, f3 rgb) {                        // SdfLight.fs:6-21
    HostLight l{}; l.dev.type = FT_LIGHT_DIRECTIONAL;
    const f3 d = ft_normalize(neg3(dir));                              // :7
    l.dev.v[0] = d.x; l.dev.v[1] = d.y; l.dev.v[2] = d.z;
    l.dev.color[0] = rgb.x; l.dev.color[1] = rgb.y; l.dev.color[2] = rgb.z;
    lights.push_back(l); return (int)lights.size() - 1;
}

int Builder::lightPoint(f3 pos, f3 rgb) {                              // SdfLight.fs:23-42
    HostLight l{}; l.dev.type = FT_LIGHT_POINT;
    l.dev.v[0] = pos.x; l.dev.v[1] = pos.y; l.dev.v[2] = pos.z;
    l.dev.color[0] = rgb.x; l.dev.color[1] = rgb.y; l.dev.color[2] = rgb.z;
    lights.push_back(l); return (int)lights.size() - 1;
}

// ------------------------------------------------------------------------------------------------
// Flattening
// ------------------------------------------------------------------------------------------------
namespace {

struct Flattener {
    const Builder& b;
    FlatScene& out;
    std::string& err;
    uint32_t maxSlot = 0;
    bool failed = false;

    Flattener(const Builder& b_, FlatScene& o, std::string& e) : b(b_), out(o), err(e) {}

    bool fail(const std::string& m) { if (!failed) err = m; failed = true; return false; }
    void useSlot(uint32_t s) { if (s > maxSlot) maxSlot = s; }

    static uint32_t primType(HostForm::Kind k) {
        switch (k) {
            case HostForm::SPHERE: return FT_PR_SPHERE; case HostForm::CAPSULE: return FT_PR_CAPSULE;
            case HostForm::TORUS: return FT_PR_TORUS; case HostForm::TRIANGLE: return FT_PR_TRIANGLE;
            default: return FT_PR_BOX;
        }
    }
    uint32_t addConsts(const std::vector<float>& p) {
        const uint32_t at = (uint32_t)out.consts.size();
        out.consts.insert(out.consts.end(), p.begin(), p.end());
        return at;
    }
    uint32_t addBoundary(const Boundary& bd) {
        const uint32_t at = (uint32_t)out.consts.size();
        out.consts.push_back(bd.center.x); out.consts.push_back(bd.center.y); out.consts.push_back(bd.center.z); out.consts.push_back(bd.radius);
        return at;
    }
    FtInstr mk(uint32_t op, uint32_t dst) { FtInstr i{}; i.op = op; i.dst = dst; return i; }

    bool unionFastQ = true;     // all union candidates so far admit the kernel's clamped fast sqrt (see noteUnionChild)
    bool anyUnion = false;
    uint32_t stageEnd = 0;      // constant-pool prefix that must be mirrored in LDS for the fast runs
    double nearR = 1e30;        // min over fast runs of 86/|strengthInverse| - max |centre| (conservative, in double)
    std::vector<int> matRemap;  // context-wide material handle -> dense index in this scene's table

    uint32_t matIndex(int handle) {
        if (matRemap.empty()) matRemap.assign(b.materials.size(), -1);
        if (matRemap[handle] < 0) {
            matRemap[handle] = (int)(out.materials.size() / 3);
            const f3& m = b.materials[handle];
            out.materials.push_back(m.x); out.materials.push_back(m.y); out.materials.push_back(m.z);
            const Builder::MatExt& e = b.materialExt[handle];
            out.materialsExt.push_back(e.glass ? 1.0f : 0.0f); out.materialsExt.push_back(e.ior);
            out.materialsExt.push_back(e.dispersion); out.materialsExt.push_back(0.0f);
            out.nGlass += e.glass;
        }
        return (uint32_t)matRemap[handle];
    }

    // A run of spheres may take the kernel's guarded fast path (kernels.hip: smooth_run_spheres_fast)
    // when t = strengthInverse * (|c - p| - r) can never exceed the range its exp shortcut is proved
    // on: finite parameters, strength > 0 and |strengthInverse| * max r <= 80.  Everything else
    // (lower bound on t, tiny or non-finite |c - p|^2) is checked per evaluation in the kernel.
    bool fastSphereRun(const HostForm& f, size_t k, size_t run, float strengthInverse) {
        if (b.forms[f.kids[k]].kind != HostForm::SPHERE) return false;
        if (!(f.strength > 0.0f) || !std::isfinite(strengthInverse)) return false;
        const float a = fabsf(strengthInverse);
        // the kernel checks |p|inf < 20000 per evaluation; with |c|inf <= 1e4 that gives |c - p| < 65536, so
        // |t| <= a * (65536 + r) stays below 2.9e6 (a <= 32, r <= 1e4) and t <= a * r below 88 (a * r <= 80).
        // r >= 2^-20 makes the kernel's clamp of tiny |c - p|^2 result-neutral.
        if (!(a <= 32.0f) || !(a >= 0x1p-20f)) return false;
        double maxC = 0.0;
        for (size_t j = 0; j < run; ++j) {
            const std::vector<float>& p = b.forms[f.kids[k + j]].params;
            for (int c = 0; c < 4; ++c) if (!std::isfinite(p[c])) return false;
            for (int c = 0; c < 3; ++c) if (!(fabsf(p[c]) <= 1.0e4f)) return false;
            if (!(p[3] >= 0x1p-20f) || !(p[3] <= 1.0e4f) || !(p[3] * a <= 80.0f)) return false;
            maxC = std::max(maxC, std::sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]));
        }
        // inside this radius t = a * (r - |c - p|) >= -a * (|p| + |c|) >= -86: the exponential is a normal
        // number and the kernel may apply 2^n by an integer add to the exponent field
        nearR = std::min(nearR, 86.0 / (double)a - maxC - 1e-3 * (1.0 + maxC));
        return true;
    }

    // length of the run of consecutive primitives of one type starting at kids[i]
    size_t runLength(const std::vector<int>& kids, size_t i) const {
        const HostForm& f0 = b.forms[kids[i]];
        if (!f0.isPrim()) return 0;
        size_t j = i + 1;
        while (j < kids.size() && b.forms[kids[j]].kind == f0.kind) ++j;
        return j - i;
    }

    bool emitForm(int h, uint32_t dst) {
        if (failed) return false;
        if (dst >= FT_MAX_SLOTS) return fail("scene nests deeper than FT_MAX_SLOTS value slots");
        useSlot(dst);
        const HostForm& f = b.forms[h];
        switch (f.kind) {
        case HostForm::SPHERE: case HostForm::CAPSULE: case HostForm::TORUS: case HostForm::TRIANGLE: case HostForm::BOX: {
            FtInstr i = mk(FT_OP_PRIM, dst); i.type = primType(f.kind); i.data = addConsts(f.params); i.count = 1;
            out.instr.push_back(i);
            return true;
        }
        case HostForm::SUBTRACT: {                                     // Max(-(b), a): a first, then b
            if (!emitForm(f.kids[0], dst) || !emitForm(f.kids[1], dst + 1)) return false;
            FtInstr i = mk(FT_OP_SUBTRACT, dst); i.src = dst + 1;
            out.instr.push_back(i);
            return true;
        }
        case HostForm::INTERSECT: return emitIntersect(f.kids, dst, -1);
        case HostForm::SMOOTH: {
            const float strengthInverse = -1.0f / f.strength;          // SdfForm.fs:75
            bool first = true;
            for (size_t k = 0; k < f.kids.size();) {
                const size_t run = runLength(f.kids, k);
                if (run > 0) {
                    const HostForm& p0 = b.forms[f.kids[k]];
                    FtInstr i = mk(FT_OP_SMOOTH_RUN, dst); i.type = primType(p0.kind); i.count = (uint32_t)run;
                    i.data = (uint32_t)out.consts.size(); i.f0 = strengthInverse; i.flags = first ? FT_FLAG_INIT : 0u;
                    for (size_t j = 0; j < run; ++j) addConsts(b.forms[f.kids[k + j]].params);
                    if (fastSphereRun(f, k, run, strengthInverse)) {
                        i.flags |= FT_FLAG_FAST;
                        stageEnd = (uint32_t)out.consts.size();
                    }
                    out.instr.push_back(i);
                    k += run;
                } else {
                    if (!emitForm(f.kids[k], dst + 1)) return false;
                    FtInstr i = mk(FT_OP_SMOOTH_ADD, dst); i.src = dst + 1; i.f0 = strengthInverse; i.flags = first ? FT_FLAG_INIT : 0u;
                    out.instr.push_back(i);
                    k += 1;
                }
                first = false;
            }
            FtInstr i = mk(FT_OP_SMOOTH_FIN, dst); i.f0 = f.strength;
            out.instr.push_back(i);
            return true;
        }
        case HostForm::UNION: {
            std::vector<int> objs(f.kids.size(), -1);
            return emitUnion(f, objs, dst);
        }
        }
        return fail("unknown form kind");
    }

    // intersect: kids[0] evaluated unconditionally into dst (firstObject >= 0: it is an object whose
    // material is kept, SdfObject.fs:56-64), every further child behind the getMaxDistance test.
    bool emitIntersect(const std::vector<int>& kids, uint32_t dst, int firstObject) {
        if (firstObject >= 0) { if (!emitObject(firstObject, dst)) return false; }
        else if (!emitForm(kids[0], dst)) return false;
        // child 0 is a grid union evaluated straight into dst (its FT_OP_UNION is the instruction just emitted): see "lazy union" below
        const size_t unionAt = (!out.instr.empty() && out.instr.back().op == FT_OP_UNION && out.instr.back().dst == dst) ? out.instr.size() - 1 : (size_t)-1;
        for (size_t k = 1; k < kids.size();) {
            const size_t run = runLength(kids, k);
            if (run > 0) {
                const HostForm& p0 = b.forms[kids[k]];
                FtInstr i = mk(FT_OP_ISECT_RUN, dst); i.type = primType(p0.kind); i.count = (uint32_t)run;
                i.data = (uint32_t)out.consts.size();
                for (size_t j = 0; j < run; ++j) addConsts(b.forms[kids[k + j]].params);
                i.aux = (uint32_t)out.consts.size();
                for (size_t j = 0; j < run; ++j) addBoundary(b.forms[kids[k + j]].boundary);
                if (k == 1 && unionAt != (size_t)-1 && unionAt + 1 == out.instr.size()) {
                    // Lazy union: the intersect's value is Max(u, d1) once u < child 1's pruning bound (SdfForm.fs:60-63), and the union's fold starts at
                    // Items.[0] and only falls — wherever Items.[0] is already <= d1 the rest of the walk cannot change the result.  The kernel
                    // (FT_OP_UNION, FT_FLAG_LAZY) recomputes d1 and the bound from these offsets with the expressions FT_OP_ISECT_RUN uses.
                    FtInstr& u = out.instr[unionAt];
                    u.flags |= FT_FLAG_LAZY; u.type = i.type; u.data = i.data; u.count = i.aux;
                }
                out.instr.push_back(i);
                k += run;
            } else {
                if (!emitForm(kids[k], dst + 1)) return false;
                FtInstr i = mk(FT_OP_ISECT_APPLY, dst); i.src = dst + 1; i.aux = addBoundary(b.forms[kids[k]].boundary);
                out.instr.push_back(i);
                k += 1;
            }
        }
        return true;
    }

    // The union loop's fast square roots (kernels.hip ft_sq<true>) need: finite coordinates of magnitude
    // <= 1e4 (operands stay far below 2^100 once |p|inf < 20000) and every radius that is subtracted from
    // a root >= 2^-20 (so clamping a root at 2^-48 cannot change the difference): the candidate's bounding
    // radius (getMinDistance), sphere / capsule / triangle radius, torus major and minor radius.
    void noteUnionChild(const HostForm& kf) {
        auto okc = [](float v) { return std::isfinite(v) && fabsf(v) <= 1.0e4f; };
        auto okr = [](float v) { return std::isfinite(v) && v >= 0x1p-20f && v <= 1.0e4f; };
        bool ok = okc(kf.boundary.center.x) && okc(kf.boundary.center.y) && okc(kf.boundary.center.z) && okr(kf.boundary.radius);
        if (kf.isPrim()) {
            const std::vector<float>& p = kf.params;
            for (float v : p) ok = ok && okc(v);
            switch (kf.kind) {
                case HostForm::SPHERE: case HostForm::CAPSULE: case HostForm::TRIANGLE: ok = ok && okr(p[3]); break;
                case HostForm::TORUS: ok = ok && okr(p[3]) && okr(p[7]); break;
                default: break;                                         // box keeps the IEEE sqrt
            }
        }
        unionFastQ = unionFastQ && ok;
    }

    // rough instruction count of one evaluation of a union-free form (only used to decide slot vs call)
    double formCost(int h) const {
        const HostForm& f = b.forms[h];
        switch (f.kind) {
            case HostForm::SPHERE: return 12.0;
            case HostForm::CAPSULE: case HostForm::BOX: return 25.0;
            case HostForm::TORUS: return 40.0;
            case HostForm::TRIANGLE: return 100.0;
            default: break;
        }
        double c = f.kind == HostForm::SMOOTH ? 60.0 : 5.0;
        for (int k : f.kids) c += formCost(k) + (f.kind == HostForm::SMOOTH ? 15.0 : (f.kind == HostForm::INTERSECT ? 18.0 : 0.0));
        return c;
    }
    bool formHasUnion(int h) const {
        const HostForm& f = b.forms[h];
        if (f.kind == HostForm::UNION) return true;
        for (int k : f.kids) if (formHasUnion(k)) return true;
        return false;
    }
    bool objectHasUnion(int h) const {
        const HostObject& o = b.objects[h];
        if (o.kind == HostObject::UNION) return true;
        if (o.kind == HostObject::CREATE) return formHasUnion(o.form);
        if (objectHasUnion(o.inner)) return true;
        for (int f : o.forms) if (formHasUnion(f)) return true;
        return false;
    }

    std::vector<FtInstr> subPrograms;   // bodies of the FT_PR_CALL children; appended behind the main program by flatten()
    struct Call { uint32_t constAt, begin, end; };
    std::vector<Call> calls;            // consts[constAt .. +2] get (first instr, end instr, slot) once the main program's length is known

    // union over f.kids; objs[i] >= 0 when child i is an SdfObject (material tracking), else -1.
    // Primitive children go straight into the child table.  A combinator child without a union inside becomes a
    // sub-program that the candidate loop runs on demand, exactly when the reference would call its Distance
    // (FT_PR_CALL; all such children of one union share the slots from callBase up — a lane runs one at a time).
    // A child that contains a union is evaluated first, unconditionally, into its own slot (evaluation has no
    // side effects, so applying the reference's pruning tests to the stored value gives the reference's result).
    bool emitUnion(const HostForm& f, const std::vector<int>& objs, uint32_t dst) {
        const uint32_t childBase = (uint32_t)out.children.size();
        out.children.resize(childBase + f.kids.size());
        uint32_t nextSlot = dst + 1;
        uint32_t callBase = dst + 1;
        // which non-primitive children are evaluated up front (slot) and which on demand (call)?  A child that
        // contains a union must be a slot.  The others become calls when the union has more than 8 of them, or when
        // the small ones among them are expensive: the kernel variant that can call is ~15 % slower in the candidate
        // walk itself (measured on the 1000-torus and "mixed nested" scenes), which cheap on-demand children do not
        // win back, while dozens or hundreds — which could not be held in slots at all — or costly ones do.  In such a union a child whose bounding sphere covers
        // much of the union (wanted by most evaluations anyway) still goes up front, at most 6 of them.
        std::vector<char> asSlot(f.kids.size(), 0);
        size_t eligible = 0;
        double smallCost = 0.0;         // rough VALU cost of the eligible children that are small next to the union
        for (size_t k = 0; k < f.kids.size(); ++k) {
            const HostForm& kf = b.forms[f.kids[k]];
            const bool solidPrim = kf.isPrim() && (objs[k] < 0 || b.objects[objs[k]].kind == HostObject::CREATE);
            if (solidPrim || (objs[k] >= 0 ? objectHasUnion(objs[k]) : formHasUnion(f.kids[k]))) continue;
            ++eligible;
            if (!(kf.boundary.radius > 0.4f * f.boundary.radius)) smallCost += formCost(f.kids[k]);
        }
        // few eligible children: calls only if evaluating the small ones up front would cost more per evaluation than
        // the slower walk does (config 5: two glass blobs of 4 spheres, ~340 instructions -> calls, +16 %; the carved
        // sphere of the "mixed nested" scene, ~30 -> slot)
        const bool useCalls = eligible > 8 || smallCost >= 200.0;
        uint32_t bigSlots = 0;
        for (size_t k = 0; k < f.kids.size(); ++k) {
            const HostForm& kf = b.forms[f.kids[k]];
            const bool solidPrim = kf.isPrim() && (objs[k] < 0 || b.objects[objs[k]].kind == HostObject::CREATE);
            if (solidPrim) continue;
            if (!useCalls || (objs[k] >= 0 ? objectHasUnion(objs[k]) : formHasUnion(f.kids[k]))) asSlot[k] = 1;
            else if (bigSlots < 6 && kf.boundary.radius > 0.4f * f.boundary.radius) { asSlot[k] = 1; ++bigSlots; }
            if (asSlot[k]) ++callBase;
        }
        for (size_t k = 0; k < f.kids.size(); ++k) {
            const HostForm& kf = b.forms[f.kids[k]];
            noteUnionChild(kf); anyUnion = true;
            FtChild c{};
            c.bc[0] = kf.boundary.center.x; c.bc[1] = kf.boundary.center.y; c.bc[2] = kf.boundary.center.z; c.br = kf.boundary.radius;
            const int ko = objs[k];
            const bool solidPrim = kf.isPrim() && (ko < 0 || b.objects[ko].kind == HostObject::CREATE);
            if (solidPrim) {
                c.type = primType(kf.kind); c.data = addConsts(kf.params);
                c.mat = ko >= 0 ? matIndex(b.objects[ko].material) : 0u;
            } else if (asSlot[k]) {
                if (nextSlot >= FT_MAX_SLOTS) return fail("union needs more up-front value slots than FT_MAX_SLOTS");
                if (ko >= 0 ? !emitObject(ko, nextSlot) : !emitForm(f.kids[k], nextSlot)) return false;
                c.type = FT_PR_SLOT; c.data = nextSlot; c.mat = 0;
                ++nextSlot;
            } else {
                std::vector<FtInstr> outer;
                outer.swap(out.instr);                                  // the child's program goes to the sub-program area
                const bool ok = ko >= 0 ? emitObject(ko, callBase) : emitForm(f.kids[k], callBase);
                outer.swap(out.instr);                                  // out.instr = main program again, outer = the child's body
                if (!ok) return false;
                Call cl;
                cl.constAt = addConsts({0.0f, 0.0f, 0.0f, 0.0f});
                cl.begin = (uint32_t)subPrograms.size();
                subPrograms.insert(subPrograms.end(), outer.begin(), outer.end());
                cl.end = (uint32_t)subPrograms.size();
                uint32_t slotBits = callBase;
                memcpy(&out.consts[cl.constAt + 2], &slotBits, 4);
                calls.push_back(cl);
                c.type = FT_PR_CALL; c.data = cl.constAt; c.mat = 0;
            }
            out.children[childBase + k] = c;
        }
        const HostGrid& g = *f.grid;
        FtGrid dg{};
        dg.aabbMin[0] = g.aabbMin.x; dg.aabbMin[1] = g.aabbMin.y; dg.aabbMin[2] = g.aabbMin.z;
        dg.cellSizeInv[0] = g.cellSizeInv.x; dg.cellSizeInv[1] = g.cellSizeInv.y; dg.cellSizeInv[2] = g.cellSizeInv.z;
        dg.count[0] = g.count[0]; dg.count[1] = g.count[1]; dg.count[2] = g.count[2];
        dg.cellBase = (uint32_t)(out.cellCenters.size() / 3);
        dg.childBase = childBase; dg.nChildren = (uint32_t)f.kids.size();
        const uint32_t itemBase = (uint32_t)out.items.size();
        for (const f3& c : g.centers) { out.cellCenters.push_back(c.x); out.cellCenters.push_back(c.y); out.cellCenters.push_back(c.z); }
        // global CSR: cellStart has one entry per cell plus a final terminator kept at the back
        if (!out.cellStart.empty()) out.cellStart.pop_back();
        for (size_t ci = 0; ci + 1 < g.cellStart.size(); ++ci) out.cellStart.push_back(itemBase + g.cellStart[ci]);
        for (const FtItem& it : g.items) {                              // fuse item + child into the device record
            const FtChild& ch = out.children[childBase + it.child];
            if (ch.data >= (1u << 28)) return fail("constant pool exceeds 2^28 floats");
            FtItemRec r{};
            r.lowerBound = it.lowerBound;
            r.bc[0] = ch.bc[0]; r.bc[1] = ch.bc[1]; r.bc[2] = ch.bc[2]; r.br = ch.br;
            r.typeData = (ch.type & 15u) | (ch.data << 4);
            r.mat = ch.mat; r.child = it.child;
            out.items.push_back(r);
        }
        out.cellStart.push_back((uint32_t)out.items.size());
        const uint32_t gi = (uint32_t)out.grids.size();
        out.grids.push_back(dg);
        FtInstr i = mk(FT_OP_UNION, dst); i.aux = gi;
        out.instr.push_back(i);
        useSlot(nextSlot - 1);
        return true;
    }

    // main program | sub-programs: fix up the (first, end) instruction indices of every call record
    void linkCalls() {
        const uint32_t base = (uint32_t)out.instr.size();
        out.nMainInstr = base;
        for (const Call& cl : calls) {
            const uint32_t first = base + cl.begin, end = base + cl.end;
            memcpy(&out.consts[cl.constAt], &first, 4);
            memcpy(&out.consts[cl.constAt + 1], &end, 4);
        }
        out.instr.insert(out.instr.end(), subPrograms.begin(), subPrograms.end());
    }

    bool emitObject(int h, uint32_t dst) {
        if (failed) return false;
        if (dst >= FT_MAX_SLOTS) return fail("scene nests deeper than FT_MAX_SLOTS value slots");
        useSlot(dst);
        const HostObject& o = b.objects[h];
        switch (o.kind) {
        case HostObject::CREATE: {
            if (!emitForm(o.form, dst)) return false;
            FtInstr i = mk(FT_OP_SETLEAF, dst); i.aux = matIndex(o.material);
            out.instr.push_back(i);
            return true;
        }
        case HostObject::UNION: return emitUnion(b.forms[o.form], o.kidObjects, dst);
        case HostObject::SUBTRACT: {
            if (!emitObject(o.inner, dst) || !emitForm(o.forms[0], dst + 1)) return false;
            FtInstr i = mk(FT_OP_SUBTRACT, dst); i.src = dst + 1;
            out.instr.push_back(i);
            return true;
        }
        case HostObject::INTERSECT: {
            if (o.forms.empty()) return emitObject(o.inner, dst);       // SdfForm.intersect [x] = x
            std::vector<int> kids; kids.push_back(b.objects[o.inner].form);
            kids.insert(kids.end(), o.forms.begin(), o.forms.end());
            return emitIntersect(kids, dst, o.inner);
        }
        }
        return fail("unknown object kind");
    }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// Support sphere of a form: a sphere (centre, radius) such that  value(form, p) < tau  implies  dist(p, sphere) < tau  for every
// tau > 0 — so a ray that stays farther than epsilon from it can never "hit" (SdfForm.fs:98) and its march is known to end in a miss.
// (kernels.hip ft_never_enters).  By structure, in double:
//   primitive (exact signed distance)   a sphere around the shape, from its own parameters
//   union      min of (a subset of) the children   ->  a sphere around the children's spheres
//   subtract   Max(-b, a) >= a                     ->  a's sphere          (SdfForm.fs:46-47)
//   intersect  the running Max starts at child 0   ->  child 0's sphere    (SdfForm.fs:60-63: later children may be skipped, never child 0)
//              ... and ends >= every SPHERE child k: at step k either max < |c_k - p| + r_k and max becomes Max(max, d_k) >= d_k, or
//              max >= |c_k - p| + r_k >= |c_k - p| - r_k = d_k (same float distance on both sides, r_k >= 0; a sphere's boundary is the
//              sphere itself, SdfForm.fs:131-135); later steps only raise max.  So that child's own sphere serves too: the smallest wins
//              (Program.fs:67-77: the 3.5 sphere instead of the 1000 tori's 4.7 — a quarter of the frame's evaluations belong to rays
//              that pass the tori's sphere and miss the 3.5 one)
//   unionSmooth  -k ln(sum exp(-d_i / k)) >= min d_i - k ln n   ->  the children's spheres grown by k ln n (needs k > 0)
// false: no finite sphere is known (degenerate parameters, k <= 0): the scene gets none and every ray marches to its end.
// ------------------------------------------------------------------------------------------------
struct Support { double c[3]; double r; };
static bool finite3(const double v[3]) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }
static bool enclose(const std::vector<Support>& ss, Support& out) {
    if (ss.empty()) return false;
    double c[3] = {0, 0, 0};
    for (const Support& s : ss) for (int k = 0; k < 3; ++k) c[k] += s.c[k] / (double)ss.size();
    double r = 0.0;
    for (const Support& s : ss) {
        const double d = std::sqrt((s.c[0] - c[0]) * (s.c[0] - c[0]) + (s.c[1] - c[1]) * (s.c[1] - c[1]) + (s.c[2] - c[2]) * (s.c[2] - c[2])) + s.r;
        if (d > r) r = d;
    }
    out = Support{{c[0], c[1], c[2]}, r};
    return finite3(out.c) && std::isfinite(out.r);
}
static bool supportOf(const Builder& b, int h, Support& out, int depth = 0) {
    if (depth > 64 || !b.okForm(h)) return false;
    const HostForm& f = b.forms[h];
    const std::vector<float>& P = f.params;
    auto V = [&](size_t i) { return Support{{(double)P[i], (double)P[i + 1], (double)P[i + 2]}, 0.0}; };
    auto dist = [](const Support& a, const Support& c) { return std::sqrt((a.c[0] - c.c[0]) * (a.c[0] - c.c[0]) + (a.c[1] - c.c[1]) * (a.c[1] - c.c[1]) + (a.c[2] - c.c[2]) * (a.c[2] - c.c[2])); };
    bool ok = false;
    // a primitive whose derived parameters are not finite (a capsule of length 0: dirInv = 0 / 0; a collinear triangle; a torus with a zero
    // normal) evaluates to NaN everywhere; the reference's march never ends on it and both sides flag that — no shortcut for such a scene
    if (f.isPrim()) for (float v : P) if (!std::isfinite(v)) return false;
    switch (f.kind) {
    case HostForm::SPHERE: out = V(0); out.r = std::fabs((double)P[3]); ok = true; break;                       // params: c, r
    case HostForm::CAPSULE: {                                                                                    // from, r, dir, dirInv
        out = V(0); for (int k = 0; k < 3; ++k) out.c[k] += 0.5 * (double)P[4 + k];
        out.r = 0.5 * std::sqrt((double)P[4] * P[4] + (double)P[5] * P[5] + (double)P[6] * P[6]) + std::fabs((double)P[3]); ok = true; break;
    }
    case HostForm::TORUS: out = V(0); out.r = std::fabs((double)P[3]) + std::fabs((double)P[7]); ok = true; break;   // c, R, n, r
    case HostForm::TRIANGLE: {                                                                                   // v1, r | v2, - | v3, - | ...
        const Support v1 = V(0), v2 = V(4), v3 = V(8);
        Support c{{(v1.c[0] + v2.c[0] + v3.c[0]) / 3.0, (v1.c[1] + v2.c[1] + v3.c[1]) / 3.0, (v1.c[2] + v2.c[2] + v3.c[2]) / 3.0}, 0.0};
        c.r = std::max(dist(c, v1), std::max(dist(c, v2), dist(c, v3))) + std::fabs((double)P[3]);
        out = c; ok = true; break;
    }
    case HostForm::BOX: out = V(0); out.r = std::sqrt((double)P[4] * P[4] + (double)P[5] * P[5] + (double)P[6] * P[6]); ok = true; break;   // c, -, half
    case HostForm::SUBTRACT: return !f.kids.empty() && supportOf(b, f.kids[0], out, depth + 1);
    case HostForm::INTERSECT: {
        if (f.kids.empty()) return false;
        bool have = supportOf(b, f.kids[0], out, depth + 1);
        for (size_t k = 1; k < f.kids.size(); ++k) {
            if (!b.okForm(f.kids[k])) return false;
            const HostForm& kf = b.forms[f.kids[k]];
            if (kf.kind != HostForm::SPHERE) continue;
            const std::vector<float>& Q = kf.params;
            const bool fin = std::isfinite(Q[0]) && std::isfinite(Q[1]) && std::isfinite(Q[2]) && std::isfinite(Q[3]) && Q[3] >= 0.0f;
            if (fin && (!have || (double)Q[3] < out.r)) { out = Support{{(double)Q[0], (double)Q[1], (double)Q[2]}, (double)Q[3]}; have = true; }
        }
        return have && finite3(out.c) && std::isfinite(out.r);
    }
    case HostForm::UNION: case HostForm::SMOOTH: {
        double grow = 0.0;
        if (f.kind == HostForm::SMOOTH) {
            if (!(f.strength > 0.0f) || !std::isfinite(f.strength)) return false;
            grow = (double)f.strength * std::log((double)f.kids.size());
        }
        std::vector<Support> ss;
        for (int k : f.kids) { Support s; if (!supportOf(b, k, s, depth + 1)) return false; s.r += grow; ss.push_back(s); }
        return enclose(ss, out);
    }
    }
    return ok && finite3(out.c) && std::isfinite(out.r);
}

// Is the program ONE grid union of plain primitives followed by at most FT_CARVE_TAIL single-primitive intersect / subtract steps
// (ft_device.h "Carved union": the shape of Program.fs:67-77)?  Then record the tail and lay the candidate lists out with terminators.
static bool carvedShape(FlatScene& out) {
    if (out.nMainInstr != out.instr.size() || out.instr.empty() || out.grids.size() != 1) return false;
    const FtInstr& u = out.instr[0];
    if (u.op != FT_OP_UNION || u.dst != 0 || u.aux != 0) return false;
    uint32_t kind = FT_PR_SLOT;
    for (const FtChild& c : out.children) {
        if (c.type > FT_PR_BOX) return false;                          // a slot or an on-demand sub-program: the general kernels
        kind = kind == FT_PR_SLOT ? c.type : (kind == c.type ? kind : FT_CARVE_MIXED);
    }
    if (kind == FT_PR_SLOT) return false;
    FtCarve cv{};
    cv.kind = kind;
    auto fastSphere = [](const float* q) {                              // the bounds of noteUnionChild for a sphere (c.xyz, r)
        for (int k = 0; k < 3; ++k) if (!(std::isfinite(q[k]) && fabsf(q[k]) <= 1.0e4f)) return false;
        return std::isfinite(q[3]) && q[3] >= 0x1p-20f && q[3] <= 1.0e4f;
    };
    auto strideOf = [](uint32_t t) { return t == FT_PR_SPHERE ? FT_STRIDE_SPHERE : t == FT_PR_CAPSULE ? FT_STRIDE_CAPSULE : t == FT_PR_TORUS ? FT_STRIDE_TORUS
                                          : t == FT_PR_TRIANGLE ? FT_STRIDE_TRIANGLE : FT_STRIDE_BOX; };
    for (size_t k = 1; k < out.instr.size(); ++k) {
        const FtInstr& in = out.instr[k];
        if (in.op == FT_OP_ISECT_RUN && in.dst == 0) {
            for (uint32_t j = 0; j < in.count; ++j) {
                if (cv.nTail == FT_CARVE_TAIL) return false;
                FtCarveOp op{FT_OP_ISECT_RUN, in.type, in.data + j * (uint32_t)strideOf(in.type), in.aux + 4u * j};
                if (in.type == FT_PR_SPHERE && fastSphere(&out.consts[op.data]) && memcmp(&out.consts[op.data], &out.consts[op.bound], 16) == 0) op.type |= FT_CARVE_FAST_SPHERE;
                cv.tail[cv.nTail++] = op;
            }
        } else if (in.op == FT_OP_PRIM && in.dst == 1 && k + 1 < out.instr.size() && out.instr[k + 1].op == FT_OP_SUBTRACT &&
                   out.instr[k + 1].dst == 0 && out.instr[k + 1].src == 1) {
            if (cv.nTail == FT_CARVE_TAIL) return false;
            FtCarveOp op{FT_OP_SUBTRACT, in.type, in.data, 0u};
            if (in.type == FT_PR_SPHERE && fastSphere(&out.consts[op.data])) op.type |= FT_CARVE_FAST_SPHERE;
            cv.tail[cv.nTail++] = op;
            ++k;
        } else return false;
    }
    const size_t ncells = out.cellStart.size() - 1;
    if ((out.items.size() + ncells + 4) * sizeof(FtItemRec) >= ((size_t)1 << 31) || out.consts.size() >= ((size_t)1 << 26)) return false;   // byte offsets stay 32-bit
    FtItemRec stop{};
    stop.lowerBound = INFINITY;
    out.itemsT.clear(); out.cellStartT.clear();
    out.itemsT.reserve(out.items.size() + ncells + 1);
    for (size_t ci = 0; ci < ncells; ++ci) {
        out.cellStartT.push_back((uint32_t)(out.itemsT.size() * sizeof(FtItemRec)));
        for (uint32_t i = out.cellStart[ci]; i < out.cellStart[ci + 1]; ++i) {
            FtItemRec r = out.items[i];
            r.typeData = (r.typeData & 15u) | ((r.typeData >> 4) << 6);        // bits 4..: BYTE offset of the constants (the general walk keeps a float index there)
            out.itemsT.push_back(r);
        }
        out.itemsT.push_back(stop);
    }
    for (int k = 0; k < 4; ++k) out.itemsT.push_back(stop);           // the walk requests two records at a time, and may do so one trip ahead
    out.carve = cv;
    return true;
}

// Every constant of every primitive under `h` is finite.  A capsule of length 0 (dirInv = 0 / 0), a collinear triangle or a torus with a zero normal
// evaluates to NaN everywhere, and MathF.Max / Min carry that NaN to the scene's value wherever the form sits in the tree — also where supportOf
// never looks (the subtrahend of a subtract, the later children of an intersect).  The reference's march never ends on a NaN and both sides flag it:
// a ray ended early would not, so such a scene gets no support sphere.
static bool finiteTree(const Builder& b, int h, int depth = 0) {
    if (depth > 64 || !b.okForm(h)) return false;
    const HostForm& f = b.forms[h];
    if (f.isPrim()) { for (float v : f.params) if (!std::isfinite(v)) return false; return true; }
    if (f.kind == HostForm::SMOOTH && !std::isfinite(f.strength)) return false;
    for (int k : f.kids) if (!finiteTree(b, k, depth + 1)) return false;
    return true;
}

bool flatten(const Builder& b, int object, const float bg[3], const int* lights, int nLights, FlatScene& out, std::string& err) {
    if (!b.okObject(object)) { err = "invalid object handle"; return false; }
    out = FlatScene{};
    Flattener fl(b, out, err);
    if (!fl.emitObject(object, 0)) return false;
    fl.linkCalls();
    out.nSlots = fl.maxSlot + 1;
    out.nStage = fl.stageEnd <= FT_MAX_STAGE_FLOATS ? fl.stageEnd : FT_MAX_STAGE_FLOATS;
    out.nearR2 = (fl.stageEnd > 0 && fl.nearR > 0.0 && fl.nearR < 1e29) ? (float)(fl.nearR * fl.nearR * (1.0 - 1e-5)) : 0.0f;
    out.fastQ = (fl.anyUnion && fl.unionFastQ) ? 1u : 0u;
    // The sphere no hit can lie outside of (see supportOf), padded twice:
    //   padEval  — what a float32 evaluation may differ from the exact value: 0.1 % of the radius, 0.01, and 0.1 % of the centre's coordinates;
    //   padDrift — "drift of the marched points".  The reference moves the ray's origin step by step in float32 (Ray.fs:9-13: Origin + Direction * d), so the
    //     points it evaluates leave the line the shortcut reasons about (kernels.hip ft_never_enters).  One step that starts or ends at most rho from the centre c
    //     adds at most e(rho) = 3 * 2^-23 * (|c|inf + rho) to that drift (per component: the rounding of a product <= 2 rho and of a sum <= |c|inf + rho).
    //     Let Rp = r + padEval + padDrift (= escR), a = |Direction| >= 1/2, 0 <= epsilon <= Rp, and suppose the drift so far is <= padDrift / 2.  The line stays
    //     >= Rp + epsilon from c, so every evaluated point is >= r + padEval + epsilon + padDrift / 2 from c: no hit there, and each step is d >= g = epsilon + padDrift / 2
    //     (support property), i.e. moves a d >= padDrift / 4 along the line.  Three kinds of steps:
    //       near      (start and end within 5 Rp of c — that is >= 2.5 (Rp + epsilon)): the line's chord through that ball is <= 10 Rp long: at most 40 Rp / padDrift + 1
    //                 steps of error <= e(5 Rp) each.  padDrift is the root of  (40 Rp / padDrift + 1) e(5 Rp) = padDrift / 4.
    //       approach  (start farther than 5 Rp, before the near steps): d >= rho - Rp - epsilon >= rho / 2 moves >= rho / 4 along a line of which at most 2 rho are left
    //                 to the near ball, so that length shrinks by 7/8 per step: N(rho0) = 7.5 ln(8 rho0 / padDrift) + 1 steps from a start at rho0, each
    //                 <= e(rho0).  The kernel takes the shortcut only from starts with N(rho0) e(rho0) <= padDrift / 4 (escRho2, by bisection below).
    //       receding  (rho >= 5 Rp, moving away): rho grows by >= sqrt(1 + a^2 / 4) >= 1.03 per step, so the errors of the steps so far form a geometric series
    //                 <= 34 e-terms of the last one: < 1.3e-5 rho + 1.2e-5 |c|inf ln(rho / 5 Rp) <= rho / 50 (|c|inf <= 1000 Rp: padEval), against a margin
    //                 rho - (r + padEval + epsilon) >= rho - 2 Rp >= 0.6 rho.
    //     Near and approach together stay <= padDrift / 2: the supposition holds step after step, every evaluated point keeps its distance, no step can be a hit.
    //     (Short rays — kernels.hip: Length <= 10 escR, the whole march within 2.5 escR of c — make at most Length / g + 1 <= 20 Rp / padDrift + 1 steps of error
    //     <= e(2.5 Rp): inside the near budget whatever |Direction| is.)
    Support sup;
    if (finiteTree(b, b.objects[object].form) && supportOf(b, b.objects[object].form, sup) && sup.r < 1e15 && std::fabs(sup.c[0]) + std::fabs(sup.c[1]) + std::fabs(sup.c[2]) < 1e15) {
        out.escC[0] = (float)sup.c[0]; out.escC[1] = (float)sup.c[1]; out.escC[2] = (float)sup.c[2];
        const double cInf = std::max(std::fabs(sup.c[0]), std::max(std::fabs(sup.c[1]), std::fabs(sup.c[2])));
        const double padEval = sup.r * 0.001 + 0.01 + 1e-3 * (std::fabs(sup.c[0]) + std::fabs(sup.c[1]) + std::fabs(sup.c[2]));
        const double u3 = 3.0 * 0x1p-23;
        double padDrift = 0.0;
        for (int it = 0; it < 8; ++it) {                               // padDrift^2 - k padDrift - 40 k Rp = 0 with k = 4 e(5 Rp), Rp depending on padDrift
            const double Rp = sup.r + padEval + padDrift;
            const double k = 4.0 * u3 * (cInf + 5.0 * Rp);
            padDrift = 1.01 * (0.5 * (k + std::sqrt(k * k + 160.0 * k * Rp)));
        }
        out.escR = (float)((sup.r + padEval + padDrift) * (1.0 + 1e-6));
        auto approach = [&](double rho0) { return (7.5 * std::log(8.0 * rho0 / padDrift) + 1.0) * u3 * (cInf + rho0); };
        double lo = 0.0, hi = 1e15;                                    // largest start distance whose approach drift stays <= padDrift / 4
        if (approach(padDrift) > 0.25 * padDrift) hi = 0.0;
        else { lo = padDrift; for (int it = 0; it < 200; ++it) { const double mid = 0.5 * (lo + hi); (approach(mid) <= 0.25 * padDrift ? lo : hi) = mid; } }
        out.escRho2 = (float)std::min(lo * lo * (1.0 - 1e-6), 1e30);
    }
    // kernel variant: 1 = the program is only staged fast sphere runs + SMOOTH_FIN + SETLEAF
    bool lean = !out.instr.empty() && out.nMainInstr == out.instr.size();
    for (const FtInstr& in : out.instr) {
        if (in.op == FT_OP_SMOOTH_RUN) lean = lean && (in.flags & FT_FLAG_FAST) && in.dst == 0 && in.data + 4u * in.count <= out.nStage;
        else if (in.op == FT_OP_SMOOTH_FIN || in.op == FT_OP_SETLEAF) lean = lean && in.dst == 0;
        else lean = false;
    }
    out.fastPath = lean ? 1u : (fl.calls.empty() ? 0u : 2u);     // kernel variant (ft_launch_trace)
    // the child-culling pass serves ONE sphere run of the main program: the longest staged fast one with at least 32 children and a positive strength
    for (uint32_t pc = 0, best = 0; pc < out.nMainInstr; ++pc) {
        const FtInstr& in = out.instr[pc];
        if (in.op == FT_OP_SMOOTH_RUN && (in.flags & FT_FLAG_FAST) && in.count >= 32u && in.count > best && in.f0 < 0.0f && in.data + 4u * in.count <= out.nStage) { best = in.count; out.cullPc = pc; }
    }
    if (out.fastPath == 0u && carvedShape(out)) out.fastPath = 3u;
    for (int i = 0; i < nLights; ++i) {
        if (lights[i] < 0 || (size_t)lights[i] >= b.lights.size()) { err = "invalid light handle"; return false; }
        out.lights.push_back(b.lights[lights[i]].dev);
    }
    if (out.materials.empty()) { out.materials.assign(3, 0.0f); out.materialsExt.assign(4, 0.0f); }           // a form-only union under no create(): index 0 must exist
    out.bg[0] = bg[0]; out.bg[1] = bg[1]; out.bg[2] = bg[2];
    if (out.cellStart.empty()) out.cellStart.push_back(0);
    // keep every pool non-empty and padded so device-side wide loads never run off the end
    for (int i = 0; i < 64; ++i) out.consts.push_back(0.0f);
    return true;
}

void spectralTable(int nw, float out[][4]) {
    // bin j looks at lambda_j = 400 + (j + 1/2) * 300 / nw nanometres.  Colour response: three tents
    // (red 610 +- 120, green 540 +- 110, blue 460 +- 120 nm), each channel scaled so that its nw weights
    // average exactly 1 in real arithmetic: without dispersion a spectral render reproduces the plain one.
    static const double mid[3] = {610.0, 540.0, 460.0}, half[3] = {120.0, 110.0, 120.0};
    double tent[16][3], total[3] = {0.0, 0.0, 0.0};
    for (int j = 0; j < nw; ++j) {
        const double nm = 400.0 + ((double)j + 0.5) * 300.0 / (double)nw;
        for (int ch = 0; ch < 3; ++ch) {
            const double off = nm < mid[ch] ? mid[ch] - nm : nm - mid[ch];
            const double v = 1.0 - off / half[ch];
            tent[j][ch] = v > 0.0 ? v : 0.0;
            total[ch] += tent[j][ch];
        }
        const double um = nm / 1000.0;
        out[j][3] = (float)(1.0 / (um * um) - 1.0 / (0.55 * 0.55));          // Cauchy term relative to 550 nm
    }
    for (int j = 0; j < nw; ++j)
        for (int ch = 0; ch < 3; ++ch) out[j][ch] = (float)(tent[j][ch] * (double)nw / total[ch]);
}

float lensCreate(float fov) {                                          // Camera.fs:11-14
    // F# `sin` on a float32 is evaluated in double precision and rounded (DESIGN.md assumptions)
    return (float)sin((double)(fov * 0.5f));
}

void cameraLookAt(f3 pos, f3 lookAt, f3 up, float nearPlaneSize, f3 o[4]) {   // Camera.fs:33-42
    const f3 forward = ft_normalize(lookAt - pos);
    const f3 right = ft_normalize(ft_cross(up, forward));
    o[0] = pos; o[1] = forward; o[2] = ft_cross(forward, right) * nearPlaneSize; o[3] = right * nearPlaneSize;
}

}  // namespace ft
