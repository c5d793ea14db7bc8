// console.cpp — the reference's console program (src/FrayTracer.Console/Program.fs:14-100) as a C++ host
// over libfraytracer_hip: System.Random(19) scene of 1000 tori cut by two spheres, two lights, 1000x1000,
// epsilon 0.01, ray length 30, timing line, result.bmp.  `--raw file` additionally dumps the float image
// (tests compare it with the Python host), `--device -1` builds the scene without a GPU and stops, `--math glibc` selects FT_OPT_MATH.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "FrayTracer.hpp"

using namespace FrayTracer;

// System.Random(seed), .NET's seeded Knuth subtractive generator — port from memory, see
// fraytracer_amd/dotnet_random.py for the caveat.
struct DotNetRandom {
    int sa[56]; int inext = 0, inextp = 21;
    explicit DotNetRandom(int seed) {
        const int MBIG = 2147483647, MSEED = 161803398;
        int mj = MSEED - std::abs(seed); sa[55] = mj; int mk = 1, ii = 0; sa[0] = 0;
        for (int i = 1; i < 55; ++i) { if ((ii += 21) >= 55) ii -= 55; sa[ii] = mk; mk = mj - mk; if (mk < 0) mk += MBIG; mj = sa[ii]; }
        for (int k = 1; k < 5; ++k) for (int i = 1; i < 56; ++i) { int n = i + 30; if (n >= 55) n -= 55; sa[i] -= sa[1 + n]; if (sa[i] < 0) sa[i] += MBIG; }
    }
    int Next() {
        const int MBIG = 2147483647;
        if (++inext >= 56) inext = 1;
        if (++inextp >= 56) inextp = 1;
        int r = sa[inext] - sa[inextp];
        if (r == MBIG) r--;
        if (r < 0) r += MBIG;
        sa[inext] = r; return r;
    }
    float range_01() { return (float)(Next() * (1.0 / 2147483647)); }                      // Random.fs:9
    float range(float lo, float hi) { return lo + range_01() * (hi - lo); }                // Random.fs:11
    Vector3 vec() { float x = range(-1, 1), y = range(-1, 1), z = range(-1, 1); return Vector3{x, y, z}; }
    Vector3 pointInBall(float radius) {                                                    // Random.fs:27-32
        for (;;) { Vector3 v = vec(); if ((v.x * v.x + v.y * v.y) + v.z * v.z <= 1.0f) return Vector3{v.x * radius, v.y * radius, v.z * radius}; }
    }
    Vector3 pointOnSphere(float radius) {                                                  // Random.fs:34-40
        for (;;) {
            Vector3 v = vec(); float len = (v.x * v.x + v.y * v.y) + v.z * v.z;
            if (0.01f <= len && len <= 1.0f) { float s = sqrtf(len); return Vector3{v.x / s * radius, v.y / s * radius, v.z / s * radius}; }
        }
    }
};

static void saveBitmap(const char* path, ft_ctx* ctx, const std::vector<float>& img, int X, int Y) {
    // Image.toColors 2.2f rng (Image.fs:37-50) and Image.toBitmap's buffer order (Image.fs:61-86) on the GPU (ft_tone_map_host); the
    // reference's dithering noise is racy, here it is the library's counter-based hash seeded with 19.  BMP rows are stored bottom-up.
    const std::vector<unsigned char> rows = Image::toColors(ctx, 2.2f, 19, img, ImageSize{X, Y}, true);
    const int stride = (3 * X + 3) & ~3;
    std::vector<unsigned char> body((size_t)stride * Y, 0);
    for (int r = 0; r < Y; ++r) memcpy(&body[(size_t)(Y - 1 - r) * stride], &rows[(size_t)r * X * 3], (size_t)X * 3);
    unsigned char h[54] = {'B', 'M'};
    auto put = [&](int at, unsigned v) { memcpy(h + at, &v, 4); };
    put(2, 54 + (unsigned)body.size()); put(10, 54); put(14, 40); put(18, X); put(22, Y); h[26] = 1; h[28] = 24; put(34, (unsigned)body.size());
    FILE* f = fopen(path, "wb"); if (!f) return; fwrite(h, 1, 54, f); fwrite(body.data(), 1, body.size(), f); fclose(f);
}

int main(int argc, char** argv) {
    int size = 1000, tori = 1000, device = 0; const char* raw = nullptr; const char* out = "result.bmp"; bool glibc = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--size") && i + 1 < argc) size = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--tori") && i + 1 < argc) tori = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--raw") && i + 1 < argc) raw = argv[++i];
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "--math") && i + 1 < argc) glibc = !strcmp(argv[++i], "glibc");
    }
    try {
        Context ctx(device);
        // --math glibc: MathF.Pow of the tone map (and MathF.Exp / Log of a unionSmooth) as this host's C runtime computes them (FT_OPT_MATH);
        // glibc picks its FMA build of expf / logf / powf iff the CPU has FMA and AVX2
        if (glibc) ctx.setOption(FT_OPT_MATH, (__builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2")) ? FT_MATH_GLIBC_FMA : FT_MATH_GLIBC_SSE2);
        DotNetRandom rng(19);                                                              // Program.fs:14
        auto camera = Camera::lookAt({Vector3{0, 0, -10}, Vector3{0, 0, 0}, Vector3{0, 1, 0}, Lens::create(60.0f)});   // :16-22
        auto randomMaterial = [&]() { float r = rng.range_01(), g = rng.range_01(), b = rng.range_01(); return SdfMaterial::createSolid(ctx, FColor::ofRGB(r, g, b)); };
        auto randomTorus = [&]() {                                                         // Program.fs:48-55
            ft_torus t; t.center = rng.pointInBall(4.0f); t.normal = rng.pointOnSphere(1.0f);
            t.major_radius = rng.range(0.1f, 0.4f); t.minor_radius = rng.range(0.1f, 0.3f);
            auto form = SdfForm::Primitive::torus(ctx, t);
            return SdfObject::create(randomMaterial(), form);
        };
        std::vector<SdfObjectV> objs; for (int i = 0; i < tori; ++i) objs.push_back(randomTorus());
        SdfScene scene{                                                                    // Program.fs:67-83
            SdfObject::subtract(SdfObject::intersect(SdfObject::unionOf(objs), {SdfForm::Primitive::sphere(ctx, ft_sphere{Vector3{0, 0, 0}, 3.5f})}),
                                SdfForm::Primitive::sphere(ctx, ft_sphere{Vector3{-0.5f, 1.0f, -2.0f}, 2.5f})),
            FColor::ofRGB(0.1f, 0.1f, 0.1f),
            {SdfLight::directional(ctx, Vector3{-0.5f, -1.0f, 1.0f}, FColor::ofRGB(0.5f, 0.5f, 0.5f)),
             SdfLight::point(ctx, Vector3{-0.5f, 0.0f, -2.0f}, FColor::ofRGB(10.0f, 0.0f, 0.0f))}};
        if (device < 0) { printf("scene built on a host-only context (%d tori); rendering needs a GPU\n", tori); return 0; }
        const float epsilon = 0.01f;                                                       // Program.fs:85
        printf("Rendering...\n");
        auto t0 = std::chrono::steady_clock::now();
        ft_stats st{};
        auto traced = Image::renderScene(epsilon, 30.0f, ImageSize{size, size}, camera, scene, &st);   // Program.fs:90-93
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("Time = %.2f sec\n", sec);                                                  // Program.fs:96
        printf("rays %llu primary + %llu shadow, kernel %.3f ms, %.1f Mrays/s (kernel)\n", (unsigned long long)st.rays_primary,
               (unsigned long long)st.rays_shadow, st.kernel_ms, (st.rays_primary + st.rays_shadow) / (st.kernel_ms * 1e3));
        if (raw) { FILE* f = fopen(raw, "wb"); if (f) { fwrite(traced.data(), 4, traced.size(), f); fclose(f); } }
        saveBitmap(out, ctx.get(), traced, size, size);                                               // Program.fs:98-100
    } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
