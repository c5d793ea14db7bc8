// FrayTracer.hpp — C++ host mirror of the reference's F# scene-composition API over the C ABI
// (include/fraytracer_hip.h).  The reference is compiled F#; no .NET toolchain exists in the build image,
// so this header is the compiled-language host layer: same module / function names and argument order
// as the F# (C++ keywords `union` -> `unionOf`), immutable value handles, errors as exceptions carrying
// ft_last_error().  INTEGRATION.md shows the F# [<DllImport>] form of the same calls.
//
//   SdfForm::Primitive::sphere/capsule/torus/triangle   src/FrayTracer/SdfForm.fs:117-268
//   SdfForm::unionOf/subtract/intersect/unionSmooth     src/FrayTracer/SdfForm.fs:14-91
//   SdfMaterial::createSolid                            src/FrayTracer/SdfMaterial.fs:4-7
//   SdfObject::create/unionOf/subtract/intersect        src/FrayTracer/SdfObject.fs:6-64
//   SdfLight::directional/point                         src/FrayTracer/SdfLight.fs:6-42
//   Lens::create, Camera::lookAt                        src/FrayTracer/Camera.fs:11-42
//   Image::renderScene                                  src/FrayTracer/Image.fs:26-35 + SdfScene.fs:7-28
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fraytracer_hip.h"

namespace FrayTracer {

struct Error : std::runtime_error {
    int code;
    Error(int c) : std::runtime_error(std::string("libfraytracer_hip: ") + ft_last_error()), code(c) {}
};
inline int check(int rc) { if (rc < 0) throw Error(rc); return rc; }

using Vector3 = ft_vec3;
struct FColor { Vector3 v; static FColor ofRGB(float r, float g, float b) { return FColor{Vector3{r, g, b}}; } };

class Context {                                   // one per GPU (device -1: construction only)
public:
    explicit Context(int device = 0) {
        if (ft_abi_version() != FT_ABI_VERSION) throw std::runtime_error("libfraytracer_hip: ABI version mismatch (header vs library)");
        check(ft_ctx_create(device, &ctx_));
    }
    ~Context() { ft_ctx_destroy(ctx_); }
    // per-context switches (ft_option); e.g. setOption(FT_OPT_MATH, FT_MATH_GLIBC_FMA): MathF.Exp / Log / Pow as this host's glibc computes them
    void setOption(ft_option option, int value) { check(ft_ctx_set_option(ctx_, (int32_t)option, value)); }
    int getOption(ft_option option) const { int32_t v = 0; check(ft_ctx_get_option(ctx_, (int32_t)option, &v)); return v; }
    static std::string buildInfo() { return ft_build_info(); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ft_ctx* get() const { return ctx_; }
private:
    ft_ctx* ctx_ = nullptr;
};

struct SdfFormV { ft_handle Node; ft_ctx* ctx; };
struct SdfMaterialV { ft_handle Node; ft_ctx* ctx; };
struct SdfObjectV { ft_handle Node; ft_ctx* ctx; };
struct SdfLightV { ft_handle Node; ft_ctx* ctx; };
struct SdfScene { SdfObjectV Object; FColor BackgroundColor; std::vector<SdfLightV> Lights; };   // Types.fs:74-79

inline std::vector<ft_handle> nodes(const std::vector<SdfFormV>& v) { std::vector<ft_handle> h; for (auto& f : v) h.push_back(f.Node); return h; }

namespace SdfForm {
namespace Primitive {
inline SdfFormV sphere(const Context& c, const ft_sphere& d) { return {check(ft_form_sphere(c.get(), &d)), c.get()}; }
inline SdfFormV capsule(const Context& c, const ft_capsule& d) { return {check(ft_form_capsule(c.get(), &d)), c.get()}; }
inline SdfFormV torus(const Context& c, const ft_torus& d) { return {check(ft_form_torus(c.get(), &d)), c.get()}; }
inline SdfFormV triangle(const Context& c, const ft_triangle& d) { return {check(ft_form_triangle(c.get(), &d)), c.get()}; }
}  // namespace Primitive
inline SdfFormV unionOf(const std::vector<SdfFormV>& forms) {
    if (forms.empty()) throw std::invalid_argument("No SdfObjects given.");
    auto h = nodes(forms); return {check(ft_form_union(forms[0].ctx, h.data(), (int)h.size())), forms[0].ctx};
}
inline SdfFormV subtract(SdfFormV a, SdfFormV b) { return {check(ft_form_subtract(a.ctx, a.Node, b.Node)), a.ctx}; }
inline SdfFormV intersect(const std::vector<SdfFormV>& forms) {
    if (forms.empty()) throw std::invalid_argument("No SdfObjects given.");
    auto h = nodes(forms); return {check(ft_form_intersect(forms[0].ctx, h.data(), (int)h.size())), forms[0].ctx};
}
inline SdfFormV unionSmooth(float strength, const std::vector<SdfFormV>& forms) {
    if (forms.empty()) throw std::invalid_argument("blub");
    auto h = nodes(forms); return {check(ft_form_union_smooth(forms[0].ctx, strength, h.data(), (int)h.size())), forms[0].ctx};
}
}  // namespace SdfForm

namespace SdfMaterial {
inline SdfMaterialV createSolid(const Context& c, FColor color) { return {check(ft_material_solid(c.get(), &color.v.x)), c.get()}; }
// EXTENSION (not in the reference): refracting material, see ft_material_glass
inline SdfMaterialV createGlass(const Context& c, FColor tint, float ior, float dispersion = 0.0f) {
    return {check(ft_material_glass(c.get(), &tint.v.x, ior, dispersion)), c.get()};
}
}

namespace SdfObject {
inline SdfObjectV create(SdfMaterialV material, SdfFormV form) { return {check(ft_object_create(form.ctx, material.Node, form.Node)), form.ctx}; }
inline SdfObjectV unionOf(const std::vector<SdfObjectV>& objects) {
    if (objects.empty()) throw std::invalid_argument("No SdfObjects given.");
    std::vector<ft_handle> h; for (auto& o : objects) h.push_back(o.Node);
    return {check(ft_object_union(objects[0].ctx, h.data(), (int)h.size())), objects[0].ctx};
}
inline SdfObjectV subtract(SdfObjectV object, SdfFormV form) { return {check(ft_object_subtract(object.ctx, object.Node, form.Node)), object.ctx}; }
inline SdfObjectV intersect(SdfObjectV object, const std::vector<SdfFormV>& forms) {
    auto h = nodes(forms); return {check(ft_object_intersect(object.ctx, object.Node, h.data(), (int)h.size())), object.ctx};
}
}  // namespace SdfObject

namespace SdfLight {
inline SdfLightV directional(const Context& c, Vector3 direction, FColor color) { return {check(ft_light_directional(c.get(), &direction.x, &color.v.x)), c.get()}; }
inline SdfLightV point(const Context& c, Vector3 position, FColor color) { return {check(ft_light_point(c.get(), &position.x, &color.v.x)), c.get()}; }
}

struct LensV { float NearPlaneSize; };
namespace Lens { inline LensV create(float fieldOfView) { return {ft_lens_create(fieldOfView)}; } }
namespace Camera {
struct LookAt { Vector3 Position, LookAtPoint, Up; LensV Lens; };
inline ft_camera lookAt(const LookAt& c) {
    ft_camera out; check(ft_camera_look_at(&c.Position.x, &c.LookAtPoint.x, &c.Up.x, c.Lens.NearPlaneSize, &out)); return out;
}
}
struct ImageSize { int X, Y; };

namespace Image {
// FColor[X,Y] as a flat vector, x-major / y contiguous (Array2D.fs:30-38): element (x, y) at 3 * (x * Y + y)
inline std::vector<float> renderScene(float epsilon, float length, ImageSize size, const ft_camera& camera, const SdfScene& scene,
                                      ft_stats* stats = nullptr) {
    ft_ctx* ctx = scene.Object.ctx;
    std::vector<ft_handle> lights; for (auto& l : scene.Lights) lights.push_back(l.Node);
    ft_scene* s = nullptr;
    check(ft_scene_create(ctx, scene.Object.Node, &scene.BackgroundColor.v.x, lights.data(), (int)lights.size(), &s));
    std::vector<float> out((size_t)size.X * size.Y * 3);
    ft_render_params p{size.X, size.Y, 0, size.X, size.X, 1, 0, 1, epsilon, length, 0, 0.0f, 0, 0};
    ft_stats st{};
    int rc = ft_render(ctx, s, &camera, &p, out.data(), &st);
    ft_scene_destroy(s);
    check(rc);
    if (stats) *stats = st;
    return out;
}
// Image.toColors gamma rng image (Image.fs:37-50) on the GPU: bytes in Color[X,Y] order (R,G,B) or, with bmpOrder, in the scan-line
// order of Image.toBitmap (Image.fs:61-86: rows from the top, B,G,R).  seed < 0: no dithering noise (the reference's is racy).
inline std::vector<unsigned char> toColors(ft_ctx* ctx, float gamma, long long seed, const std::vector<float>& image, ImageSize size, bool bmpOrder = false) {
    std::vector<unsigned char> out((size_t)size.X * size.Y * 3);
    ft_tonemap_params tm{gamma, seed >= 0 ? 1 : 0, (uint32_t)(seed >= 0 ? seed : 0), bmpOrder ? 1 : 0};
    check(ft_tone_map_host(ctx, image.data(), size.X, size.Y, &tm, out.data(), nullptr));
    return out;
}
}  // namespace Image

// SdfObject.tryTrace / SdfForm.tryTrace (SdfObject.fs:66-78, SdfForm.fs:93-104) over a ray buffer; hit == 0 is ValueNone
inline std::vector<ft_object_trace_result> tryTrace(const SdfObjectV& object, const std::vector<ft_ray>& rays) {
    const float bg[3] = {0.0f, 0.0f, 0.0f};
    ft_scene* s = nullptr;
    check(ft_scene_create(object.ctx, object.Node, bg, nullptr, 0, &s));
    std::vector<ft_object_trace_result> out(rays.size());
    int rc = ft_object_try_trace(object.ctx, s, rays.data(), (int64_t)rays.size(), out.data(), nullptr);
    ft_scene_destroy(s);
    check(rc);
    return out;
}
inline std::vector<ft_form_trace_result> formTryTrace(const SdfObjectV& object, const std::vector<ft_ray>& rays) {
    const float bg[3] = {0.0f, 0.0f, 0.0f};
    ft_scene* s = nullptr;
    check(ft_scene_create(object.ctx, object.Node, bg, nullptr, 0, &s));
    std::vector<ft_form_trace_result> out(rays.size());
    int rc = ft_form_try_trace(object.ctx, s, rays.data(), (int64_t)rays.size(), out.data(), nullptr);
    ft_scene_destroy(s);
    check(rc);
    return out;
}

}  // namespace FrayTracer
