// scene.hpp — host side of libfraytracer_hip: the reified scene tree the F# closures lack
//  and its flattening into the device layout of ft_device.h.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "ft_device.h"
#include "ft_math.h"

namespace ft {

// mirrors ft_status in include/fraytracer_hip.h
enum { FT_ERR_INVALID_ = -1, FT_ERR_UNSUPPORTED_ = -4, FT_ERR_EMPTY_ = -5 };

struct Boundary { f3 center; float radius; };                 // Types.fs:19-24

struct HostGrid {                                             // result of SdfBoundary.buildSpatialLookup
    f3 aabbMin, cellSize, cellSizeInv;
    int count[3];
    std::vector<f3> centers;                                  // [x,y,z] -> (x*cy + y)*cz + z
    std::vector<uint32_t> cellStart;                          // ncells + 1 (relative to this grid)
    std::vector<FtItem> items;                                // child = index into the union's child list
};

struct HostForm {                                             // reified SdfForm (Types.fs:40-44)
    enum Kind { SPHERE, CAPSULE, TORUS, TRIANGLE, BOX, UNION, SUBTRACT, INTERSECT, SMOOTH } kind;
    Boundary boundary;
    std::vector<float> params;                                // primitives: constant-pool record (ft_device.h strides)
    std::vector<int> kids;                                    // combinators: child form handles, reference order
    float strength = 0.0f;                                    // SMOOTH
    std::shared_ptr<HostGrid> grid;                           // UNION
    bool isPrim() const { return kind <= BOX; }
};

struct HostObject {                                           // reified SdfObject (Types.fs:51-55)
    enum Kind { CREATE, UNION, SUBTRACT, INTERSECT } kind;
    int form = -1;                                            // handle of object.Form
    int material = -1;                                        // CREATE
    std::vector<int> kidObjects;                              // UNION
    int inner = -1;                                           // SUBTRACT / INTERSECT: the object whose material is kept
    std::vector<int> forms;                                   // SUBTRACT: [b]; INTERSECT: the extra forms
};

struct HostLight { FtLight dev; };

// Optional accelerator for the per-cell part of buildSpatialLookup: given the grid frame
// (aabbMin, cellSize, count) and the item boundaries it must fill centers / cellStart / items exactly as the
// host loop does; returns false (leaving g untouched) if it declines (no GPU, too many items, ...).
struct GridFiller {
    virtual ~GridFiller() {}
    virtual bool fill(HostGrid& g, const std::vector<Boundary>& bounds, float halfDiag, std::string& err) = 0;
};

struct Builder {                                              // one per ft_ctx
    std::vector<HostForm> forms;
    std::vector<HostObject> objects;
    std::vector<f3> materials;                                // colour (solid) or tint (glass)
    struct MatExt { uint32_t glass = 0; float ior = 1.0f, dispersion = 0.0f; };
    std::vector<MatExt> materialExt;                          // EXTENSION, same index as materials
    std::vector<HostLight> lights;
    std::string err;
    GridFiller* gridFiller = nullptr;                         // owned by the context; null = host build

    int sphere(f3 c, float r);
    int capsule(f3 from, f3 to, float r);
    int torus(f3 c, f3 n, float R, float r);
    int triangle(f3 v1, f3 v2, f3 v3, float r);
    int box(f3 c, f3 half);
    int formUnion(const int* kids, int n);
    int formSubtract(int a, int b);
    int formIntersect(const int* kids, int n);
    int formUnionSmooth(float strength, const int* kids, int n);
    int materialSolid(f3 rgb);
    int materialGlass(f3 tint, float ior, float dispersion);  // EXTENSION
    int objectCreate(int material, int form);
    int objectUnion(const int* objs, int n);
    int objectSubtract(int obj, int form);
    int objectIntersect(int obj, const int* forms, int n);
    int lightDirectional(f3 dir, f3 rgb);
    int lightPoint(f3 pos, f3 rgb);

    bool okForm(int h) const { return h >= 0 && (size_t)h < forms.size(); }
    bool okObject(int h) const { return h >= 0 && (size_t)h < objects.size(); }
};

struct FlatScene {                                            // host copy of everything that goes to HBM
    std::vector<FtInstr> instr;                               // main program [0, nMainInstr), then the sub-programs of call children
    uint32_t nMainInstr = 0;
    std::vector<float> consts;
    std::vector<FtGrid> grids;
    std::vector<FtChild> children;
    std::vector<float> cellCenters;
    std::vector<uint32_t> cellStart;                          // global CSR over all cells of all grids (+1)
    std::vector<FtItemRec> items;
    std::vector<FtLight> lights;
    std::vector<float> materials;
    std::vector<float> materialsExt;                          // EXTENSION: 4 floats per material (glass flag, ior, dispersion, 0)
    uint32_t nGlass = 0;                                      // glass materials in this scene
    uint32_t nSlots = 1;
    uint32_t nStage = 0;                                      // consts[0, nStage) is mirrored in LDS by every workgroup
    float nearR2 = 0.0f;                                      // see FtSceneDev::nearR2
    uint32_t fastQ = 0;                                       // see FtSceneDev::fastQ
    float escC[3] = {0, 0, 0}, escR = -1.0f;                  // see FtSceneDev::escR (support sphere of the scene; -1: none)
    float escRho2 = 0.0f;                                     // see FtSceneDev::escRho2
    uint32_t cullPc = 0xffffffffu;                            // see FtSceneDev::cullPc
    uint32_t fastPath = 0;
    FtCarve carve{};                                          // fastPath == 3 (ft_device.h "Carved union"); the two pointers are set at upload
    std::vector<FtItemRec> itemsT;                            // fastPath == 3: candidate lists with a terminator per cell
    std::vector<uint32_t> cellStartT;                         // fastPath == 3: byte offsets into itemsT
    float bg[3] = {0, 0, 0};
};

// Boundary algebra (SdfBoundary.fs:7-67) and grid build (SdfBoundary.fs:225-274)
Boundary boundaryUnion(Boundary a, Boundary b);
Boundary boundaryIntersection(Boundary a, Boundary b);
std::shared_ptr<HostGrid> buildSpatialLookup(const std::vector<Boundary>& bounds, std::string& err, GridFiller* filler = nullptr);

// flatten the tree under `object` (+ lights, background) into the device layout; false + err on failure
bool flatten(const Builder& b, int object, const float bg[3], const int* lights, int nLights, FlatScene& out, std::string& err);

// EXTENSION: wavelength bins of ft_render_params.spectral (rgb weight, Cauchy term); double arithmetic with
// + - * / only, rounded once to float, so every host produces the same table
void spectralTable(int nw, float out[][4]);

float lensCreate(float fov);                                                              // Camera.fs:11-14
void cameraLookAt(f3 pos, f3 lookAt, f3 up, float nearPlaneSize, f3 out[4]);              // Camera.fs:33-42

}  // namespace ft
