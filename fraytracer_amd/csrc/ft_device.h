// ft_device.h — layout of the flattened, immutable scene in HBM (shared by host flattener and kernels).
//
// The F# scene is a tree of closures (Types.fs:40-79).  It is flattened once into
//   * a linear, wave-uniform program of FtInstr over per-lane value slots (one slot per nesting level),
//   * a float constant pool holding what each primitive closure captures (SdfForm.fs:148-149,
//     182-188, 216-225), laid out per type with a fixed stride so a uniform loop reads it with
//     scalar loads and a per-lane union loop gathers it,
//   * for every `union` a uniform grid in CSR form (SdfBoundary.fs:225-274): cellStart / items
//     (LowerBound, child) sorted per cell, cell centres, and a child table (boundary, type, data,
//     material).
// Everything is read-only during rendering and a few KB to a few MB, i.e. L2 / scalar-cache resident.
#pragma once
#include <stdint.h>

enum FtOp : uint32_t {
    FT_OP_PRIM = 0,      // slot[dst] = prim(type, data)
    FT_OP_SETLEAF,       // leaf[dst] = aux                              (SdfObject.create: one solid material)
    FT_OP_SMOOTH_RUN,    // acc = flags&1 ? 0 : slot[dst]; for i<count: acc += exp(f0 * prim_i); slot[dst] = acc
    FT_OP_SMOOTH_ADD,    // slot[dst] = (flags&1 ? 0 : slot[dst]) + exp(f0 * slot[src])
    FT_OP_SMOOTH_FIN,    // slot[dst] = -log(slot[dst]) * f0             (SdfForm.fs:82)
    FT_OP_SUBTRACT,      // slot[dst] = Max(-slot[src], slot[dst])       (SdfForm.fs:46-47)
    FT_OP_ISECT_RUN,     // for i<count: if slot[dst] < maxDist(bound_i) then slot[dst] = Max(slot[dst], prim_i)  (SdfForm.fs:60-63)
    FT_OP_ISECT_APPLY,   // same test with bound at consts[aux], value = slot[src]
    FT_OP_UNION,         // slot[dst], leaf[dst] = grid union aux        (SdfForm.fs:22-34 + SdfObject.fs:27-46)
};

enum FtPrim : uint32_t {
    FT_PR_SPHERE = 0, FT_PR_CAPSULE = 1, FT_PR_TORUS = 2, FT_PR_TRIANGLE = 3, FT_PR_BOX = 4,
    FT_PR_CALL = 14,     // union child that is a combinator WITHOUT a union inside: evaluated on demand, like a primitive,
                         // by running the sub-program consts[data] = (first instr, end instr, result slot) (scene.cpp)
    FT_PR_SLOT = 15      // union child that contains a union itself: value pre-evaluated in slot `data`
};

// constant-pool strides (floats) per primitive type
#define FT_STRIDE_SPHERE 4      // C.xyz, r
#define FT_STRIDE_CAPSULE 12    // From.xyz, r | dir.xyz, _ | dirInv.xyz, _
#define FT_STRIDE_TORUS 12      // C.xyz, R | N.xyz, r | planeD, _, _, _
#define FT_STRIDE_TRIANGLE 52   // V1,r | V2,_ | V3,_ | v21 | v32 | v13 | v21' | v32' | v13' | nor | n21 | n32 | n13 (xyz_ each)
#define FT_STRIDE_BOX 8         // C.xyz, _ | H.xyz, _

struct FtInstr {                // 12 dwords
    uint32_t op, dst, src, type;
    uint32_t count, data, aux, flags;
    float f0, f1;
    uint32_t pad0, pad1;
};

struct FtGrid {                 // 12 dwords
    float aabbMin[3];
    float cellSizeInv[3];
    int32_t count[3];
    uint32_t cellBase;          // first cell of this grid in cellStart / cellCenters
    uint32_t childBase;         // first child of this union in children[]
    uint32_t nChildren;
};

struct FtChild {                // 8 dwords
    float bc[3]; float br;      // child.Boundary
    uint32_t type;              // FtPrim
    uint32_t data;              // constant-pool offset, or slot index for FT_PR_SLOT
    uint32_t mat;               // material index if the child is `SdfObject.create solid prim`
    uint32_t pad;
};

struct FtItem { float lowerBound; uint32_t child; };   // host-side grid build / introspection

// Device form of one (cell, candidate) pair: everything the union loop needs for both pruning tests and
// for locating the candidate's constants, in ONE 32-byte record (two dwordx4 loads from consecutive
// addresses; a lane walks its cell's list front to back) instead of item -> child table -> constants.
struct FtItemRec {              // 8 dwords
    float lowerBound;           // SpatialLookupItem.LowerBound (SdfBoundary.fs:214)
    float bc[3]; float br;      // candidate.Boundary
    uint32_t typeData;          // FtPrim in bits 0..3, constant-pool offset (or slot index) in bits 4..31
    uint32_t mat;               // material index of a `create solid prim` child
    uint32_t child;             // index in the union's child list (introspection)
};

enum FtLightType : uint32_t { FT_LIGHT_DIRECTIONAL = 0, FT_LIGHT_POINT = 1 };
struct FtLight {                // 8 dwords
    uint32_t type;
    float v[3];                 // directional: normalize(-direction) (SdfLight.fs:7); point: position
    float color[3];
    float pad;
};

struct FtSceneDev {             // passed by value as kernel argument
    const FtInstr* instr;
    const float* consts;
    const FtGrid* grids;
    const FtChild* children;
    const float* cellCenters;   // 3 floats per cell
    const uint32_t* cellStart;  // per grid: nCells+1 entries, absolute item indices
    const FtItemRec* items;
    const FtLight* lights;
    const float* materials;     // 3 floats per material (colour, or tint of a glass)
    uint32_t nInstr, nSlots, nLights, fastPath;   // nInstr: the main program; sub-programs of FT_PR_CALL children follow it
    float bg[3];
    uint32_t nStage;            // leading floats of consts[] that every workgroup stages into LDS
    float nearR2;               // |p|^2 <= nearR2  =>  every t of the fast sphere runs is >= -87 (exp result normal)
    uint32_t fastQ;             // 1: every union candidate admits the clamped fast sqrt (scene.cpp: unionFastQ)
    uint32_t nGlass;            // EXTENSION: glass materials in the scene
    float escC[3], escR;        // support sphere of the scene's form (scene.cpp supportOf): no evaluation outside of it, grown by epsilon, can be a hit;
                                // escR < 0: none known, or FT_OPT_ESCAPE = 0
    uint32_t mathFma;           // FT_OPT_MATH (set per launch, not by the flattener): 1 = glibc's FMA build of expf / logf, 0 = its SSE2 build
                                // (read by the *_libm kernels only)
    uint32_t cullPc;            // instruction of the main program whose sphere run the child-culling pass serves (kernels.hip "Exact child culling"): a staged fast
                                // SMOOTH_RUN of >= 32 children, the longest one; 0xffffffff: none
    float escRho2;              // the escape shortcut is taken only by rays that start within sqrt(escRho2) of escC: the bound on the float32 drift of
                                // the marched points that escR's padding covers holds from there (scene.cpp "drift of the marched points")
};

// "Carved union" kernels (FtSceneDev.fastPath == 3; kernels.hip ft_eval_carved): the whole program is ONE grid union of plain primitives
// followed by at most FT_CARVE_TAIL intersect / subtract steps with one primitive each — the shape of the reference's own scene,
// subtract(intersect(union [1000 tori], [sphere]), sphere) (Program.fs:67-77).  The flattener records the tail here and lays the union's
// candidate lists out a second time with a terminator behind every cell's list (LowerBound = +inf: the reference's test :30 fails on it
// for every running minimum, NaN included), so that the walk needs neither an end index nor a bounds check; cellStartT[cell] is the BYTE
// offset of the cell's first record in itemsT.
#define FT_CARVE_TAIL 2
#define FT_CARVE_MIXED 7u       // FtCarve.kind when the union's children are not all of one primitive kind
struct FtCarveOp { uint32_t op, type, data, bound; };   // op: FT_OP_ISECT_RUN (one child: consts[data], boundary consts[bound]) or FT_OP_SUBTRACT (consts[data])
#define FT_CARVE_FAST_SPHERE 0x100u   // FtCarveOp.type flag: a sphere whose parameters admit the clamped fast root (|c| <= 1e4, 2^-20 <= r <= 1e4, like a union candidate's)
                                      // and, under an intersect, whose pruning boundary is the sphere itself bit for bit: ONE root serves distance and bound
struct FtCarve {
    uint32_t kind;              // FtPrim of every child of the union, or FT_CARVE_MIXED
    uint32_t nTail;
    FtCarveOp tail[FT_CARVE_TAIL];
    const FtItemRec* itemsT;    // per cell: its candidates in list order, then one terminator record; one spare record at the very end
    const uint32_t* cellStartT; // per cell: byte offset into itemsT
};

struct FtStatsDev {
    unsigned long long rays_primary, rays_shadow, rays_ext, hits_primary, hits_shadow, sdf_evals, flags, wave_evals;
    unsigned long long coop_evals;            // evaluations done in latency mode (one ray per wave)
    unsigned long long cull_total, cull_skipped;   // lean kernel: (child, ray) pairs the culling pass looked at / dropped, in units of 64
    unsigned long long clk_shader, clk_ref;   // s_memtime / s_memrealtime ticks the first wave of block 0 lived (summed over launches):
                                              // their ratio x the constant s_memrealtime rate (100 MHz) = the shader clock the kernel ran at
};

#define FT_STEP_CAP (1u << 20)  // the reference has no cap (SdfForm.fs:93-104); see DESIGN.md "NaN / step cap"
#define FT_MAX_SLOTS 48
#define FT_FLAG_INIT 1u         // FtInstr.flags: accumulator starts at 0 (first child of a smooth union)
#define FT_FLAG_FAST 2u         // FtInstr.flags: sphere run whose parameters admit the guarded fast path (see scene.cpp)
#define FT_FLAG_LAZY 4u         // FtInstr.flags of an FT_OP_UNION that is child 0 of an intersect whose child 1 is a primitive: type / data / count = that
                                // primitive's kind, constant-pool offset and boundary offset (kernels.hip "lazy union")
#define FT_MAX_STAGE_FLOATS 12288   // <= 48 KB of the constant pool is mirrored in LDS
