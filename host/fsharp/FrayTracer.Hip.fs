// FrayTracer.Hip.fs — F# binding of libfraytracer_hip.so for JanWosnitza/FrayTracer.
//
// Add this file to src/FrayTracer/FrayTracer.fsproj AFTER Image.fs.  It does not change any reference
// type: every scene value is paired with the handle of its native twin (`Gpu*` records), built by
// constructors that have the reference's names, argument order and error behaviour, so a scene script
// switches from `SdfForm.` / `SdfObject.` / `SdfLight.` to `Hip.SdfForm.` / ... and from
// `scene |> FrayTracer.SdfScene.trace |> FrayTracer.Image.render eps len size camera` to `Hip.Image.renderScene eps len size camera scene`.
// The closures stay available (`.Form`, `.Object`, `.Light`) as the CPU path.
//
// NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE (no .NET toolchain there); the same C ABI calls are
// exercised by fraytracer_amd/api.py (Python) and host/cpp/FrayTracer.hpp (C++).  Header: include/fraytracer_hip.h.
namespace FrayTracer.Hip

open System
open System.Numerics
open System.Runtime.InteropServices
open FrayTracer

[<Struct; StructLayout(LayoutKind.Sequential)>]
type FtRenderParams =
    { Width : int; Height : int; X0 : int; NColumns : int
      StripeWidth : int; StripeRanks : int; StripeRank : int; Spp : int
      Epsilon : float32; Length : float32; AoSamples : int; AoRadius : float32
      MaxBounces : int; Spectral : int }

[<Struct; StructLayout(LayoutKind.Sequential)>]
type FtStats =
    { RaysPrimary : uint64; RaysShadow : uint64; RaysExt : uint64; HitsPrimary : uint64; HitsShadow : uint64
      SdfEvals : uint64; Flags : uint64; KernelMs : float32; CulledFraction : float32; WaveEvals : uint64
      ShaderMHz : float32; TailFraction : float32 }

/// ft_camera (48 B).  The reference's `Camera` (Camera.fs:16-22) is an ordinary F# record — a reference type with
/// automatic layout, NOT a [<Struct>] — so it cannot cross P/Invoke by reference; its four vectors are copied into
/// this blittable twin for the call (`FtCamera.ofCamera`).  `Lens` and `Camera.LookAt` are reference records too and
/// never cross the boundary (Camera.lookAt stays on the F# side).
[<Struct; StructLayout(LayoutKind.Sequential)>]
type FtCamera =
    { Position : Vector3; Forward : Vector3; UpScaled : Vector3; RightScaled : Vector3 }
    static member ofCamera (c : Camera) : FtCamera =
        { Position = c.Position; Forward = c.Forward; UpScaled = c.UpScaled; RightScaled = c.RightScaled }

/// ft_tonemap_params (16 B): FrayTracer.Image.toColors' gamma; Dither = 0 -> no noise (u = 0.5), 1 -> counter-based hash of (x, y, channel, Seed);
/// BmpOrder = 1 -> the scan0 buffer FrayTracer.Image.toBitmap builds (Image.fs:61-86) instead of Color[X,Y] as R,G,B bytes
[<Struct; StructLayout(LayoutKind.Sequential)>]
type FtTonemapParams = { Gamma : float32; Dither : int; Seed : uint32; BmpOrder : int }

/// ft_form_trace_result: SdfFormTraceResult voption (Types.fs:32-37), Hit = 0 is ValueNone
[<Struct; StructLayout(LayoutKind.Sequential)>]
type FtFormTraceResult = { Ray : Ray; Distance : float32; Hit : int }

/// ft_object_trace_result: SdfObjectTraceResult voption (Types.fs:57-65), Hit = 0 is ValueNone
[<Struct; StructLayout(LayoutKind.Sequential)>]
type FtObjectTraceResult = { Ray : Ray; Normal : Vector3; Color : Vector3; Hit : int; Reserved : int }

module Native =
    [<Literal>]
    let Lib = "fraytracer_hip"

    /// FT_ABI_VERSION this file was written against (include/fraytracer_hip.h); checked when the context is created
    [<Literal>]
    let AbiVersion = 5

    [<DllImport(Lib)>] extern int ft_abi_version()
    [<DllImport(Lib)>] extern nativeint ft_build_info()
    [<DllImport(Lib)>] extern int ft_ctx_create(int device, nativeint& ctx)
    // per-context switches (ft_option: 1 refill_min, 2 max_blocks_per_cu, 3 host_chunks, 4 host_pin, 5 tail_k, 6 math, 7 guided, 8 chunk); the library reads no environment
    [<DllImport(Lib)>] extern int ft_ctx_set_option(nativeint ctx, int option, int value)
    [<DllImport(Lib)>] extern void ft_ctx_destroy(nativeint ctx)
    [<DllImport(Lib)>] extern nativeint ft_last_error()
    // By-reference parameters are [<Struct>] types only.  The reference's primitive records (SdfForm.fs:118-212), Ray and
    // SdfBoundary (Types.fs:9-24) are [<Struct>] records of float32 / Vector3 fields: sequential, blittable.  Vector3 is a
    // blittable BCL struct.  Camera, Lens, Camera.LookAt, SdfForm, SdfMaterial, SdfObject, SdfScene are reference records
    // and are never passed: Camera goes through FtCamera, the others through integer handles.
    [<DllImport(Lib)>] extern int ft_form_sphere(nativeint ctx, FrayTracer.SdfForm.Primitive.Sphere& data)
    [<DllImport(Lib)>] extern int ft_form_capsule(nativeint ctx, FrayTracer.SdfForm.Primitive.Capsule& data)
    [<DllImport(Lib)>] extern int ft_form_torus(nativeint ctx, FrayTracer.SdfForm.Primitive.Torus& data)
    [<DllImport(Lib)>] extern int ft_form_triangle(nativeint ctx, FrayTracer.SdfForm.Primitive.Triangle& data)
    [<DllImport(Lib)>] extern int ft_form_union(nativeint ctx, int[] forms, int n)
    [<DllImport(Lib)>] extern int ft_form_subtract(nativeint ctx, int a, int b)
    [<DllImport(Lib)>] extern int ft_form_intersect(nativeint ctx, int[] forms, int n)
    [<DllImport(Lib)>] extern int ft_form_union_smooth(nativeint ctx, float32 strength, int[] forms, int n)
    [<DllImport(Lib)>] extern int ft_material_solid(nativeint ctx, Vector3& rgb)
    [<DllImport(Lib)>] extern int ft_material_glass(nativeint ctx, Vector3& tint, float32 ior, float32 dispersion)
    [<DllImport(Lib)>] extern int ft_object_create(nativeint ctx, int material, int form)
    [<DllImport(Lib)>] extern int ft_object_union(nativeint ctx, int[] objects, int n)
    [<DllImport(Lib)>] extern int ft_object_subtract(nativeint ctx, int obj, int form)
    [<DllImport(Lib)>] extern int ft_object_intersect(nativeint ctx, int obj, int[] forms, int n)
    [<DllImport(Lib)>] extern int ft_light_directional(nativeint ctx, Vector3& direction, Vector3& rgb)
    [<DllImport(Lib)>] extern int ft_light_point(nativeint ctx, Vector3& position, Vector3& rgb)
    [<DllImport(Lib)>] extern int ft_scene_create(nativeint ctx, int obj, Vector3& background, int[] lights, int n, nativeint& scene)
    [<DllImport(Lib)>] extern void ft_scene_destroy(nativeint scene)
    [<DllImport(Lib)>] extern int ft_form_try_trace(nativeint ctx, nativeint scene, Ray[] rays, int64 n, [<Out>] FtFormTraceResult[] out, FtStats& stats)
    [<DllImport(Lib)>] extern int ft_object_try_trace(nativeint ctx, nativeint scene, Ray[] rays, int64 n, [<Out>] FtObjectTraceResult[] out, FtStats& stats)
    [<DllImport(Lib)>] extern int ft_render(nativeint ctx, nativeint scene, FtCamera& camera, FtRenderParams& p, nativeint out, FtStats& stats)
    // a destination that is reused over many frames can be page-locked once (otherwise ft_render pins it for the duration of each call)
    [<DllImport(Lib)>] extern int ft_host_register(nativeint ctx, nativeint p, uint64 bytes)
    [<DllImport(Lib)>] extern int ft_host_unregister(nativeint ctx, nativeint p)
    // FrayTracer.Image.toColors on the GPU: host FColor[,] in, bytes out / render + tone map in one call (only 3 bytes per pixel cross PCIe)
    [<DllImport(Lib)>] extern int ft_tone_map_host(nativeint ctx, nativeint frame, int X, int Y, FtTonemapParams& p, nativeint out, float32& maxOut)
    [<DllImport(Lib)>] extern int ft_render_colors(nativeint ctx, nativeint scene, FtCamera& camera, FtRenderParams& p, FtTonemapParams& tm, nativeint out, float32& maxOut, FtStats& stats)

    /// one context for the process (GPU 0); there is no CPU fallback inside the library
    let ctx =
        lazy (if ft_abi_version () <> AbiVersion then failwithf "libfraytracer_hip has ABI %d, this binding was written for %d" (ft_abi_version ()) AbiVersion
              let mutable c = 0n
              if ft_ctx_create (0, &c) < 0 then failwith (Marshal.PtrToStringAnsi (ft_last_error ()))
              c)

    /// FT_OPT_MATH (6): 0 = the library's fixed exp / log / pow (same bits on every machine), 1 = glibc's expf / logf / powf restated
    /// on the GPU — what MathF.Exp / Log / Pow return under .NET on Linux x64, i.e. the CPU path of this very process
    let setMath (mode : int) = if ft_ctx_set_option (ctx.Value, 6, mode) < 0 then failwith (Marshal.PtrToStringAnsi (ft_last_error ()))

    /// FT_OPT_CULL (9), FT_OPT_ESCAPE (10), FT_OPT_LAZY_UNION (11), all on by default: the smooth-union kernel skips children whose terms cannot change
    /// the running float32 sum, a ray that can no longer come within epsilon of the scene's support sphere ends as a miss at once, and a union under
    /// an intersect stops at Items.[0] where the intersect's next child already decides.  All are exact — the frame and the ray / hit counters do
    /// not change; `false` makes the GPU do every evaluation, child and candidate the CPU path does.  (FT_OPT_CARVED (12) only picks the kernel: the
    /// specialised one for a union of primitives with at most two intersect / subtract steps behind it — Program.fs's own scene — or the interpreter;
    /// FT_OPT_REUSE (13, on) takes a ray's first evaluation from a value already known — a shadow ray's from the normal's centre probe, a primary ray's
    /// from one evaluation at the camera position per wave — where the reference computes it again for every ray.)
    let setExactShortcuts (on : bool) =
        for opt in [ 9; 10; 11 ] do
            if ft_ctx_set_option (ctx.Value, opt, (if on then 1 else 0)) < 0 then failwith (Marshal.PtrToStringAnsi (ft_last_error ()))

    let check (h : int) =
        if h < 0 then failwith (Marshal.PtrToStringAnsi (ft_last_error ())) else h

// The reference's closures are built LAZILY (round 4): FrayTracer.SdfForm.union / SdfObject.union run the O(cells x items) buildSpatialLookup on the
// CPU (SdfBoundary.fs:225-274, called at SdfForm.fs:19 and again at SdfObject.fs:26) — seconds for the 1000 tori of Program.fs, before a GPU frame of
// 1.3 ms.  `.Form` / `.Object` force them on first use (renderSceneCpu, or a caller mixing both paths); a scene that is only rendered on
// the GPU never builds them.  The native handle is made eagerly: it is what Image.renderScene needs.
type GpuForm =
    { FormCpu : Lazy<FrayTracer.SdfForm>; Node : int }
    member this.Form = this.FormCpu.Value
type GpuMaterial = { Material : FrayTracer.SdfMaterial; Node : int }
type GpuObject =
    { ObjectCpu : Lazy<FrayTracer.SdfObject>; Node : int }
    member this.Object = this.ObjectCpu.Value
type GpuLight = { Light : FrayTracer.SdfLight; Node : int }
type GpuScene = { Object : GpuObject; BackgroundColor : FColor; Lights : GpuLight list }        // Types.fs:74-79

[<CompilationRepresentation(CompilationRepresentationFlags.ModuleSuffix)>]
module SdfForm =
    let private c () = Native.ctx.Value
    let private nodes (forms : GpuForm[]) = forms |> Array.map (fun f -> f.Node)

    module Primitive =
        let sphere (data : FrayTracer.SdfForm.Primitive.Sphere) =                                           // SdfForm.fs:125-135
            let mutable d = data
            { FormCpu = lazy (FrayTracer.SdfForm.Primitive.sphere data); Node = Native.check (Native.ft_form_sphere (Native.ctx.Value, &d)) }
        let capsule (data : FrayTracer.SdfForm.Primitive.Capsule) =                                         // SdfForm.fs:145-170
            let mutable d = data
            { FormCpu = lazy (FrayTracer.SdfForm.Primitive.capsule data); Node = Native.check (Native.ft_form_capsule (Native.ctx.Value, &d)) }
        let torus (data : FrayTracer.SdfForm.Primitive.Torus) =                                             // SdfForm.fs:181-203
            let mutable d = data
            { FormCpu = lazy (FrayTracer.SdfForm.Primitive.torus data); Node = Native.check (Native.ft_form_torus (Native.ctx.Value, &d)) }
        let triangle (data : FrayTracer.SdfForm.Primitive.Triangle) =                                       // SdfForm.fs:214-268
            let mutable d = data
            { FormCpu = lazy (FrayTracer.SdfForm.Primitive.triangle data); Node = Native.check (Native.ft_form_triangle (Native.ctx.Value, &d)) }

    let union (forms : seq<GpuForm>) =                                                           // SdfForm.fs:14-40
        match forms |> Seq.toArray with
        | [||] -> failwith "No SdfObjects given."
        | [| form |] -> form
        | forms ->
            { FormCpu = lazy (forms |> Seq.map (fun f -> f.Form) |> FrayTracer.SdfForm.union)
              Node = Native.check (Native.ft_form_union (c (), nodes forms, forms.Length)) }

    let subtract (a : GpuForm) (b : GpuForm) =                                                   // SdfForm.fs:42-49
        { FormCpu = lazy (FrayTracer.SdfForm.subtract a.Form b.Form); Node = Native.check (Native.ft_form_subtract (c (), a.Node, b.Node)) }

    let intersect (forms : seq<GpuForm>) =                                                       // SdfForm.fs:51-67
        match forms |> Seq.toArray with
        | [||] -> failwith "No SdfObjects given."
        | [| form |] -> form
        | forms ->
            { FormCpu = lazy (forms |> Seq.map (fun f -> f.Form) |> FrayTracer.SdfForm.intersect)
              Node = Native.check (Native.ft_form_intersect (c (), nodes forms, forms.Length)) }

    let unionSmooth (strength : float32) (forms : seq<GpuForm>) =                                // SdfForm.fs:69-91
        match forms |> Seq.toArray with
        | [||] -> failwithf "blub"
        | [| sdf |] -> sdf
        | sdfs ->
            { FormCpu = lazy (sdfs |> Seq.map (fun f -> f.Form) |> FrayTracer.SdfForm.unionSmooth strength)
              Node = Native.check (Native.ft_form_union_smooth (c (), strength, nodes sdfs, sdfs.Length)) }

[<CompilationRepresentation(CompilationRepresentationFlags.ModuleSuffix)>]
module SdfMaterial =
    let createSolid (color : FColor) =                                                           // SdfMaterial.fs:4-7
        let mutable rgb = let (FColor v) = color in v
        { Material = FrayTracer.SdfMaterial.createSolid color; Node = Native.check (Native.ft_material_solid (Native.ctx.Value, &rgb)) }

    /// EXTENSION (no reference counterpart): glass.  The CPU closure is the solid tint (what the device renders
    /// with MaxBounces = 0); refraction exists on the device path only.
    let createGlass (tint : FColor) (ior : float32) (dispersion : float32) =
        let mutable rgb = let (FColor v) = tint in v
        { Material = FrayTracer.SdfMaterial.createSolid tint; Node = Native.check (Native.ft_material_glass (Native.ctx.Value, &rgb, ior, dispersion)) }

[<CompilationRepresentation(CompilationRepresentationFlags.ModuleSuffix)>]
module SdfObject =
    let private c () = Native.ctx.Value

    let create (material : GpuMaterial) (form : GpuForm) : GpuObject =                           // SdfObject.fs:6-10
        { ObjectCpu = lazy (FrayTracer.SdfObject.create material.Material form.Form)
          Node = Native.check (Native.ft_object_create (c (), material.Node, form.Node)) }

    let union (objects : seq<GpuObject>) : GpuObject =                                           // SdfObject.fs:12-48
        match objects |> Seq.toArray with
        | [||] -> failwith "No SdfObjects given."
        | [| o |] -> o
        | objects ->
            let nodes = objects |> Array.map (fun o -> o.Node)
            { ObjectCpu = lazy (objects |> Seq.map (fun o -> o.Object) |> FrayTracer.SdfObject.union)
              Node = Native.check (Native.ft_object_union (c (), nodes, nodes.Length)) }

    let subtract (object : GpuObject) (form : GpuForm) : GpuObject =                             // SdfObject.fs:50-54
        { ObjectCpu = lazy (FrayTracer.SdfObject.subtract object.Object form.Form)
          Node = Native.check (Native.ft_object_subtract (c (), object.Node, form.Node)) }

    let intersect (object : GpuObject) (forms : seq<GpuForm>) : GpuObject =                      // SdfObject.fs:56-64
        let forms = forms |> Seq.toArray
        let nodes = forms |> Array.map (fun f -> f.Node)
        { ObjectCpu = lazy (FrayTracer.SdfObject.intersect object.Object (forms |> Seq.map (fun f -> f.Form)))
          Node = Native.check (Native.ft_object_intersect (c (), object.Node, nodes, nodes.Length)) }

[<CompilationRepresentation(CompilationRepresentationFlags.ModuleSuffix)>]
module SdfLight =
    let directional (direction : Direction) (color : FColor) =                                   // SdfLight.fs:6-21
        let mutable d = direction
        let mutable rgb = let (FColor v) = color in v
        { Light = FrayTracer.SdfLight.directional direction color
          Node = Native.check (Native.ft_light_directional (Native.ctx.Value, &d, &rgb)) }

    let point (position : Position) (color : FColor) =                                           // SdfLight.fs:23-42
        let mutable p = position
        let mutable rgb = let (FColor v) = color in v
        { Light = FrayTracer.SdfLight.point position color
          Node = Native.check (Native.ft_light_point (Native.ctx.Value, &p, &rgb)) }

module Trace =
    let private withObjectScene (object : GpuObject) (f : nativeint -> 'a) =
        let ctx = Native.ctx.Value
        let mutable bg = Vector3.Zero
        let mutable handle = 0n
        Native.check (Native.ft_scene_create (ctx, object.Node, &bg, [||], 0, &handle)) |> ignore
        try f handle finally Native.ft_scene_destroy handle

    /// GPU sibling of `rays |> Array.map (FrayTracer.SdfObject.tryTrace object)` (SdfObject.fs:66-78)
    let objectTryTrace (object : GpuObject) (rays : Ray[]) : SdfObjectTraceResult voption[] =
        withObjectScene object (fun scene ->
            let out : FtObjectTraceResult[] = Array.zeroCreate rays.Length
            let mutable stats = Unchecked.defaultof<FtStats>
            Native.check (Native.ft_object_try_trace (Native.ctx.Value, scene, rays, int64 rays.Length, out, &stats)) |> ignore
            out |> Array.map (fun r ->
                if r.Hit = 0 then ValueNone
                else ValueSome { SdfObjectTraceResult.Ray = r.Ray; Normal = r.Normal; Color = FColor r.Color }))

    /// GPU sibling of `rays |> Array.map (FrayTracer.SdfForm.tryTrace object.Form)` (SdfForm.fs:93-104)
    let formTryTrace (object : GpuObject) (rays : Ray[]) : SdfFormTraceResult voption[] =
        withObjectScene object (fun scene ->
            let out : FtFormTraceResult[] = Array.zeroCreate rays.Length
            let mutable stats = Unchecked.defaultof<FtStats>
            Native.check (Native.ft_form_try_trace (Native.ctx.Value, scene, rays, int64 rays.Length, out, &stats)) |> ignore
            out |> Array.map (fun r ->
                if r.Hit = 0 then ValueNone else ValueSome { SdfFormTraceResult.Ray = r.Ray; Distance = r.Distance }))

module Image =
    /// GPU sibling of `scene |> FrayTracer.SdfScene.trace |> FrayTracer.Image.render epsilon length imageSize camera`
    /// (Image.fs:26-35 + SdfScene.fs:7-28).  Result layout = FColor[X,Y] as Array2D.Parallel.init makes it.
    let renderScene (epsilon : float32) (length : float32) (imageSize : ImageSize) (camera : Camera) (scene : GpuScene) : FColor[,] =
        let ctx = Native.ctx.Value
        let lights = scene.Lights |> List.map (fun l -> l.Node) |> List.toArray
        let mutable bg = let (FColor v) = scene.BackgroundColor in v
        let mutable handle = 0n
        Native.check (Native.ft_scene_create (ctx, scene.Object.Node, &bg, lights, lights.Length, &handle)) |> ignore
        try
            let image : FColor[,] = Array2D.zeroCreate imageSize.X imageSize.Y
            let pin = GCHandle.Alloc (image, GCHandleType.Pinned)                                // as Image.fs:77-86 pins its buffer
            try
                let mutable cam = FtCamera.ofCamera camera                                         // Camera is a reference record: copy into the blittable twin
                let mutable p =
                    { Width = imageSize.X; Height = imageSize.Y; X0 = 0; NColumns = imageSize.X
                      StripeWidth = imageSize.X; StripeRanks = 1; StripeRank = 0; Spp = 1
                      Epsilon = epsilon; Length = length; AoSamples = 0; AoRadius = 0f
                      MaxBounces = 0; Spectral = 0 }
                let mutable stats = Unchecked.defaultof<FtStats>
                Native.check (Native.ft_render (ctx, handle, &cam, &p, pin.AddrOfPinnedObject (), &stats)) |> ignore
                image
            finally
                pin.Free ()
        finally
            Native.ft_scene_destroy handle

    /// GPU sibling of `image |> FrayTracer.Image.toColors gamma rng` (Image.fs:37-50).  The reference draws its dithering noise from one
    /// System.Random shared by a parallel map (racy); here `seed = ValueNone` means no noise, `ValueSome s` a counter-based hash.
    let toColors (gamma : float32) (seed : uint32 voption) (image : FColor[,]) : System.Drawing.Color[,] =
        let X, Y = image.GetLength 0, image.GetLength 1
        let bytes : byte[] = Array.zeroCreate (X * Y * 3)
        let pinIn = GCHandle.Alloc (image, GCHandleType.Pinned)
        let pinOut = GCHandle.Alloc (bytes, GCHandleType.Pinned)
        try
            let mutable tm = { Gamma = gamma; Dither = (if seed.IsSome then 1 else 0); Seed = (match seed with ValueSome s -> s | ValueNone -> 0u); BmpOrder = 0 }
            let mutable mx = 0f
            Native.check (Native.ft_tone_map_host (Native.ctx.Value, pinIn.AddrOfPinnedObject (), X, Y, &tm, pinOut.AddrOfPinnedObject (), &mx)) |> ignore
            Array2D.init X Y (fun x y -> let i = (x * Y + y) * 3 in System.Drawing.Color.FromArgb (int bytes.[i], int bytes.[i + 1], int bytes.[i + 2]))
        finally
            pinOut.Free (); pinIn.Free ()

    /// the CPU path of the reference on the same scene value (closures)
    let renderSceneCpu (epsilon : float32) (length : float32) (imageSize : ImageSize) (camera : Camera) (scene : GpuScene) =
        { FrayTracer.SdfScene.Object = scene.Object.Object
          BackgroundColor = scene.BackgroundColor
          Lights = scene.Lights |> List.map (fun l -> l.Light) }
        |> FrayTracer.SdfScene.trace
        |> FrayTracer.Image.render epsilon length imageSize camera
