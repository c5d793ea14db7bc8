"""The F# binding (host/fsharp/FrayTracer.Hip.fs) cannot be compiled here (no .NET toolchain), so it is kept in step with the C
header mechanically: every [<DllImport>] must name a function include/fraytracer_hip.h declares, with the same number of
parameters, the same kind of parameter in every position and the same kind of result, and every [<Struct>] twin must list the
fields of its C struct in the C order with matching scalar types.  (The C side of the layouts — sizeof / offsetof — is
tests/test_abi.py.)  Reference interface: /root/reference/src/FrayTracer/Types.fs:9-79."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "fraytracer_hip.h")).read()
FS = open(os.path.join(ROOT, "host", "fsharp", "FrayTracer.Hip.fs")).read()


def strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_kind(t):
    """class of a C parameter / result type as it crosses P/Invoke"""
    t = re.sub(r"\bconst\b", "", t)
    stars = t.count("*") + t.count("[")
    base = re.sub(r"[\*\[\]\s]", "", t)
    if stars == 0:
        return {"int": "i32", "int32_t": "i32", "ft_handle": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "float": "f32",
                "void": "void"}[base]
    if base in ("ft_ctx", "ft_scene"):
        return "ptr" if stars == 1 else "ref:ptr"          # opaque handle / out-parameter (or array) of handles
    if base in ("void", "char", "uint8_t"):
        return "ptr"
    return "ref:" + base                                   # pointer to a struct / scalar / array of scalars


def c_functions():
    text = strip_c(HEADER)
    text = re.sub(r"typedef\s+(struct|enum)\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ft_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = "[" in a
                a = re.sub(r"\[[^\]]*\]", "", a)
                mm = re.match(r"(.*?[\s\*])(\w+)$", a)          # type + name
                ty = mm.group(1).strip() if mm and mm.group(1).strip() not in ("const", "") and not a.endswith("*") else a
                params.append(c_kind(ty + ("*" if arr else "")))
        out[name] = (c_kind(ret), params)
    return out


FS_SCALARS = {"int": "i32", "int64": "i64", "uint64": "u64", "uint32": "u32", "float32": "f32", "void": "void"}


def fs_kind(t):
    t = t.strip()
    t = re.sub(r"\[<\w+>\]\s*", "", t)
    if t == "nativeint":
        return "ptr"
    if t == "nativeint&":
        return "ref:ptr"
    if t.endswith("&") or t.endswith("[]"):
        return "ref:" + t.rstrip("&").rstrip("[]").strip()
    return FS_SCALARS[t]


def fs_imports():
    out = {}
    for m in re.finditer(r"\[<DllImport\(Lib\)>\]\s*extern\s+(\w+)\s+(ft_[a-z0-9_]+)\s*\(([^)]*)\)", FS):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = []
        if args:
            for a in args.split(","):
                a = a.strip()
                ty = a.rsplit(" ", 1)[0]
                params.append(fs_kind(ty))
        out[name] = (FS_SCALARS.get(ret, "ptr" if ret == "nativeint" else ret), params)
    return out


# which C pointee each F# by-reference / array type may stand for
REF_OK = {
    "ref:ptr": {"ref:ptr"},
    "ref:FrayTracer.SdfForm.Primitive.Sphere": {"ref:ft_sphere"}, "ref:FrayTracer.SdfForm.Primitive.Capsule": {"ref:ft_capsule"},
    "ref:FrayTracer.SdfForm.Primitive.Torus": {"ref:ft_torus"}, "ref:FrayTracer.SdfForm.Primitive.Triangle": {"ref:ft_triangle"},
    "ref:int": {"ref:ft_handle", "ref:int32_t"}, "ref:Vector3": {"ref:float", "ref:ft_vec3"}, "ref:Ray": {"ref:ft_ray"},
    "ref:FtFormTraceResult": {"ref:ft_form_trace_result"}, "ref:FtObjectTraceResult": {"ref:ft_object_trace_result"},
    "ref:FtStats": {"ref:ft_stats"}, "ref:FtCamera": {"ref:ft_camera"}, "ref:FtRenderParams": {"ref:ft_render_params"},
    "ref:FtTonemapParams": {"ref:ft_tonemap_params"}, "ref:float32": {"ref:float"},
}


def compatible(fs, c):
    if fs == c:
        return True
    if fs == "ptr":                               # nativeint: any pointer (pinned arrays, opaque handles, const char*)
        return c == "ptr" or c.startswith("ref:")
    return c in REF_OK.get(fs, set())


def test_every_dllimport_matches_a_declaration_of_the_header():
    cf, ff = c_functions(), fs_imports()
    assert len(cf) >= 45 and len(ff) >= 30, (len(cf), len(ff))
    for name, (fret, fparams) in ff.items():
        assert name in cf, f"{name}: imported by the F# binding but not declared in the header"
        cret, cparams = cf[name]
        assert len(fparams) == len(cparams), (name, fparams, cparams)
        assert compatible(fret, cret), (name, "result", fret, cret)
        for i, (fp, cp) in enumerate(zip(fparams, cparams)):
            assert compatible(fp, cp), (name, i, fp, cp)
    # the render path of INTEGRATION.md is bound
    for need in ("ft_ctx_create", "ft_scene_create", "ft_render", "ft_render_colors", "ft_tone_map_host", "ft_abi_version", "ft_ctx_set_option"):
        assert need in ff, need


def c_struct_fields(name):
    m = re.search(r"typedef\s+struct\s+" + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", strip_c(HEADER), flags=re.S)
    assert m, name
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        for n in names.split(","):
            fields.append((n.strip(), ty))
    return fields


def fs_struct_fields(name):
    m = re.search(r"type\s+" + name + r"\s*=\s*\{(.*?)\}", FS, flags=re.S)
    assert m, name
    return [(n.strip(), t.strip()) for n, t in re.findall(r"(\w+)\s*:\s*([\w.]+)", m.group(1))]


C2FS = {"int32_t": {"int"}, "uint32_t": {"uint32"}, "uint64_t": {"uint64"}, "float": {"float32"}, "ft_vec3": {"Vector3"}, "ft_ray": {"Ray"}}


def norm(n):
    return n.replace("_", "").lower()


def test_struct_twins_list_the_c_fields_in_order():
    for cname, fname in (("ft_render_params", "FtRenderParams"), ("ft_stats", "FtStats"), ("ft_camera", "FtCamera"),
                         ("ft_tonemap_params", "FtTonemapParams"), ("ft_form_trace_result", "FtFormTraceResult"),
                         ("ft_object_trace_result", "FtObjectTraceResult")):
        cf, ff = c_struct_fields(cname), fs_struct_fields(fname)
        assert len(cf) == len(ff), (cname, cf, ff)
        for (cn, ct), (fn, ftt) in zip(cf, ff):
            assert norm(cn) == norm(fn), (cname, cn, fn)
            assert ftt in C2FS[ct], (cname, cn, ct, ftt)


def test_abi_version_literal_and_qualified_reference_modules():
    v = int(re.search(r"#define FT_ABI_VERSION (\d+)", HEADER).group(1))
    assert int(re.search(r"let AbiVersion = (\d+)", FS).group(1)) == v
    # inside the same-named Hip modules the reference's modules are named in full (FrayTracer.SdfForm.union, ...): nothing
    # relies on how F# resolves `SdfForm.` from within `module SdfForm` of another namespace
    code = "\n".join(l.split("//")[0] for l in FS.splitlines() if not l.strip().startswith("//"))
    for mod in ("SdfForm", "SdfMaterial", "SdfObject", "SdfLight", "SdfScene"):
        for m in re.finditer(r"(?<![\w.])" + mod + r"\.[a-zA-Z]", code):
            raise AssertionError(f"unqualified {mod}. at offset {m.start()}: {code[max(0, m.start() - 40):m.start() + 40]!r}")
