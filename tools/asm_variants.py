#!/usr/bin/env python3
"""Machine-code placement experiments on the C3 inner loop.  Compiles kernels.hip to gfx950 assembly,
edits the near-regime loop of ft_trace_kernel_smooth_spheres (LBB*: the loop after the first `.rept <pad>` block), and
re-assembles every variant into tools/_asmvar/<name>.hsaco.  tools/asm_variants_run.py times them on the GPU through the
diagnostic library (`make -C fraytracer_amd/csrc experiment`).

    python tools/asm_variants.py nopscan      one s_nop inserted before instruction i of the loop, i = 0, 8, 16, ...
    python tools/asm_variants.py base         the unmodified kernel with the loop at both 8-byte phases
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_asmvar")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = "-O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()


def base_asm():
    s = os.path.join(OUT, "kernels.s")
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *FLAGS, "--cuda-device-only", "-S",
                           os.path.join(ROOT, "fraytracer_amd", "csrc", "kernels.hip"), "-o", s], stderr=subprocess.DEVNULL)
    return open(s).read().splitlines()


def find_loop(lines, want=os.environ.get("FT_ASM_LOOP", "mul:4")):
    """(index of the '.rept' line, first and last instruction line) of the near loop of ft_trace_kernel_smooth_spheres whose body
    contains `want` (default: the strength -4 variant, the one C3 runs; FT_ASM_LOOP=v_mul_f32_e32 picks a general-strength loop)"""
    f0 = next(i for i, l in enumerate(lines) if l.startswith("ft_trace_kernel_smooth_spheres:"))
    f1 = next(i for i in range(f0 + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    for rept in (i for i in range(f0, f1) if lines[i].strip().startswith(".rept")):
        head = next((i for i in range(rept, f1) if re.match(r"\.LBB\d+_\d+:", lines[i]) and "Inner Loop Header" in lines[i + 2]), None)
        if head is None:
            continue
        label = lines[head].split(":")[0]
        first = next(i for i in range(head, f1) if lines[i].startswith("\t") and not lines[i].strip().startswith(";"))
        last = next(i for i in range(first, f1) if lines[i].strip().startswith("s_cbranch") and label in lines[i])
        body = "\n".join(lines[first:last + 1])
        if want in body and "v_lshl_add_u32" in body:
            return rept, first, last
    raise SystemExit(f"no near loop containing '{want}' in ft_trace_kernel_smooth_spheres")


def is_instr(l):
    t = l.strip()
    return l.startswith("\t") and t and not t.startswith(";") and not t.startswith(".")


def assemble(lines, name):
    s = os.path.join(OUT, name + ".s")
    open(s, "w").write("\n".join(lines) + "\n")
    o = os.path.join(OUT, name + ".o")
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", o, "-o", os.path.join(OUT, name + ".hsaco")])
    os.remove(o); os.remove(s)


def with_pad(lines, rept, pad):
    out = list(lines)
    out[rept] = f"\t.rept {pad}"
    return out


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "base"
    lines = base_asm()
    rept, first, last = find_loop(lines)
    idx = [i for i in range(first, last + 1) if is_instr(lines[i])]
    print(f"loop: {len(idx)} instructions, lines {first}..{last}", file=sys.stderr)
    if "--keep" not in sys.argv:
        for f in os.listdir(OUT):
            if f.endswith(".hsaco"):
                os.remove(os.path.join(OUT, f))
    else:
        sys.argv.remove("--keep")
    if mode == "base":
        for pad in (14, 15):
            assemble(with_pad(lines, rept, pad), f"base_pad{pad}")
    elif mode == "nopscan":
        step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        for pad in (14, 15):
            assemble(with_pad(lines, rept, pad), f"base_pad{pad}")
            for k in range(0, len(idx), step):
                v = with_pad(lines, rept, pad)
                v.insert(idx[k], "\ts_nop 0")
                assemble(v, f"pad{pad}_nop_before_{k:03d}")
    elif mode == "batch":
        # python tools/asm_variants.py batch name:pad:k1,k2,... [...]: s_nops before the listed instruction indices of the loop
        for spec in sys.argv[2:]:
            name, pad, ks = spec.split(":")
            v = with_pad(lines, rept, int(pad))
            for k in sorted((int(t) for t in ks.split(",") if t), reverse=True):
                v.insert(idx[k], "\ts_nop 0")
            assemble(v, name)
    elif mode == "rewrite":
        # VOP2 instructions with a 32-bit literal as src0 -> the dedicated-literal opcodes (same size, same result):
        #   v_fmac_f32 D, K, S  ->  v_fmamk_f32 D, S, K, D        v_add_f32 D, K, S  ->  v_fmaak_f32 D, 1.0, S, K
        which = sys.argv[2] if len(sys.argv) > 2 else "fmac,add"
        for pad in (14, 15):
            v = with_pad(lines, rept, pad)
            for i in idx:
                l = v[i]
                m = re.match(r"\tv_fmac_f32_e32 (v\d+), (0x[0-9a-f]+), (v\d+)", l)
                if m and "fmac" in which: v[i] = f"\tv_fmamk_f32 {m.group(1)}, {m.group(3)}, {m.group(2)}, {m.group(1)}"
                m = re.match(r"\tv_add_f32_e32 (v\d+), (0x[0-9a-f]+), (v\d+)", l)
                if m and "add" in which: v[i] = f"\tv_fmaak_f32 {m.group(1)}, 1.0, {m.group(3)}, {m.group(2)}"
            assemble(v, f"rewrite_{which.replace(',', '_')}_pad{pad}")
    elif mode == "custom":
        # python tools/asm_variants.py custom <name> <pad> <k1,k2,...>: s_nops before the listed instruction indices
        name, pad, ks = sys.argv[2], int(sys.argv[3]), sorted(int(t) for t in sys.argv[4].split(","))
        v = with_pad(lines, rept, pad)
        for k in reversed(ks):
            v.insert(idx[k], "\ts_nop 0")
        assemble(v, name)
    print("\n".join(sorted(os.listdir(OUT))))


if __name__ == "__main__":
    main()
