#!/usr/bin/env python3
"""Build step: machine-code placement of the sphere loops of the trace kernels.

Measured on gfx950 (tools/asm_variants*.py, tools/pad_sweep.py; profiles/r02_*): the 4-children-per-trip loops of
smooth_run_spheres_fast consist of ~72 four-byte instructions (distances, square roots) followed by ONE run of ~38
64-bit encoded VALU instructions (the exp part).  When that run starts on an 8-byte boundary a C3 frame takes 0.148-0.155 G
shader cycles, when it starts at 4 mod 8 it takes 0.1377-0.1395 G (-7 ... -10 %) — the same on every MI355X tried; which of
the two a plain compile produces is a lottery of the preceding code (a one-dword change flips it).  A scan of all single-s_nop
insertions (profiles/r02_asm_nopscan2.jsonl) shows the best layout: loop head at 4 mod 8 AND the run at 4 mod 8 (another 1 %),
which takes one s_nop inside the loop, anywhere in the 40 instructions before the run.  This pass removes the lottery:
every loop is preceded (kernels.hip FT_LOOP_PHASE) by `.p2align 6` + `.rept N` s_nops; the pass assembles the device
assembly, disassembles it, finds every such loop, sets N to 0 or 1 so that the loop head lands at 4 mod 8, inserts one
`s_nop 0` fourteen instructions ahead of the run where the run would otherwise start at 0 mod 8, and verifies the result.

    loop_layout.py fix  in.s out.s      (Makefile: between `hipcc -S --cuda-device-only` and the assembler)
    loop_layout.py check kernels.o      (exit status 1 if a loop of a trace kernel is in the slow phase)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("FT_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ARCH = os.environ.get("FT_ARCH", "gfx950")
MARK = "ft_loop_pad_"


def disassemble(obj):
    return subprocess.check_output([LLVM + "/llvm-objdump", "-d", obj], text=True)


def device_code_object(path, tmp):
    """the gfx950 code object hipcc embedded in a host object / shared library (offload bundle in .hip_fatbin)"""
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(tmp, "unused")])
    listing = subprocess.run([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + fat], capture_output=True, text=True).stdout
    target = next((t for t in listing.split() if ARCH in t), None)
    if target is None:
        raise SystemExit(f"{path}: no {ARCH} code object inside")
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + target, "--input=" + fat, "--output=" + co])
    return co


def sphere_loops(dis):
    """[(symbol, kind, [(address, size, text)])]: every backward-branch loop that holds exactly 4 v_rsq_f32"""
    out, sym, ins = [], None, []

    def flush():
        if sym and "carved" in sym:          # the carved-union kernels hold no sphere loop: four roots in one loop there are the candidate walk's
            return
        for addr, _, text in ins:
            m = re.match(r"s_cbranch_scc\d (\d+)", text)
            if not m:
                continue
            off = int(m.group(1))
            off = off - 65536 if off >= 32768 else off
            if off >= 0:
                continue
            target = addr + 4 + 4 * off
            body = [i for i in ins if target <= i[0] <= addr]
            # not the hot loops: the latency-mode evaluation (one ray per wave: its loop WRITES the wave's LDS row) and the glibc-mode
            # loops (double-precision exponentials: any *_f64 instruction) also hold four roots, but their placement is not what a frame's time depends on
            if any(i[2].startswith("ds_write") or "_f64" in i[2].split()[0] for i in body):
                continue
            if sum(1 for i in body if i[2].startswith("v_rsq_f32")) == 4:
                out.append((sym, "near" if any(i[2].startswith("v_lshl_add_u32") for i in body) else "far", body))

    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([\w.$]+)>:", line)
        if m:
            if not m.group(1).startswith(MARK):                # analysis labels do not end a function
                flush()
                sym, ins = m.group(1), []
            continue
        m = re.match(r"\s+(\S.*?)\s+// ([0-9A-F]+): ((?:[0-9A-F]{8} ?)+)", line)
        if m and sym:
            ins.append((int(m.group(2), 16), 4 * len(m.group(3).split()), m.group(1)))
    flush()
    return out


def longest_run(body):
    best, cur = [], []
    for i in body + [(0, 0, "")]:
        if i[1] == 8 and i[2].startswith("v_"):
            cur.append(i)
        else:
            if len(cur) > len(best):
                best = cur
            cur = []
    return best


def run_phase(body):
    run = longest_run(body)
    return run[0][0] % 8, run


def assemble(lines, tmp, name):
    s, o, co = (os.path.join(tmp, name + ext) for ext in (".s", ".o", ".co"))
    open(s, "w").write("\n".join(lines) + "\n")
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + ARCH, "-c", s, "-o", o])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", o, "-o", co])
    return co


NOP_AHEAD = 14          # instructions between the inserted s_nop and the first instruction of the 64-bit run


def text_loops(lines, marks):
    """for every FT_LOOP_PHASE marker (line index of its `.rept`): the text lines of the instructions of the loop that follows
    it, i.e. of the first backward branch after the marker whose body holds 4 v_rsq_f32 (None if there is none before the next marker)"""
    out = []
    for j, m in enumerate(marks):
        stop = marks[j + 1] if j + 1 < len(marks) else len(lines)
        labels, found = {}, None
        for i in range(m, stop):
            lm = re.match(r"(\.LBB\d+_\d+):", lines[i])
            if lm:
                labels[lm.group(1)] = i
            bm = re.match(r"\ts_cbranch_scc\d (\.LBB\d+_\d+)\s*$", lines[i])
            if bm and bm.group(1) in labels:
                body = [k for k in range(labels[bm.group(1)], i + 1) if lines[k].startswith("\t") and not lines[k].strip().startswith((";", "."))]
                if any(lines[k].strip().startswith("ds_write") or "_f64" in lines[k].split()[0] for k in body):
                    continue
                if sum(1 for k in body if lines[k].strip().startswith("v_rsq_f32")) == 4:
                    found = body
                    break
        out.append(found)
    return out


def fix(src, dst):
    lines = open(src).read().splitlines()
    marks = [i for i, l in enumerate(lines) if re.match(r"\s*\.rept \d+\s*$", l) and i > 0 and lines[i - 1].strip() == ".p2align 6"]
    if not marks:
        raise SystemExit("loop_layout: no FT_LOOP_PHASE marker in the assembly")
    pads = [0] * len(marks)
    nops = {}                                                    # marker -> instruction index inside its loop that gets an s_nop in front

    def render(with_labels):
        out = list(lines)
        inserts = []
        loops = text_loops(lines, marks)
        for j, i in enumerate(marks):
            out[i] = f"\t.rept {pads[j]}"
            if j in nops and loops[j] is not None:
                inserts.append((loops[j][nops[j]], "\ts_nop 0"))
            if with_labels:                                      # a symbol per marker: where it landed (objdump -t)
                inserts.append((i - 1, f"{MARK}{j}:"))
        for at, text in sorted(inserts, reverse=True):
            out.insert(at, text)
        return out

    def analyse(tmp):
        co = assemble(render(True), tmp, "probe")
        symtab = subprocess.check_output([LLVM + "/llvm-objdump", "-t", co], text=True)
        where = {int(m.group(2)): int(m.group(1), 16) for m in re.finditer(r"^([0-9a-f]+) .*\b" + MARK + r"(\d+)$", symtab, re.M)}
        res = []
        for sym, kind, body in sphere_loops(disassemble(co)):
            owner = max((j for j, a in where.items() if a <= body[0][0]), key=lambda j: where[j], default=None)
            if owner is None or body[0][0] - where[owner] > 256:
                raise SystemExit(f"loop_layout: the {kind} loop at {body[0][0]:#x} of {sym} has no FT_LOOP_PHASE marker in front of it")
            phase, run = run_phase(body)
            res.append((owner, sym, kind, body, phase, run))
        return res

    with tempfile.TemporaryDirectory() as tmp:
        for owner, sym, kind, body, phase, run in analyse(tmp):       # 1: loop heads to 4 mod 8
            if body[0][0] % 8 != 4:
                pads[owner] ^= 1
        for owner, sym, kind, body, phase, run in analyse(tmp):       # 2: runs to 4 mod 8 with one s_nop inside the loop
            if body[0][0] % 8 != 4:
                raise SystemExit("loop_layout: could not place a loop head at 4 mod 8")
            if phase != 4:
                at = next(k for k, ins in enumerate(body) if ins[0] == run[0][0]) - NOP_AHEAD
                if at < 1:
                    raise SystemExit("loop_layout: the 64-bit run starts too early in the loop for the s_nop")
                nops[owner] = at
        final = render(False)
        co = assemble(final, tmp, "final")
        report = []
        for sym, kind, body in sphere_loops(disassemble(co)):
            phase, run = run_phase(body)
            report.append(f"{sym}: {kind} loop {len(body)} instr / {body[-1][0] + 4 - body[0][0]} B, head {body[0][0] % 8} mod 8, 64-bit run of {len(run)} at {run[0][0]:#x} = {phase} mod 8")
            if phase != 4 or body[0][0] % 8 != 4:
                raise SystemExit("loop_layout: verification failed: " + report[-1])
    open(dst, "w").write("\n".join(final) + "\n")
    print(f"loop_layout: {len(report)} sphere loops placed (head and 64-bit run at 4 mod 8; pads {pads}, s_nop in loops {sorted(nops)})")
    return report


# What may stand between the two MODE writes of the NEAR sphere loop (kernels.hip ft_omod_on / ft_omod_off: IEEE off, f32 denormals flushed,
# so that output modifiers work): the loop's own float forms, and anything that does not look at the float mode (integer, lane, LDS, scalar).
MODE_REGION_OK = re.compile(r"v_(sub|mul|add|max|fma|fmac|fmamk|fmaak)_f32|v_rsq_f32|v_lshl_add_u32|v_mov_b32|v_readlane_b32|v_writelane_b32|"
                            r"v_(add|sub|subrev|lshlrev|lshrrev|and|or)_[ub]32|v_add3_u32|v_add_lshl_u32|v_cmp_\w+_[ui]32|v_cndmask_b32|ds_read_b128|s_|buffer_|scratch_|global_load")


def mode_regions(dis):
    """[(symbol, [instruction text])] for every stretch between the MODE write that follows an s_andn2 (on) and the next MODE write (off);
    an instruction with an output modifier outside such a stretch ends the build (the hardware would ignore the modifier there)"""
    out, sym, on, prev = [], None, None, ""
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([\w.$]+)>:", line)
        if m:
            if not m.group(1).startswith(MARK):
                sym = m.group(1)
            continue
        m = re.match(r"\s+(\S.*?)\s+// [0-9A-F]+:", line)
        if not m:
            continue
        text = m.group(1)
        if text.startswith("s_setreg_b32 hwreg(HW_REG_MODE, 4, 6)"):
            if prev.startswith("s_andn2_b32"):
                on = []
            elif on is not None:
                out.append((sym, on)); on = None
        elif on is not None:
            on.append(text)
        elif re.search(r" (mul:[24]|div:2)\b", text):
            raise SystemExit(f"loop_layout: {sym}: '{text}' carries an output modifier outside a mode region")
        prev = text
    if on is not None:
        raise SystemExit(f"loop_layout: {sym}: a MODE write without its restore")
    return out


def check(path, only_trace_kernels=True):
    with tempfile.TemporaryDirectory() as tmp:
        co = device_code_object(path, tmp) if not path.endswith((".co", ".hsaco")) else path
        bad = n = 0
        dis = disassemble(co)
        regions = mode_regions(dis)
        for sym, body in regions:
            foreign = sorted({t.split()[0] for t in body if not MODE_REGION_OK.match(t)})
            rsq = sum(1 for t in body if t.startswith("v_rsq_f32"))
            # the loop (4 roots) and its remainder (1); the self-test kernel's regions hold one form each
            ok = not foreign and (rsq == 5 or sym == "ft_selftest_kernel") and len(body) < 260
            bad += not ok
            print(f"{sym}: output-modifier region of {len(body)} instructions, {rsq} roots -> " + ("only the sphere loop inside" if ok else f"FOREIGN CODE inside: {foreign}"))
        if not regions:
            print("no output-modifier region in this build (-DFT_SQRT_5?)")       # tests/test_build_layout.py insists on them for the product library
        for sym, kind, body in sphere_loops(dis):
            if only_trace_kernels and not sym.startswith("ft_trace_kernel"):
                continue
            phase, run = run_phase(body)
            n += 1
            ok = phase == 4 and body[0][0] % 8 == 4
            bad += not ok
            print(f"{sym}: {kind} loop at {body[0][0]:#x} = {body[0][0] % 8} mod 8 ({len(body)} instructions, {body[-1][0] + 4 - body[0][0]} bytes), longest 64-bit VALU run = "
                  f"{len(run)} instructions at {run[0][0]:#x} = {phase} mod 8 -> {'fast phase' if ok else ('SLOW phase' if phase != 4 else 'fast phase, head not at 4 mod 8')}")
    if n == 0:
        raise SystemExit("loop_layout: no sphere loop found: the check needs updating")
    return bad


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "fix":
        fix(sys.argv[2], sys.argv[3])
    elif len(sys.argv) == 3 and sys.argv[1] == "check":
        sys.exit(1 if check(sys.argv[2]) else 0)
    else:
        raise SystemExit(__doc__)
