#!/usr/bin/env python3
"""source_hash.py — one hash over the sources libfraytracer_hip.so is built from.

The Makefile writes it into build_hash.h (only when it changed), capi.cpp embeds it, and ft_build_info() returns it, so that
  * a test can tell whether the library it loaded was built from the sources lying next to it (tests/test_abi.py),
  * every bench line and every profiles/*_summary.txt carries the hash of the code it measured (bench.py, tools/profile.sh):
    a number can be tied to a source state, and `python source_hash.py --rev <commit>` says which commit that state is.
Round 2 lost an hour to a GPU suite that had silently run on an experiment build.

    python3 source_hash.py                 # hash of the work tree
    python3 source_hash.py --header F      # write `#define FT_SOURCE_HASH "..."` to F if it differs
    python3 source_hash.py --rev HEAD~3    # hash of the same files in a commit (needs git)
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# everything that decides what the shared object contains (paths relative to the repository root)
FILES = ["fraytracer_amd/csrc/kernels.hip", "fraytracer_amd/csrc/ft_math.h", "fraytracer_amd/csrc/ft_libm.h", "fraytracer_amd/csrc/ft_device.h",
         "fraytracer_amd/csrc/ft_kernels.h", "fraytracer_amd/csrc/capi.cpp", "fraytracer_amd/csrc/scene.cpp", "fraytracer_amd/csrc/scene.hpp",
         "fraytracer_amd/csrc/multi.cpp", "fraytracer_amd/csrc/loop_layout.py", "fraytracer_amd/csrc/Makefile", "include/fraytracer_hip.h"]


def _digest(read):
    h = hashlib.sha256()
    for rel in FILES:
        data = read(rel)
        h.update(rel.encode() + b"\0" + str(len(data)).encode() + b"\0" + data)
    return h.hexdigest()[:16]


def source_hash(root=ROOT):
    def read(rel):
        p = os.path.join(root, rel)
        return open(p, "rb").read() if os.path.exists(p) else b""
    return _digest(read)


def source_hash_of_rev(rev, root=ROOT):
    def read(rel):
        r = subprocess.run(["git", "-C", root, "show", f"{rev}:{rel}"], capture_output=True)
        return r.stdout if r.returncode == 0 else b""
    return _digest(read)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--header":
        text = f'#define FT_SOURCE_HASH "{source_hash()}"\n'
        path = sys.argv[2]
        if not os.path.exists(path) or open(path).read() != text:
            open(path, "w").write(text)
    elif len(sys.argv) >= 3 and sys.argv[1] == "--rev":
        print(source_hash_of_rev(sys.argv[2]))
    else:
        print(source_hash())
