#!/usr/bin/env python3
"""Build libr3g.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

    python 3d-re-gen_amd/build.py [--force]

Each translation unit is compiled to build/<name>.o and linked into 3d-re-gen_amd/libr3g.so.
mc_kernels.hip is compiled with -ffp-contract=off (its fp64 ambiguity tests and interpolation must
reproduce the sequential reference bit-for-bit); the MFMA kernels keep hipcc's default contraction.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libr3g.so")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-missing-braces",
          "-I" + os.path.join(os.path.dirname(HERE), "include"), "-I" + CSRC]
PER_FILE = {"mc_kernels.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newest_header():
    return max(os.path.getmtime(os.path.join(d, f))
               for d in (CSRC, os.path.join(os.path.dirname(HERE), "include"))
               for f in os.listdir(d) if f.endswith(".h"))


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), _newest_header())):
        return obj, False
    cmd = ["hipcc"] + COMMON + PER_FILE.get(src, []) + ["-c", path, "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda s: _compile(s, force), sources()))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(OUT):
        subprocess.check_call(["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs)
        if verbose:
            print("linked", OUT)
    elif verbose:
        print("up to date:", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
