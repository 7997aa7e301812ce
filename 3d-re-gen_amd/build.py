#!/usr/bin/env python3
"""Build libr3g.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

    python 3d-re-gen_amd/build.py [--force]

Each translation unit is compiled to build/<name>.o and linked into 3d-re-gen_amd/libr3g.so.
mc_kernels.hip is compiled with -ffp-contract=off (its fp64 ambiguity tests and interpolation must
reproduce the sequential reference bit-for-bit); the MFMA kernels keep hipcc's default contraction.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.

Staleness is decided by CONTENT, not by mtimes: build/manifest.json records, per object, the SHA-256 of its source,
of every header it can include and of the compile flags, and for the library the hashes of all objects' inputs.
A snapshot that carries a prebuilt .so from different sources is therefore rebuilt, and `source_digest()` lets the
tests assert that the library they loaded was built from the sources next to it (tests/test_abi.py).
"""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libr3g.so")
MANIFEST = os.path.join(OBJ, "manifest.json")
STAMP = os.path.join(HERE, "libr3g.digest")     # travels with the .so (the GPU box gets both)
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-missing-braces", "-I" + INC, "-I" + CSRC]
PER_FILE = {"mc_kernels.hip": ["-ffp-contract=off"], "tex_kernels.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def _headers_digest():
    h = hashlib.sha256()
    for d in (CSRC, INC):
        for f in sorted(os.listdir(d)):
            if f.endswith(".h"):
                h.update(f.encode())
                h.update(_sha(os.path.join(d, f)).encode())
    return h.hexdigest()


def _unit_key(src, hdr):
    flags = " ".join(COMMON[:5] + PER_FILE.get(src, []))      # without the absolute -I paths
    return hashlib.sha256((_sha(os.path.join(CSRC, src)) + hdr + flags).encode()).hexdigest()


def source_digest():
    """digest of everything libr3g.so is built from (sources, headers, flags)"""
    hdr = _headers_digest()
    return hashlib.sha256("".join(_unit_key(s, hdr) for s in sources()).encode()).hexdigest()


def built_digest():
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def _load_manifest():
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _compile(src, key, old, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    if not force and os.path.exists(obj) and old.get(src) == key:
        return obj, False
    cmd = ["hipcc"] + COMMON + PER_FILE.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    old = _load_manifest()
    hdr = _headers_digest()
    keys = {s: _unit_key(s, hdr) for s in sources()}
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda s: _compile(s, keys[s], old.get("units", {}), force), sources()))
    objs = [o for o, _ in res]
    digest = source_digest()
    if any(ch for _, ch in res) or not os.path.exists(OUT) or built_digest() != digest:
        subprocess.check_call(["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs)
        with open(STAMP, "w") as f:
            f.write(digest + "\n")
        if verbose:
            print("linked", OUT)
    elif verbose:
        print("up to date:", OUT)
    with open(MANIFEST, "w") as f:
        json.dump({"units": keys, "library": digest}, f, indent=1)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
