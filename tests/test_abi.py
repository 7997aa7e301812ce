"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/r3g.h declares, and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "r3g.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(r3g_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    from r3g import ffi
    L = ffi.lib()
    names = declared_symbols()
    assert "r3g_mc_count" in names and "r3g_create" in names
    for n in names:
        assert hasattr(L, n), "libr3g.so does not export " + n
        assert n in ffi.SYMBOLS, "r3g/ffi.py does not bind " + n
    assert sorted(ffi.SYMBOLS) == names
    assert L.r3g_version() >= 100


def test_no_cpu_fallback():
    import torch
    from r3g import ffi, mc
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ffi.R3GError) as e:
        ffi.context(0)
    assert e.value.code == -3 and "no CPU path" in str(e.value)
    with pytest.raises(ValueError):
        mc.marching_cubes(torch.zeros(4, 4, 4), 0.0)   # CPU tensor is refused, not silently handled


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "3d-re-gen_amd")
    for d, _, files in os.walk(pkg):
        if "build" in d.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(d, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), (d, f)
                assert "oracle/" not in txt.replace("never includes anything from oracle/", ""), (d, f)


def test_binding_matches_the_prototypes():
    """every prototype of include/r3g.h against the ctypes table: number of parameters, and pointer / integer /
    floating kind of each one (a drifted binding corrupts the call instead of failing)"""
    import ctypes
    from r3g import ffi
    txt = open(os.path.join(ROOT, "include", "r3g.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = re.findall(r"\b([a-z_0-9 ]+?[\s\*]+)(r3g_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", txt, flags=re.S)
    assert len(protos) == len(ffi.SYMBOLS)

    def kind_c(param):
        p = " ".join(param.split())
        if "*" in p:
            return "ptr"
        if re.search(r"\b(double|float)\b", p):
            return "flt"
        return "int"

    def kind_py(t):
        if t in (ctypes.c_double, ctypes.c_float):
            return "flt"
        if t in (ctypes.c_int, ctypes.c_int64, ctypes.c_int32):
            return "int"
        return "ptr"       # c_void_p, c_char_p, POINTER(...)

    for ret, name, params in protos:
        params = params.strip()
        plist = [] if params in ("", "void") else [q for q in params.split(",")]
        res, args = ffi.SYMBOLS[name]
        assert len(plist) == len(args), name
        assert [kind_c(q) for q in plist] == [kind_py(t) for t in args], name
        if "*" in ret:
            assert res is ctypes.c_char_p, name
        elif "void" in ret:
            assert res is None, name
        else:
            assert res is ctypes.c_int, name


def test_library_was_built_from_these_sources():
    """libr3g.so travels to the GPU box prebuilt (it is git-ignored but part of the snapshot): the digest stamped beside it
    by 3d-re-gen_amd/build.py must equal the digest of the sources, headers and flags in this tree"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("r3g_build", os.path.join(ROOT, "3d-re-gen_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.built_digest() is not None, "libr3g.digest missing: run python 3d-re-gen_amd/build.py"
    assert b.built_digest() == b.source_digest(), "libr3g.so is stale: rebuild with python 3d-re-gen_amd/build.py"


def test_every_option_is_documented_and_settable():
    """r3g_set_option's names (csrc/model.cpp) against the list in include/r3g.h: an option nobody can find, or a documented
    one that the library refuses, is a bug in the boundary.  Setting needs no GPU (plain host state)."""
    import re
    src = open(os.path.join(ROOT, "3d-re-gen_amd", "csrc", "model.cpp")).read()
    hdr = open(os.path.join(ROOT, "include", "r3g.h")).read()
    body = src[src.index("int r3g_set_option("):]
    body = body[:body.index("\n}\n")]
    names = re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', body)
    assert len(names) >= 20 and len(set(names)) == len(names)
    doc = hdr[hdr.index("/* A/B switches for tests and ablations."):hdr.index("int r3g_set_option(")]
    missing = [n for n in names if '"%s"' % n not in doc]
    assert not missing, "options without documentation in include/r3g.h: %s" % missing
    from r3g import ffi
    L = ffi.lib()
    assert L.r3g_set_option(b"no_such_option", 1) != 0
    defaults = {"attn_generation": 7, "ln_rows": 0, "ln_fixed": 1, "overlap_mlp": 0, "gemm_waves": 0, "gemm_raster": -1,
                "gemm_num_cu": 256, "gemm_auto_rule": 1, "mc_rows": 16, "geo_fp8": 0, "gemm_splitk": 0, "attn_pipelined": 0,
                "attn_ablate": 0}
    for n in names:
        v = defaults.get(n, 1)
        assert L.r3g_set_option(n.encode(), v) == 0, n       # (re)sets the default: the other tests share this process
