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
