#!/usr/bin/env python3
"""Where a tile of the persistent 256 x 256 GEMM spends its cycles: s_memtime sums per section of gemm8p_kernel's tile loop.

    python tools/gemm_stamps.py --build          (no GPU needed) compiles csrc/gemm.hip with -DR3G_GEMM_STAMPS and links
                                                 3d-re-gen_amd/libr3g_stamps.so from it and the ordinary objects
    python tools/gemm_stamps.py [--shapes ...]   (on the MI355X) runs each shape once on that library and prints, per wave of
                                                 workgroup 8, the cycles per tile of: wait (k-tile 1 staged, wait for k-tile 0 and
                                                 the previous stores, opening barriers) | k-loop | stage (bias, next tile located,
                                                 sources, k-tile 0 issued) | epilogue | tail (sources again, loop overhead)

The instrumented library is a measurement build: libr3g.so never contains the stamps.
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-re-gen_amd")
LIB = os.path.join(PKG, "libr3g_stamps.so")

SHAPES = [(131072, 1024, 1024, 0), (131072, 4096, 1024, 2), (30080, 4096, 1024, 1), (30080, 3072, 1024, 0), (131072, 1024, 2048, 0)]


def build():
    csrc, inc, bdir = os.path.join(PKG, "csrc"), os.path.join(ROOT, "include"), os.path.join(PKG, "build")
    obj = os.path.join(bdir, "gemm_stamps.o")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-missing-braces", "-DR3G_GEMM_STAMPS",
                           "-I" + inc, "-I" + csrc, "-c", os.path.join(csrc, "gemm.hip"), "-o", obj])
    objs = [os.path.join(bdir, f) for f in sorted(os.listdir(bdir)) if f.endswith(".o") and f not in ("gemm.o", "gemm_stamps.o")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, obj] + objs)
    print("linked", LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--variant", type=int, default=12)
    a_ = ap.parse_args()
    if a_.build:
        return build()
    os.environ["R3G_LIBRARY"] = LIB
    sys.path.insert(0, ROOT)
    sys.path.insert(0, PKG)
    import torch
    from r3g import ffi
    ffi.context(0)
    L = ffi.lib()
    L.r3g_debug_gemm_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    L.r3g_debug_gemm_stamps.restype = ctypes.c_int
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    names = ["wait", "k-loop", "stage", "epilogue", "-", "tail"]
    for (M, N, K, epi) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g)
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ffi.check(L.r3g_set_option(b"gemm_waves", a_.variant))
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        rounds = (tiles + 255) // 256
        for rep in range(3):   # the last run is reported (clocks settled)
            ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, None, M, N, K, epi, 1, s))
            torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, None, M, N, K, epi, 1, s))
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / 5 * 1e3
        out = (ctypes.c_ulonglong * 64)()
        assert L.r3g_debug_gemm_stamps(out) == 0
        print("## %d x %d x %d epilogue %d: %.1f us per launch, %d tiles = %d per workgroup; s_memtime ticks per tile, workgroup 8" % (M, N, K, epi, us, tiles, rounds))
        print("| wave | " + " | ".join(n for n in names if n != "-") + " | sum |")
        print("|---|---|---|---|---|---|---|")
        for wv in range(8):
            v = [out[wv * 8 + k] / rounds for k in range(6)]
            print("| %d | " % wv + " | ".join("%.0f" % v[k] for k in range(6) if names[k] != "-") + " | %.0f |" % sum(v))
        del a, w, c
    ffi.check(L.r3g_set_option(b"gemm_waves", 0))


if __name__ == "__main__":
    main()
