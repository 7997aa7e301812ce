#!/usr/bin/env python3
"""timing ablations of the stream GEMM (gemm4.hip): what a k-tile costs without its LDS-DMA, barrier, fragment reads, packing, stores"""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from r3g import ffi
from bench_gemm import make
ffi.context(0)
L = ffi.lib()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
masks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,16,17,19,4,8,12,31").split(",")]
shapes = [(30080, 4096, 1024, 1), (131072, 4096, 1024, 2), (30080, 3072, 1024, 0)]
for (M, N, K, epi) in shapes:
    a, w, bias, gate, c0 = make(M, N, K, epi, 1)
    c = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    def run(v):
        ffi.check(L.r3g_set_option(b"gemm_waves", v))
        ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, None, M, N, K, epi, 1, s))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    res = {}
    for rnd in range(4):
        for key in ["v12"] + masks:
            if key == "v12":
                ffi.check(L.r3g_set_option(b"gemm4_ablate", 0)); fn = lambda: run(12)
            else:
                ffi.check(L.r3g_set_option(b"gemm4_ablate", key)); fn = lambda: run(14)
            fn(); ev[0].record()
            for _ in range(5): fn()
            ev[1].record(); torch.cuda.synchronize()
            if rnd: res.setdefault(key, []).append(ev[0].elapsed_time(ev[1]) / 5 * 1e3)
    tiles = ((M + 255) // 256) * (N // 256)
    per_wg = -(-tiles // 256)
    for key, ts in res.items():
        med = statistics.median(ts)
        print("shape %s  %-5s  %8.1f us   %6.0f TF/s   %.2f us per tile per CU" % ((M, N, K, epi), key, med, 2.0 * M * N * K / med / 1e6, med / per_wg), flush=True)
ffi.check(L.r3g_set_option(b"gemm4_ablate", 0)); ffi.check(L.r3g_set_option(b"gemm_waves", 0))
