#!/usr/bin/env python3
"""GEMM with warm vs cold weights: the same W every launch (stays in L2 / Infinity Cache) against a rotation over
more weight matrices than the 256 MB Infinity Cache holds (what the DiT loop sees: 2.2 GB of weights per forward)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import torch
from r3g import ffi

ffi.context(0)
L = ffi.lib()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, epi) in [(7552, 4096, 1024, 1), (7552, 3072, 1024, 0), (7552, 1024, 5120, 3), (7552, 1024, 1024, 3)]:
    nW = max(2, int(600e6 // (N * K * 2)))
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(nW)]
    bias = torch.randn(N, device="cuda")
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi >= 3 else torch.bfloat16)
    def run(i):
        w = ws[i]
        ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, None, M, N, K, epi, 1, s))
    res = {}
    for mode in ("warm", "cold", "prefetched"):
        for i in range(4): run(i % nW)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 2 * nW
        tot = 0.0
        if mode == "prefetched":
            # touch the next weight (a reduction reads every byte) before timing its GEMM; only the GEMM is timed
            for i in range(iters):
                ws[i % nW].view(torch.int16).sum()
                e0.record(); run(i % nW); e1.record(); torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
        else:
            e0.record()
            for i in range(iters): run(0 if mode == "warm" else i % nW)
            e1.record(); torch.cuda.synchronize()
            tot = e0.elapsed_time(e1)
        res[mode] = round(2.0 * M * N * K / (tot / iters) / 1e9, 1)
    print(json.dumps(dict(M=M, N=N, K=K, epi=epi, weights_in_rotation=nW, tflops=res)), flush=True)
