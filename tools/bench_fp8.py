#!/usr/bin/env python3
"""FP8 (e4m3, row scales) GEMM against the bf16 phased kernel: K sweep at M = 32768, N = 1024 (slope = k-loop rate,
intercept = fixed cost per launch) and the geo decoder's MLP shapes.   python tools/bench_fp8.py"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import torch  # noqa: E402
from r3g import ffi  # noqa: E402

ffi.context(0)
L = ffi.lib()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def t(fn, n=10):
    fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


SHAPES = [(32768, 1024, 1024, 0), (32768, 1024, 4096, 0), (32768, 1024, 16384, 0), (131072, 4096, 1024, 2),
          (131072, 1024, 4096, 6)]
if os.environ.get("ZERO"):
    SHAPES = SHAPES[1:3]
for (M, N, K, epi) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    if os.environ.get("ZERO"):
        a.zero_()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    a8 = torch.empty(M, K, dtype=torch.uint8, device="cuda")
    w8 = torch.empty(N, K, dtype=torch.uint8, device="cuda")
    sa = torch.empty(M, device="cuda")
    sw = torch.empty(N, device="cuda")
    ffi.check(L.r3g_op_quant_fp8(a.data_ptr(), K, M, K, a8.data_ptr(), K, sa.data_ptr(), s))
    ffi.check(L.r3g_op_quant_fp8(w.data_ptr(), K, N, K, w8.data_ptr(), K, sw.data_ptr(), s))
    c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    us8 = t(lambda: ffi.check(L.r3g_op_gemm_fp8(a8.data_ptr(), K, sa.data_ptr(), w8.data_ptr(), K, sw.data_ptr(), bias.data_ptr(),
                                                  c.data_ptr(), N, None, M, N, K, epi, s)))
    us16 = t(lambda: ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, None, M, N, K,
                                               epi, 1, s)))
    usq = t(lambda: ffi.check(L.r3g_op_quant_fp8(a.data_ptr(), K, M, K, a8.data_ptr(), K, sa.data_ptr(), s)))
    fl = 2.0 * M * N * K
    print(json.dumps(dict(M=M, N=N, K=K, epi=epi, fp8_us=us8, bf16_us=us16, quant_us=usq, fp8_tflops=fl / us8 / 1e6,
                          bf16_tflops=fl / us16 / 1e6)), flush=True)
