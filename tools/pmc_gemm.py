#!/usr/bin/env python3
"""A handful of launches of one kernel for `rocprofv3 --pmc ...` (counter collection needs few dispatches).
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... -- python tools/pmc_gemm.py gemm|attn"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import torch  # noqa: E402
from r3g import ffi  # noqa: E402

ffi.context(0)
L = ffi.lib()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
if which == "gemm":
    M, N, K = 8960, 4096, 1024
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, None, c.data_ptr(), N, None, M, N, K, 0, 1, s))
elif which == "gemm8":   # the phased 256x256 kernel on one shape: gemm8 M N K epilogue
    M, N, K, epi = (int(x) for x in sys.argv[2:6])
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi >= 3 else torch.bfloat16)
    ffi.check(L.r3g_set_option(b"gemm_waves", 11))
    for _ in range(3):
        ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, None, M, N, K, epi, 1, s))
else:
    B, H, Lq, Lk = 2, 16, 4442, 4442
    lqp, lkp = 4480, 4480
    Q = torch.randn(B, H, lqp, 64, device="cuda").to(torch.bfloat16)
    Kt = torch.randn(B, H, lkp, 64, device="cuda").to(torch.bfloat16)
    Vt = torch.randn(B, H, 64, lkp, device="cuda").to(torch.bfloat16)
    o = torch.zeros(B, Lq, H * 64, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ffi.check(L.r3g_op_attention(Q.data_ptr(), Kt.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp, 0, 1, s))
torch.cuda.synchronize()
