#!/usr/bin/env python3
"""Per-phase s_memtime sums of the phased attention kernel (attn_generation 9, option attn_stamps): one launch per shape, the
library prints work / barrier-wait ticks per block and wave for each of the three groups to stderr."""
import ctypes
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
for (B, H, Lq, Lk, shared) in [(1, 16, 131072, 3072, 1), (2, 16, 4442, 4442, 0)]:
    lqp, lkp = (Lq + 127) // 128 * 128, (Lk + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(1)
    Q = torch.randn(B, H, lqp, 64, device="cuda", generator=g).to(torch.bfloat16)
    K = torch.randn(1 if shared else B, H, lkp, 64, device="cuda", generator=g).to(torch.bfloat16)
    Vt = torch.randn(1 if shared else B, H, 64, lkp, device="cuda", generator=g).to(torch.bfloat16)
    o = torch.zeros(B, Lq, H * 64, device="cuda", dtype=torch.bfloat16)
    ffi.check(L.r3g_set_option(b"attn_generation", 9))
    for prio in (0, 1, 2):
        ffi.check(L.r3g_set_option(b"attn_prio", prio))
        ffi.check(L.r3g_set_option(b"attn_stamps", 0))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(2):
            ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp, shared, 1, s))
        ev[0].record()
        for _ in range(5):
            ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp, shared, 1, s))
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / 5 * 1e3
        sys.stderr.write("shape %s prio %d: %.1f us, %.0f TF/s\n" % ((B, H, Lq, Lk), prio, us, 4.0 * B * H * Lq * Lk * 64 / us / 1e6))
        sys.stderr.flush()
        ffi.check(L.r3g_set_option(b"attn_stamps", 1))
        ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp, shared, 1, s))
        torch.cuda.synchronize()
ffi.check(L.r3g_set_option(b"attn_stamps", 0))
ffi.check(L.r3g_set_option(b"attn_prio", 1))
ffi.check(L.r3g_set_option(b"attn_generation", 7))
