#!/usr/bin/env python3
"""Micro-benchmarks of the MFMA kernels at the hot path's shapes (HIP events on the launch stream)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import torch  # noqa: E402
from r3g import ffi  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


SWEEP = [tuple(int(x) for x in item.split(':')) for item in
         os.environ.get('R3G_BENCH_SWEEP', '4:0,8:0,9:4').split(',')]   # waves:raster_group pairs


def main():
    ffi.context(0)
    L = ffi.lib()
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for (M, N, K, epi) in [(8960, 3072, 1024, 0), (8960, 4096, 1024, 1), (8960, 1024, 5120, 3), (8960, 1024, 1024, 3), (7552, 3072, 1024, 0), (7552, 4096, 1024, 1), (7552, 1024, 5120, 3), (6144, 1024, 4096, 3),
                           (131072, 4096, 1024, 2), (131072, 1024, 4096, 3), (131072, 1024, 1024, 0)]:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        c = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi >= 3 else torch.bfloat16)
        for waves, raster in SWEEP:
            ffi.check(L.r3g_set_option(b"gemm_waves", waves))
            ffi.check(L.r3g_set_option(b"gemm_raster", raster))
            ms = timeit(lambda: ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N,
                                                        None, M, N, K, epi, 1, s)))
            out.append(dict(op="gemm", M=M, N=N, K=K, epi=epi, waves=waves, raster=raster, ms=ms,
                            tflops=2.0 * M * N * K / ms / 1e9))
        ffi.check(L.r3g_set_option(b"gemm_waves", 0))
        ffi.check(L.r3g_set_option(b"gemm_raster", -1))
        ref = timeit(lambda: torch.matmul(a, w.t()))
        out.append(dict(op="torch.matmul(hipBLASLt)", M=M, N=N, K=K, ms=ref, tflops=2.0 * M * N * K / ref / 1e9))
    for (B, H, Lq, Lk, shared) in [(2, 16, 4442, 4442, 0), (1, 16, 3072, 3072, 0), (1, 16, 131072, 3072, 1),
                                   (1, 24, 1370, 1370, 0)]:
        lqp, lkp = (Lq + 127) // 128 * 128, (Lk + 63) // 64 * 64
        Q = torch.randn(B, H, lqp, 64, device="cuda").to(torch.bfloat16)
        Kt = torch.randn(1 if shared else B, H, lkp, 64, device="cuda").to(torch.bfloat16)
        Vt = torch.randn(1 if shared else B, H, 64, lkp, device="cuda").to(torch.bfloat16)
        o = torch.zeros(B, Lq, H * 64, device="cuda", dtype=torch.bfloat16)
        for dma in (1,):
            ms = timeit(lambda: ffi.check(L.r3g_op_attention(Q.data_ptr(), Kt.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H,
                                                             Lq, lqp, Lk, lkp, shared, dma, s)), iters=5)
            out.append(dict(op="attn", B=B, H=H, Lq=Lq, Lk=Lk, dma=dma, ms=ms, tflops=4.0 * B * H * Lq * Lk * 64 / ms / 1e9))
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
