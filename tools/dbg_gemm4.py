#!/usr/bin/env python3
"""where the stream kernel (variant 14) differs from the 128x128 kernel (variant 8): coordinates of the mismatching elements"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import torch
from r3g import ffi
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_gemm import make
ffi.context(0)
L = ffi.lib()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(v, a, w, bias, c, M, N, K, epi):
    ffi.check(L.r3g_set_option(b"gemm_waves", v))
    ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr() if bias is not None else None, c.data_ptr(), N, None, M, N, K, epi, 1, s))
for (M, N, K, epi, use_bias) in [(256, 256, 1024, 0, True), (256, 256, 1024, 0, False), (512, 512, 1024, 0, True), (256, 256, 2048, 0, True)]:
    a, w, bias, gate, c0 = make(M, N, K, epi, M + N + K + epi)
    b = bias if use_bias else None
    ref = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    got = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    run(8, a, w, b, ref, M, N, K, epi)
    run(14, a, w, b, got, M, N, K, epi)
    torch.cuda.synchronize()
    bad = (got.float() != ref.float()) | torch.isnan(got.float())
    idx = bad.nonzero().cpu()
    print("shape", M, N, K, "bias", use_bias, "bad", len(idx), "nan", int(torch.isnan(got.float()).sum()))
    if len(idx) == 0:
        continue
    r, c = idx[:, 0], idx[:, 1]
    tiles = {}
    for rr, cc in zip((r // 16).tolist(), (c // 16).tolist()):
        tiles[(rr, cc)] = tiles.get((rr, cc), 0) + 1
    print(" 16x16 tiles hit:", len(tiles), sorted(tiles.items())[:40])
    print(" rows mod 16:", sorted(set((r % 16).tolist())), " cols mod 32:", sorted(set((c % 32).tolist())))
    d = (got.float() - ref.float())[bad]
    print(" diff abs mean %.3f max %.3f" % (float(d.abs().mean()), float(d.abs().max())))
    # is the wrong value what one gets without bias / with another column's bias?
    nb = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    run(8, a, w, None, nb, M, N, K, epi)
    print(" equal to the no-bias result at the bad places:", int((got[bad] == nb[bad]).sum()))
ffi.check(L.r3g_set_option(b"gemm_waves", 0))
