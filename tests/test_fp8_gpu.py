"""FP8 (OCP e4m3) operand path of BASELINE.json configs[3]: the row-scaled quantiser and the fp8 form of the phased GEMM,
against fp32 arithmetic on exactly the quantised values (tight) and on the original bf16 operands (what e4m3 costs)."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.gpu


def _e4m3_to_float(torch, q):
    """decode OCP e4m3fn bytes (no infinities, NaN = 0x7f / 0xff) -- independent of torch's float8 support"""
    b = q.to(torch.int32)
    sign = torch.where((b & 0x80) != 0, -1.0, 1.0)
    e = (b >> 3) & 0xF
    m = (b & 0x7).float()
    val = torch.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * torch.pow(torch.tensor(2.0, device=q.device), (e - 7).float()))
    return sign * val


def _quant(torch, L, ffi, x):
    rows, K = x.shape
    q = torch.empty((rows, K), dtype=torch.uint8, device="cuda")
    sc = torch.empty(rows, dtype=torch.float32, device="cuda")
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ffi.check(L.r3g_op_quant_fp8(x.data_ptr(), K, rows, K, q.data_ptr(), K, sc.data_ptr(), s))
    return q, sc


@pytest.fixture(scope="module")
def env():
    import torch
    from r3g import ffi
    ffi.context(0)
    return torch, ffi.lib(), ffi


def test_quantiser_rounds_to_nearest_e4m3(env):
    torch, L, ffi = env
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn(37, 1024, device="cuda", generator=g) * torch.logspace(-3, 2, 37, device="cuda")[:, None]).to(torch.bfloat16)
    x[5] = 0
    q, sc = _quant(torch, L, ffi, x)
    amax = x.float().abs().amax(dim=1)
    want_sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(sc, want_sc, rtol=1e-6)
    deq = _e4m3_to_float(torch, q) * sc[:, None]
    assert torch.isfinite(deq).all() and float(deq[5].abs().max()) == 0.0
    # nearest representable value: the error is at most half an e4m3 step of the scaled value (2^-4 relative, 2^-10 * scale absolute)
    y = x.float() / sc[:, None]
    err = (_e4m3_to_float(torch, q) - y).abs()
    bound = torch.maximum(y.abs() * 2.0 ** -4, torch.full_like(y, 2.0 ** -10))
    assert bool((err <= bound * 1.0001).all())
    assert float(_e4m3_to_float(torch, q).abs().max()) == 448.0


@pytest.mark.parametrize("epi", [0, 2, 3, 6])
@pytest.mark.parametrize("M,N,K", [(515, 768, 1024), (256, 256, 256), (1000, 1024, 4096)])
def test_fp8_gemm(env, epi, M, N, K):
    torch, L, ffi = env
    g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(N, device="cuda", generator=g)
    a8, sa = _quant(torch, L, ffi, a)
    w8, sw = _quant(torch, L, ffi, w)
    lin_q = (_e4m3_to_float(torch, a8) * sa[:, None]) @ (_e4m3_to_float(torch, w8) * sw[:, None]).t() + bias
    lin = a.float() @ w.float().t() + bias
    c0 = torch.randn(M, N, device="cuda", generator=g)
    if epi == 0:
        c, want_q, want = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), lin_q, lin
    elif epi == 2:
        c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        want_q, want = torch.nn.functional.gelu(lin_q), torch.nn.functional.gelu(lin)
    elif epi == 3:
        c, want_q, want = c0.clone(), c0 + gate * lin_q, c0 + gate * lin
    else:
        c = c0.to(torch.bfloat16)
        want_q, want = c.float() + gate * lin_q, c.float() + gate * lin
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ffi.check(L.r3g_op_gemm_fp8(a8.data_ptr(), K, sa.data_ptr(), w8.data_ptr(), K, sw.data_ptr(), bias.data_ptr(), c.data_ptr(), N,
                                gate.data_ptr() if epi in (3, 6) else None, M, N, K, epi, s))
    rel = lambda x, y: float(torch.linalg.norm(x.float() - y) / torch.linalg.norm(y))   # noqa: E731
    assert rel(c, want_q) <= 4e-3       # the kernel's arithmetic on the quantised operands (bf16 output rounding)
    assert rel(c, want) <= 6e-2         # what two e4m3 operands cost against the bf16 operands


def test_fp8_gemm_throughput(env):
    """131072 x 4096 x 1024 (the geo decoder's MLP-in shape): fp8 against the bf16 phased kernel, reported"""
    torch, L, ffi = env
    from parity_support import report
    M, N, K = 131072, 4096, 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    a8, sa = _quant(torch, L, ffi, a)
    w8, sw = _quant(torch, L, ffi, w)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
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
    us8 = t(lambda: ffi.check(L.r3g_op_gemm_fp8(a8.data_ptr(), K, sa.data_ptr(), w8.data_ptr(), K, sw.data_ptr(), bias.data_ptr(),
                                                  c.data_ptr(), N, None, M, N, K, 0, s)))
    us16 = t(lambda: ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, None, M, N, K, 0, 1, s)))
    usq = t(lambda: _quant(torch, L, ffi, a))
    fl = 2.0 * M * N * K
    report("fp8 GEMM 131072x4096x1024: TFLOP/s", fl / us8 / 1e6, 5000.0)
    report("bf16 GEMM same shape: TFLOP/s", fl / us16 / 1e6, 2500.0)
    report("quantising the 131072x1024 activations: microseconds", usq, 1e6)
    assert us8 > 0
