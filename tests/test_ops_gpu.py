"""GPU parity of the MFMA kernels (through the C ABI) against a plain PyTorch fp32 reference of the same op.
Stated tolerances (bf16 operands, fp32 accumulate): GEMM rel-L2 <= 5e-3; attention rel-L2 <= 1e-2."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from r3g import ffi
    ffi.context(0)
    return torch, ffi.lib(), ffi


def rel_l2(a, b):
    import torch
    return float(torch.linalg.norm(a.double() - b.double()) / (torch.linalg.norm(b.double()) + 1e-30))


def stream(torch):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def gelu_ref(torch, x, mode):
    import torch.nn.functional as F
    return F.gelu(x, approximate="tanh") if mode == 1 else F.gelu(x)


@pytest.mark.parametrize("dma", [1, 0])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 256), (300, 200, 128), (8960, 1024, 1024),
                                   (1370, 3072, 1024), (77, 64, 5120), (4442, 4096, 1024)])
def test_gemm_bf16_bias(env, dma, M, N, K):
    torch, L, ffi = env
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, None, M, N, K, 0, dma,
                            stream(torch)))
    ref = a.float() @ w.float().t() + bias
    assert torch.isfinite(c.float()).all()
    assert rel_l2(c.float(), ref) <= 5e-3
    # asymmetric operands catch a transposed / permuted store: check a few exact positions too
    idx = torch.randint(0, M, (16,), device="cuda"), torch.randint(0, N, (16,), device="cuda")
    assert torch.allclose(c.float()[idx], ref[idx], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("epi", [1, 2, 3, 4])
def test_gemm_epilogues(env, epi):
    torch, L, ffi = env
    M, N, K = 515, 768, 1024
    g = torch.Generator(device="cuda").manual_seed(epi)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(N, device="cuda", generator=g)
    lin = a.float() @ w.float().t() + bias
    if epi in (1, 2):
        c = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
        ref = gelu_ref(torch, lin, epi)
    elif epi == 3:
        c = torch.randn(M, N, device="cuda", generator=g)
        ref = c + gate * lin
    else:
        c = torch.zeros((M, N), device="cuda")
        ref = lin
    ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N,
                            gate.data_ptr() if epi == 3 else None, M, N, K, epi, 1, stream(torch)))
    assert rel_l2(c.float(), ref) <= 5e-3


@pytest.mark.parametrize("variant", [0, 8, 9, 11, 12])
@pytest.mark.parametrize("M,N,K", [(515, 768, 1024), (1024, 1024, 256), (77, 256, 128)])
def test_gemm_bf16_residual_epilogue(env, variant, M, N, K):
    """EPI_RESID_BF16: c(bf16) = bf16(c + gate * (a w^T + bias)), the sum in fp32, on every tile kernel (wide LDS path, ragged
    edges, persistent grid); the old values must be read before they are overwritten."""
    torch, L, ffi = env
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(N, device="cuda", generator=g)
    c0 = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    c = c0.clone()
    want = (c0.float() + gate * (a.float() @ w.float().t() + bias)).to(torch.bfloat16)
    try:
        ffi.check(L.r3g_set_option(b"gemm_waves", variant))
        ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N, gate.data_ptr(), M, N, K, 6,
                                1, stream(torch)))
    finally:
        ffi.check(L.r3g_set_option(b"gemm_waves", 0))
    assert rel_l2(c.float(), want.float()) <= 4e-3
    # one bf16 step at most, almost everywhere identical (the kernel's k order differs from torch's)
    ulp = (c.float() - want.float()).abs() / want.float().abs().clamp_min(1e-3)
    assert float(ulp.max()) <= 2 ** -6 and float((ulp > 0).float().mean()) < 0.2


@pytest.mark.parametrize("epi", [3, 6, 0])
@pytest.mark.parametrize("M,N,K", [(7552, 1024, 5120), (1000, 512, 512), (256, 256, 2048)])
def test_gemm_split_k_is_correct_and_deterministic(env, epi, M, N, K):
    """gemm_waves = 13: two workgroups per 256x256 tile, each half of K, partial accumulators exchanged through a workspace
    and added in a fixed order -- same bits on every run, and the usual accuracy against fp32"""
    torch, L, ffi = env
    g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(N, device="cuda", generator=g)
    c0 = torch.randn(M, N, device="cuda", generator=g)
    lin = a.float() @ w.float().t() + bias
    if epi == 3:
        init, want = c0, c0 + gate * lin
    elif epi == 6:
        init = c0.to(torch.bfloat16)
        want = init.float() + gate * lin
    else:
        init, want = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), lin
    outs = []
    try:
        ffi.check(L.r3g_set_option(b"gemm_waves", 13))
        for _ in range(3):
            c = init.clone()
            ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N,
                                    gate.data_ptr() if epi in (3, 6) else None, M, N, K, epi, 1, stream(torch)))
            outs.append(c)
    finally:
        ffi.check(L.r3g_set_option(b"gemm_waves", 0))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_l2(outs[0].float(), want) <= 5e-3


@pytest.mark.parametrize("M,N,K,epi,slices", [(384, 1280, 11520, 4, 15), (64, 1280, 23040, 3, 45), (1000, 320, 2880, 4, 5),
                                              (4096, 320, 5760, 3, 5), (1537, 644, 5760, 4, 6), (24576, 320, 2880, 4, 1),
                                              (300, 256, 1024, 3, 1)])
def test_gemm_split_k_of_the_convolutions(env, M, N, K, epi, slices):
    """r3g_op_gemm_splitk, the launch of the texture UNets' 3 x 3 convolutions: few tiles over a deep k run as slices of k into a
    workspace, added in a fixed order.  The slice count is the stated rule's; the result is the unsplit kernel's up to the order of
    fp32 additions, the same bits on every run, and nothing is written past [m][n] of a padded output; a problem the rule leaves
    alone (a full grid, a short k) is the ordinary launch bit for bit."""
    import ctypes
    torch, L, ffi = env
    g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(N, device="cuda", generator=g) if epi == 3 else None
    ldc = N + 4
    c0 = torch.randn(M, ldc, device="cuda", generator=g)
    ws = torch.empty(512 * 128 * 128, device="cuda", dtype=torch.float32)
    plain = c0.clone()
    ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), plain.data_ptr(), ldc,
                            gate.data_ptr() if gate is not None else None, M, N, K, epi, 1, stream(torch)))
    outs, got = [], ctypes.c_int(0)
    for i in range(3):
        ws.fill_(float("nan") if i == 1 else float(i))          # whatever the workspace held
        c = c0.clone()
        ffi.check(L.r3g_op_gemm_splitk(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), ldc,
                                       gate.data_ptr() if gate is not None else None, M, N, K, epi, ws.data_ptr(), ws.numel(),
                                       ctypes.byref(got), stream(torch)))
        outs.append(c)
    assert got.value == slices
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0][:, N:], c0[:, N:])
    if slices == 1:
        assert torch.equal(outs[0], plain)
    else:
        assert not torch.equal(outs[0], plain)                   # it did take another path ...
        assert rel_l2(outs[0][:, :N], plain[:, :N]) <= 2e-6      # ... to the same sums
    lin = a.float() @ w.float().t() + bias
    want = c0[:, :N] + gate * lin if epi == 3 else lin
    assert rel_l2(outs[0][:, :N], want) <= 1e-4
    # the option turns it off
    try:
        ffi.check(L.r3g_set_option(b"gemm_splitk128", 0))
        c = c0.clone()
        ffi.check(L.r3g_op_gemm_splitk(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), ldc,
                                       gate.data_ptr() if gate is not None else None, M, N, K, epi, ws.data_ptr(), ws.numel(),
                                       None, stream(torch)))
    finally:
        ffi.check(L.r3g_set_option(b"gemm_splitk128", 1))
    assert torch.equal(c, plain)


def test_gemm_strided_views(env):
    """column slab of a wider weight (ldw > K) and of wider activations / outputs (lda, ldc > width)"""
    torch, L, ffi = env
    M, N, K = 200, 256, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    a_full = torch.randn(M, 5 * K, device="cuda", generator=g).to(torch.bfloat16)
    w_full = torch.randn(N, 3 * K, device="cuda", generator=g).to(torch.bfloat16)
    c_full = torch.zeros(M, 2 * N, device="cuda", dtype=torch.bfloat16)
    a, w, c = a_full[:, K:2 * K], w_full[:, 2 * K:], c_full[:, N:]
    ffi.check(L.r3g_op_gemm(a.data_ptr(), 5 * K, w.data_ptr(), 3 * K, None, c.data_ptr(), 2 * N, None, M, N, K, 0, 1,
                            stream(torch)))
    assert rel_l2(c.float(), a.float() @ w.float().t()) <= 5e-3
    assert float(c_full[:, :N].abs().max()) == 0.0


def _attn_case(torch, B, H, Lq, Lk, shared, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Bk = 1 if shared else B
    q = torch.randn(B, H, Lq, 64, device="cuda", generator=g)
    k = torch.randn(Bk, H, Lk, 64, device="cuda", generator=g)
    v = torch.randn(Bk, H, Lk, 64, device="cuda", generator=g)
    lqp, lkp = (Lq + 127) // 128 * 128, (Lk + 63) // 64 * 64
    Q = torch.full((B, H, lqp, 64), 3.0, device="cuda", dtype=torch.bfloat16)
    K = torch.full((Bk, H, lkp, 64), 7.0, device="cuda", dtype=torch.bfloat16)     # junk in the padded keys
    from r3g.layout import make_vt
    Vt = make_vt(v, lkp)          # V^T in the kernel's key order (kernels.h vt_key_pos); padded keys zero
    Q[:, :, :Lq] = q.to(torch.bfloat16)
    K[:, :, :Lk] = k.to(torch.bfloat16)
    ref = torch.nn.functional.scaled_dot_product_attention(Q[:, :, :Lq].float(), K[:, :, :Lk].float().expand(B, -1, -1, -1),
                                                           v.to(torch.bfloat16).float().expand(B, -1, -1, -1))
    ref = ref.permute(0, 2, 1, 3).reshape(B, Lq, H * 64)
    return Q, K, Vt, ref, lqp, lkp


@pytest.mark.parametrize("dma", [1, 0])
@pytest.mark.parametrize("B,H,Lq,Lk,shared", [(1, 1, 128, 64, 0), (1, 2, 200, 200, 0), (2, 16, 4442, 4442, 0),
                                              (1, 3, 26, 26, 0), (3, 4, 500, 3072, 1), (1, 24, 1370, 1370, 0)])
def test_attention(env, dma, B, H, Lq, Lk, shared):
    torch, L, ffi = env
    Q, K, Vt, ref, lqp, lkp = _attn_case(torch, B, H, Lq, Lk, shared, Lq + Lk)
    o = torch.zeros(B, Lq, H * 64, device="cuda", dtype=torch.bfloat16)
    ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp, shared,
                                 dma, stream(torch)))
    assert torch.isfinite(o.float()).all()
    assert rel_l2(o.float(), ref) <= 1e-2


@pytest.mark.parametrize("B,H,Lq,Lk,shared", [(1, 2, 200, 200, 0), (2, 16, 4442, 4442, 0), (3, 4, 500, 3072, 1),
                                              (1, 3, 26, 26, 0), (1, 2, 1000, 64, 0), (1, 2, 300, 129, 0), (1, 1, 700, 513, 0)])
def test_attention_64_query_waves_match_the_default_kernel(env, B, H, Lq, Lk, shared):
    """attn_generation 6 (64 queries per wave, K / V^T fragments shared by two 32-query blocks) makes every decision per
    aligned group of 32 queries like the default kernel: same bits, and the same tolerance against the fp32 reference"""
    torch, L, ffi = env
    Q, K, Vt, ref, lqp, lkp = _attn_case(torch, B, H, Lq, Lk, shared, Lq + 3 * Lk)
    outs = []
    try:
        # generation 9 (round 5): the phased 12-wave kernel (three groups one phase apart) -- the same arithmetic in
        # the same order per 32-query block, so the same bits
        for gen in (2, 6, 9):
            ffi.check(L.r3g_set_option(b"attn_generation", gen))
            o = torch.zeros(B, Lq, H * 64, device="cuda", dtype=torch.bfloat16)
            ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp,
                                         shared, 1, stream(torch)))
            torch.cuda.synchronize()
            outs.append(o)
    finally:
        ffi.check(L.r3g_set_option(b"attn_generation", 7))
    assert rel_l2(outs[1].float(), ref) <= 1e-2
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("gen", [2, 6, 9])
def test_attention_ignores_stale_rows_past_lq(env, gen):
    """The query rows between Lq and the padded length hold whatever an earlier launch left there.  They are computed and
    dropped; they must not steer the wave-uniform re-stabilise branch either, or the rounding of the valid queries of
    the same wave would depend on stale data (seen as a first-run / later-run difference in the pipeline)."""
    torch, L, ffi = env
    B, H, Lq, Lk = 1, 2, 200, 300
    Q, K, Vt, ref, lqp, lkp = _attn_case(torch, B, H, Lq, Lk, 0, 5)
    outs = []
    ffi.check(L.r3g_set_option(b"attn_generation", gen))
    try:
        for junk in (0.0, 40.0, -40.0):
            Q[:, :, Lq:] = junk
            o = torch.zeros(B, Lq, H * 64, device="cuda", dtype=torch.bfloat16)
            ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp, 0, 1,
                                         stream(torch)))
            torch.cuda.synchronize()
            outs.append(o)
    finally:
        ffi.check(L.r3g_set_option(b"attn_generation", 7))
    assert rel_l2(outs[0].float(), ref) <= 1e-2
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("gen", [2, 6, 9])
def test_attention_forced_rescale(env, gen):
    """One key row spiked against one query so the running max jumps late in the sequence
    (exercises the online-softmax rescale branch with a large factor)."""
    torch, L, ffi = env
    B, H, Lq, Lk = 1, 1, 128, 512
    Q, K, Vt, _, lqp, lkp = _attn_case(torch, B, H, Lq, Lk, 0, 11)
    from r3g.layout import read_vt
    K[0, 0, 400] = (Q[0, 0, 5].float() * 4).to(torch.bfloat16)
    vrows = read_vt(Vt, Lk).float()
    ref = torch.nn.functional.scaled_dot_product_attention(Q[:, :, :Lq].float(), K[:, :, :Lk].float(), vrows)
    ref = ref.permute(0, 2, 1, 3).reshape(B, Lq, 64)
    o = torch.zeros(B, Lq, 64, device="cuda", dtype=torch.bfloat16)
    ffi.check(L.r3g_set_option(b"attn_generation", gen))
    try:
        ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp, 0, 1,
                                     stream(torch)))
        torch.cuda.synchronize()
    finally:
        ffi.check(L.r3g_set_option(b"attn_generation", 7))
    assert rel_l2(o.float(), ref) <= 1e-2
    assert torch.allclose(o.float()[0, 5], vrows[0, 0, 400], atol=3e-2)


def _attn_run(torch, L, ffi, Q, K, Vt, B, H, Lq, lqp, Lk, lkp, shared, gen, variant):
    o = torch.zeros(B, Lq, H * 64, device="cuda", dtype=torch.bfloat16)
    ffi.check(L.r3g_set_option(b"attn_generation", gen))
    ffi.check(L.r3g_set_option(b"attn_variant", variant))
    try:
        ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp, shared, 1,
                                     stream(torch)))
        torch.cuda.synchronize()
    finally:
        ffi.check(L.r3g_set_option(b"attn_generation", 7))
        ffi.check(L.r3g_set_option(b"attn_variant", ATTN_VARIANT_DEFAULT))
    return o


ATTN_VARIANT_DEFAULT = 1    # the library's default (csrc/attn.hip g_attn_variant); the tests restore it


@pytest.mark.parametrize("gen", [2, 6])
@pytest.mark.parametrize("variant", [1])
def test_attention_variants_without_a_maximum_in_the_common_path(env, gen, variant):
    """Round 6, option attn_variant 1 (default): the fast pass takes no maximum after the first key block and tests the SUM of a lane's
    exponentials against 2^16 into a sticky flag; a workgroup whose valid queries set it runs its tile again with variant 0's body.
    (a) random data: no re-stabilisation either way, so the same
    bits as variant 0; (b) a key 46 log2 units above everything late in the sequence: the safe pass runs, the same bits again;
    (c) a key ~12 units above: variant 0 moves its stabiliser, the fast pass does not -- equal within the bf16 rounding of P;
    (d) stale query rows past Lq with huge values set no flag and change no bit."""
    torch, L, ffi = env
    from r3g.layout import read_vt
    for (B, H, Lq, Lk, shared) in [(1, 2, 200, 200, 0), (2, 16, 4442, 4442, 0), (3, 4, 500, 3072, 1), (1, 2, 300, 129, 0)]:
        Q, K, Vt, ref, lqp, lkp = _attn_case(torch, B, H, Lq, Lk, shared, Lq + 5 * Lk)
        a = _attn_run(torch, L, ffi, Q, K, Vt, B, H, Lq, lqp, Lk, lkp, shared, gen, 0)
        b = _attn_run(torch, L, ffi, Q, K, Vt, B, H, Lq, lqp, Lk, lkp, shared, gen, variant)
        assert rel_l2(b.float(), ref) <= 1e-2
        assert torch.equal(a, b), "random data, %s: max |d| %.3e" % ((B, H, Lq, Lk), float((a.float() - b.float()).abs().max()))
    B, H, Lq, Lk = 1, 1, 128, 512
    for factor, same_bits in ((4.0, True), (1.04, False)):
        Q, K, Vt, _, lqp, lkp = _attn_case(torch, B, H, Lq, Lk, 0, 11)
        K[0, 0, 400] = (Q[0, 0, 5].float() * factor).to(torch.bfloat16)
        vrows = read_vt(Vt, Lk).float()
        ref = torch.nn.functional.scaled_dot_product_attention(Q[:, :, :Lq].float(), K[:, :, :Lk].float(), vrows)
        ref = ref.permute(0, 2, 1, 3).reshape(B, Lq, 64)
        a = _attn_run(torch, L, ffi, Q, K, Vt, B, H, Lq, lqp, Lk, lkp, 0, gen, 0)
        b = _attn_run(torch, L, ffi, Q, K, Vt, B, H, Lq, lqp, Lk, lkp, 0, gen, variant)
        assert torch.isfinite(b.float()).all()
        assert rel_l2(b.float(), ref) <= 1e-2 and rel_l2(a.float(), ref) <= 1e-2
        if same_bits:
            assert torch.equal(a, b)
            assert torch.allclose(b.float()[0, 5], vrows[0, 0, 400], atol=3e-2)
    B, H, Lq, Lk = 1, 2, 200, 300
    Q, K, Vt, ref, lqp, lkp = _attn_case(torch, B, H, Lq, Lk, 0, 5)
    outs = []
    for junk in (0.0, 40.0, -40.0):
        Q[:, :, Lq:] = junk
        outs.append(_attn_run(torch, L, ffi, Q, K, Vt, B, H, Lq, lqp, Lk, lkp, 0, gen, variant))
    assert rel_l2(outs[0].float(), ref) <= 1e-2
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0], _attn_run(torch, L, ffi, Q, K, Vt, B, H, Lq, lqp, Lk, lkp, 0, gen, 0))


@pytest.mark.parametrize("epi,flavour,other", [(1, "tanh", "erf"), (2, "erf", "tanh")])
def test_gelu_flavour_is_the_stated_one(env, epi, flavour, other):
    """tanh- and erf-GELU differ by <= 4.7e-4: less than the bf16 step of most outputs, so no rel-L2 can tell them apart
    (tests/test_mutation_cpu.py).  With W = identity the epilogue sees x exactly; the binned MEAN error against the stated
    flavour must vanish (rounding noise averages out) while the other flavour is >= 5x the tolerance away."""
    torch, L, ffi = env
    from parity_support import MARGIN, TOL_GELU_STAT, gelu_flavour_statistic
    # (the fp32 forms: the packed-fp16 form of round 5 has a systematic error of a few 1e-4 in this tail -- the size of the
    # difference between the two flavours -- and is pinned by test_gelu_packed_fp16_form_equals_its_emulation below)
    ffi.check(L.r3g_set_option(b"gelu_pk", 0))
    try:
        _gelu_flavour_case(torch, L, ffi, epi, flavour, other, MARGIN, TOL_GELU_STAT, gelu_flavour_statistic)
    finally:
        ffi.check(L.r3g_set_option(b"gelu_pk", 1))


def _gelu_flavour_case(torch, L, ffi, epi, flavour, other, MARGIN, TOL_GELU_STAT, gelu_flavour_statistic):
    M, N = 65536, 64
    g = torch.Generator(device="cuda").manual_seed(epi)
    x = (torch.rand(M, N, device="cuda", generator=g) * 2.0 - 3.75).to(torch.bfloat16)     # x in [-3.75, -1.75)
    w = torch.eye(N, device="cuda").to(torch.bfloat16)
    c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ffi.check(L.r3g_op_gemm(x.data_ptr(), N, w.data_ptr(), N, None, c.data_ptr(), N, None, M, N, N, epi, 1, stream(torch)))
    y, xf = c.float().cpu(), x.float().cpu()
    assert gelu_flavour_statistic(y, xf, flavour) <= TOL_GELU_STAT
    assert gelu_flavour_statistic(y, xf, other) >= MARGIN * TOL_GELU_STAT


@pytest.mark.parametrize("epi", [1, 2])
def test_gelu_packed_fp16_form_equals_its_emulation(env, epi):
    """Round 5: the GELU epilogues evaluate x S(x) with S in packed fp16 (csrc/gemm_common.h).  With W = identity the epilogue
    sees x exactly, so the stored bf16 values must equal the instruction-exact numpy emulation of the sequence
    (tools/fit_gelu_pk.py: fp16 round-to-nearest per instruction, fused multiply-adds rounded once, the clamps, x * S in fp32)
    BIT FOR BIT -- that pins the inline-asm operand selection (low / high halves), the coefficients and the clamp modifiers.
    And the form's accuracy against the stated flavour: rel-L2 of the stored values within 3 % of an exactly rounded GELU's,
    |error of S| <= 1e-3, exact saturation beyond |x| = 4 (infinities included), NaN stays NaN."""
    import os
    import sys
    torch, L, ffi = env
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fit_gelu_pk as G
    M, N = 16384, 64
    g = torch.Generator(device="cuda").manual_seed(10 + epi)
    x = (torch.randn(M, N, device="cuda", generator=g) * 2.0).to(torch.bfloat16)
    # (no infinities / NaN here: through the identity GEMM an infinite value turns its whole row into inf * 0 = NaN; the
    # emulation's handling of them is checked on the CPU, tests/test_gelu_pk_cpu.py)
    x[0, :6] = torch.tensor([4.0, -4.0, 1e4, -1e4, 0.0, -0.0], device="cuda").to(torch.bfloat16)
    w = torch.eye(N, device="cuda").to(torch.bfloat16)
    c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ffi.check(L.r3g_op_gemm(x.data_ptr(), N, w.data_ptr(), N, None, c.data_ptr(), N, None, M, N, N, epi, 1, stream(torch)))
    xf = x.float().cpu().numpy()
    got = c.float().cpu().numpy()
    want = G.emulate(xf, erf=(epi == 2))
    nan = np.isnan(xf) | np.isnan(want)        # (x = -inf: -inf * 0 = NaN, as in the reference's own x * Phi(x))
    assert np.isnan(got[nan]).all() and np.isnan(want[nan]).all()
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))
    assert got[0, 0] == 4.0 and got[0, 1] == 0.0 and got[0, 2] == xf[0, 2] and got[0, 3] == 0.0 and got[0, 4] == 0.0
    ok = np.isfinite(xf)
    S = G.S_erf if epi == 2 else G.S_tanh
    ref = xf[ok].astype(np.float64) * S(xf[ok].astype(np.float64))
    exact = G.bf16_rne(ref.astype(np.float32)).astype(np.float64)
    rl = lambda a: float(np.sqrt(((a - ref) ** 2).sum() / (ref ** 2).sum()))
    assert rl(got[ok].astype(np.float64)) <= 1.03 * rl(exact)
    assert np.abs(G.emulate_s(xf[ok], epi == 2).astype(np.float64) - S(xf[ok].astype(np.float64))).max() <= 1e-3
    # the fp32 forms stay selectable and differ from the packed one by less than a bf16 step almost everywhere
    ffi.check(L.r3g_set_option(b"gelu_pk", 0))
    try:
        c32 = torch.zeros_like(c)
        ffi.check(L.r3g_op_gemm(x.data_ptr(), N, w.data_ptr(), N, None, c32.data_ptr(), N, None, M, N, N, epi, 1, stream(torch)))
    finally:
        ffi.check(L.r3g_set_option(b"gelu_pk", 1))
    d = (c32.float() - c.float()).abs().cpu().numpy()[ok]
    assert d.max() <= 2.0 ** -7 * max(1.0, float(np.abs(ref).max())) and rl(c32.float().cpu().numpy()[ok].astype(np.float64)) <= 1.02 * rl(exact)
