"""GPU parity of the SD-family VAE (include/r3g.h r3g_aekl_encode / r3g_aekl_decode) through the C ABI
against the PyTorch-CPU fp32 restatement oracle/aekl_torch.py, on seeded unit-scale synthetic weights that are
bf16-representable on both sides.  Metric: rel-L2 of the output; tolerance 2e-2 -- a pass is ~25 (decode: ~35) GEMM-backed
layers deep with bf16 operands, fp32 accumulation and an fp32 hidden state; the bf16-operand mirror of the same data flow
sits at 0.8-1.0e-2 from the fp32 oracle (tests/test_aekl_cpu.py)."""
import time

import pytest

from parity_support import rel_l2, report

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _round(sd):
    import torch
    return {k: (v.to(torch.bfloat16).float() if v.ndim >= 2 else v.clone()) for k, v in sd.items()}


class Setup:
    def __init__(self, cfg, seed, max_image_hw):
        from oracle import aekl_torch as A
        from r3g import unet as RU
        self.cfg = cfg
        sd = _round(A.build(cfg, seed=seed).state_dict())
        self.oracle = A.AutoencoderKL(cfg).eval()
        self.oracle.load_state_dict(sd, strict=True)
        self.gpu = RU.AutoencoderKLBlocks(sd, block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"],
                                          latent_channels=cfg["latent_channels"], image_channels=cfg["image_channels"],
                                          groups=cfg["groups"], max_image_hw=max_image_hw)


@pytest.fixture(scope="module")
def small():
    from oracle import aekl_torch as A
    return Setup(A.small_config(), 5, 64 * 64)


def _check(name, got, ref, tol=TOL):
    import torch
    got = got.cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    err = rel_l2(got, ref)
    report(name, err, tol)
    assert err <= tol, (name, err)


@pytest.mark.parametrize("h,w", [(32, 32), (64, 32)])
def test_encode_small(small, h, w):
    import torch
    x = torch.randn(1, 3, h, w, generator=torch.Generator().manual_seed(h + w))
    with torch.no_grad():
        ref = small.oracle.encode_moments(x)
    _check("aekl.encode small %dx%d" % (h, w), small.gpu.encode(x), ref)


@pytest.mark.parametrize("h,w", [(8, 8), (16, 8)])
def test_decode_small(small, h, w):
    import torch
    z = torch.randn(1, 4, h, w, generator=torch.Generator().manual_seed(3 * h + w))
    with torch.no_grad():
        ref = small.oracle.decode(z)
    _check("aekl.decode small %dx%d" % (h, w), small.gpu.decode(z), ref)


def test_implicit_convolution_equals_im2col_bit_for_bit(small):
    """the VAE's convolutions -- among them the encoder's stride-2 Downsample2D with its ONE-SIDED padding F.pad(x, (0, 1, 0, 1)) --
    as implicit GEMMs (option conv_implicit, the default) against im2col + GEMM: the same bits"""
    import torch
    from r3g import ffi
    L = ffi.lib()
    x = torch.randn(1, 3, 64, 32, generator=torch.Generator().manual_seed(21))
    z = torch.randn(1, 4, 16, 8, generator=torch.Generator().manual_seed(22))
    got = {}
    try:
        for mode in (1, 0):
            ffi.check(L.r3g_set_option(b"conv_implicit", mode))
            got[mode] = (small.gpu.encode(x).clone(), small.gpu.decode(z).clone())
    finally:
        ffi.check(L.r3g_set_option(b"conv_implicit", 1))
    assert torch.equal(got[1][0], got[0][0]) and torch.equal(got[1][1], got[0][1])


def test_decode_of_encode_round_trip_matches_the_oracles(small):
    """the two passes chained as the pipelines chain them: image -> mode of the latent distribution -> image"""
    import torch
    x = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = small.oracle.decode(small.oracle.encode_mode(x))
    got = small.gpu.decode(small.gpu.encode(x)[:, :4])
    _check("aekl.decode(encode) small", got, ref, tol=2 * TOL)


def test_mid_attention_needs_a_multiple_of_64_latent_pixels(small):
    import torch
    from r3g import ffi
    with pytest.raises(ffi.R3GError):
        small.gpu.decode(torch.zeros(1, 4, 6, 6))          # 36 latent pixels
    with pytest.raises(ValueError):
        small.gpu.encode(torch.zeros(1, 3, 30, 32))        # not divisible by the down-sampling factor
    with pytest.raises(ffi.R3GError):
        small.gpu.encode(torch.zeros(1, 3, 128, 128))      # beyond the arena this instance was created with


def test_sd_dims():
    """stabilityai/stable-diffusion-2-1 vae dims (83.7 M parameters): 128 x 128 image <-> 16 x 16 latent against the oracle, and
    the time of a 512 x 512 decode / encode (64 x 64 latent: the texture pipelines' working size)"""
    import torch
    from oracle import aekl_torch as A
    s = Setup(A.sd_config(), 7, 512 * 512)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 3, 128, 128, generator=g)
    z = torch.randn(1, 4, 16, 16, generator=g)
    with torch.no_grad():
        ref_e, ref_d = s.oracle.encode_moments(x), s.oracle.decode(z)
    _check("aekl.encode sd dims 128x128", s.gpu.encode(x), ref_e)
    _check("aekl.decode sd dims 16x16", s.gpu.decode(z), ref_d)
    zb = torch.randn(1, 4, 64, 64, generator=g)
    xb = torch.randn(1, 3, 512, 512, generator=g)
    for name, fn, arg in (("decode 64x64 -> 512x512", s.gpu.decode, zb), ("encode 512x512 -> 64x64", s.gpu.encode, xb)):
        out = fn(arg)                                         # warm-up (sizes the side buffer)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(arg)
        torch.cuda.synchronize()
        ms = 1000.0 * (time.perf_counter() - t0)
        assert torch.isfinite(out).all()
        report("aekl.%s milliseconds" % name, ms, 1e6)
