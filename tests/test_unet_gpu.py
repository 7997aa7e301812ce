"""GPU parity of the UNet blocks of the texture stage (include/r3g.h "UNet blocks"; SURVEY.md 8f rank 3, first slice) through
the C ABI against the PyTorch-CPU fp32 restatement oracle/unet_torch.py, on seeded unit-scale synthetic weights that are
bf16-representable on both sides.  Metric: rel-L2 of the BRANCH contribution (block(x) - x) where the block has a residual,
of the output otherwise; tolerance 1e-2 (bf16 GEMM operands, fp32 accumulation and residual stream)."""
import numpy as np
import pytest

from parity_support import rel_l2, report

pytestmark = pytest.mark.gpu
TOL = 1e-2


def _round(sd):
    import torch
    return {k: (v.to(torch.bfloat16).float() if v.ndim >= 2 else v.clone()) for k, v in sd.items()}


class Setup:
    def __init__(self, cfg, seed, max_hw):
        from oracle import unet_torch as U
        from r3g import unet as RU
        self.cfg = cfg
        self.sd = _round(U.synthetic_state_dict(cfg, seed=seed))
        self.oracle = U.load(cfg, self.sd)
        self.gpu = RU.UnetBlocks(self.sd, max_hw=max_hw, max_channels=max(cfg["block_out_channels"]), temb_dim=cfg["temb_dim"],
                                 ctx_dim=cfg["cross_attention_dim"], ctx_tokens=cfg["ctx_tokens"], groups=cfg["groups"])

    def inputs(self, c, h, w, seed, tokens=None):
        import torch
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(1, c, h, w, generator=g)
        temb = torch.randn(1, self.cfg["temb_dim"], generator=g)
        ctx = torch.randn(1, tokens or self.cfg["ctx_tokens"], self.cfg["cross_attention_dim"], generator=g).to(torch.bfloat16).float()
        return x, temb, ctx


@pytest.fixture(scope="module")
def small():
    from oracle import unet_torch as U
    return Setup(U.small_config(), 3, 24 * 20)


TOL_MODEL = 2e-2      # a whole UNet forward: ~60 GEMM-backed layers deep


def _check(name, got, ref, base=None, tol=TOL):
    import torch
    got = got.cpu()
    assert torch.isfinite(got).all()
    err = rel_l2(got - base, ref - base) if base is not None else rel_l2(got, ref)
    report(name, err, tol)
    assert err <= tol, (name, err)


@pytest.mark.parametrize("h,w", [(8, 8), (24, 20), (5, 7)])
def test_resnet_transformer_downsample_small(small, h, w):
    import torch
    x, temb, ctx = small.inputs(64, h, w, 10 + h)
    m = small.oracle.down_blocks[0]
    with torch.no_grad():
        _check("unet small %dx%d resnet (branch)" % (h, w), small.gpu.resnet("down_blocks.0.resnets.0", x, temb, 64), m.resnets[0](x, temb), x)
        _check("unet small %dx%d transformer (branch)" % (h, w), small.gpu.transformer("down_blocks.0.attentions.0", x, ctx),
               m.attentions[0](x, ctx), x)
        _check("unet small %dx%d downsample" % (h, w), small.gpu.downsample("down_blocks.0.downsamplers.0", x), m.downsamplers[0](x))


def test_context_shorter_than_the_maximum(small):
    import torch
    x, temb, ctx = small.inputs(64, 8, 8, 77, tokens=5)
    with torch.no_grad():
        _check("unet small transformer, 5 context tokens", small.gpu.transformer("down_blocks.0.attentions.1", x, ctx),
               small.oracle.down_blocks[0].attentions[1](x, ctx), x)


def test_down_block_and_mid_block_small(small):
    import torch
    x, temb, ctx = small.inputs(64, 16, 12, 21)
    with torch.no_grad():
        ref_out, ref_states = small.oracle.down_blocks[0](x, temb, ctx)
    out, states = small.gpu.down_block("down_blocks.0", x, temb, ctx, 64)
    assert len(states) == len(ref_states) == 3
    for i, (a, b) in enumerate(zip(states, ref_states)):
        _check("unet small down block, output state %d" % i, a, b)
    _check("unet small down block output", out, ref_out)
    xm, temb, ctx = small.inputs(128, 6, 6, 22)
    with torch.no_grad():
        ref = small.oracle.mid_block(xm, temb, ctx)
    _check("unet small mid block (branch)", small.gpu.mid_block("mid_block", xm, temb, ctx), ref, xm)
    # a channel-changing resnet (conv_shortcut) does not occur in this slice's blocks; covered separately below


def test_resnet_with_conv_shortcut():
    """ResnetBlock2D with c_in != c_out (diffusers' conv_shortcut, a 1x1 convolution): first resnet of every later down block"""
    import torch
    from oracle import unet_torch as U
    from r3g import unet as RU
    g = torch.Generator().manual_seed(9)
    r = U.ResnetBlock2D(64, 128, 256)
    sd = {}
    for k, v in r.state_dict().items():
        sd["r." + k] = (torch.randn(v.shape, generator=g) / np.sqrt(v[0].numel())) if v.ndim >= 2 else \
            ((1.0 + 0.1 * torch.randn(v.shape, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(v.shape, generator=g))
    sd = _round(sd)
    r.load_state_dict({k[2:]: v for k, v in sd.items()})
    gpu = RU.UnetBlocks(sd, max_hw=100, max_channels=128, temb_dim=256, ctx_dim=128, ctx_tokens=4)
    x, temb = torch.randn(1, 64, 10, 9, generator=g), torch.randn(1, 256, generator=g)
    with torch.no_grad():
        _check("unet resnet 64 -> 128 with conv_shortcut", gpu.resnet("r", x, temb, 128), r(x, temb))


def test_sd21_dims_blocks():
    """Stable-Diffusion-2.1 dims: 320 channels at 64x64 (4096 pixels, 5 heads), context 77 x 1024, time embedding 1280; the mid
    block at 1280 channels, 8x8 (20 heads)"""
    import torch
    from oracle import unet_torch as U
    cfg = U.sd21_config()
    s = Setup(cfg, 7, 64 * 64)
    x, temb, ctx = s.inputs(320, 64, 64, 31)
    m = s.oracle.down_blocks[0]
    with torch.no_grad():
        _check("unet SD-2.1 dims resnet 320 @ 64x64 (branch)", s.gpu.resnet("down_blocks.0.resnets.0", x, temb, 320), m.resnets[0](x, temb), x)
        _check("unet SD-2.1 dims transformer 320 @ 64x64 (branch)", s.gpu.transformer("down_blocks.0.attentions.0", x, ctx),
               m.attentions[0](x, ctx), x)
        _check("unet SD-2.1 dims downsample 320 @ 64x64", s.gpu.downsample("down_blocks.0.downsamplers.0", x), m.downsamplers[0](x))
        xm, temb, ctx = s.inputs(1280, 8, 8, 32)
        _check("unet SD-2.1 dims mid block 1280 @ 8x8 (branch)", s.gpu.mid_block("mid_block", xm, temb, ctx), s.oracle.mid_block(xm, temb, ctx), xm)
    # timing of the SD-2.1-dims down block (2 x (resnet + transformer) + downsample), for the record
    x, temb, ctx = s.inputs(320, 64, 64, 33)
    s.gpu.down_block("down_blocks.0", x, temb, ctx, 320)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        s.gpu.down_block("down_blocks.0", x, temb, ctx, 320)
    torch.cuda.synchronize()
    report("unet SD-2.1 dims down block 320 @ 64x64: milliseconds per call (incl. host re-layout of the input)",
           (time.perf_counter() - t0) / 5 * 1e3, 1e9)


def test_unet_errors():
    from oracle import unet_torch as U
    import torch
    s = Setup(U.small_config(), 3, 64)
    x, temb, ctx = s.inputs(64, 16, 16, 1)          # 256 pixels > max_hw 64
    from r3g import ffi
    with pytest.raises(ffi.R3GError):
        s.gpu.resnet("down_blocks.0.resnets.0", x, temb, 64)
    x, temb, ctx = s.inputs(64, 4, 4, 1)
    with pytest.raises(ffi.R3GError):
        s.gpu.resnet("down_blocks.9.resnets.0", x, temb, 64)      # no such weights


def _full(cfg, seed, h, w):
    import torch
    from oracle import unet_torch as U
    from r3g import unet as RU
    sd = _round(U.synthetic_state_dict(cfg, seed=seed, full=True))
    oracle = U.load(cfg, sd, full=True)
    ch = cfg["block_out_channels"]
    gpu = RU.UnetBlocks(sd, max_hw=h * w, max_channels=2 * max(ch), temb_dim=cfg["temb_dim"], ctx_dim=cfg["cross_attention_dim"],
                        ctx_tokens=cfg["ctx_tokens"], groups=cfg["groups"], block_out_channels=ch,
                        layers_per_block=cfg["layers_per_block"])
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(1, 4, h, w, generator=g)
    ctx = torch.randn(1, cfg["ctx_tokens"], cfg["cross_attention_dim"], generator=g).to(torch.bfloat16).float()
    return oracle, gpu, x, ctx


@pytest.mark.parametrize("h,w,t", [(16, 12, 37.0), (8, 8, 981.0)])
def test_full_unet_forward_small(h, w, t):
    """UNet2DConditionModel.forward end to end (conv_in, time embedding, down path with skips, mid block, up path with
    cat(hidden, skip) and nearest upsampling, conv_norm_out + conv_out) on the two-level CI configuration"""
    import torch
    from oracle import unet_torch as U
    oracle, gpu, x, ctx = _full(U.small_config(), 5, h, w)
    with torch.no_grad():
        ref = oracle(x, t, ctx)
    _check("unet small full forward %dx%d t=%g" % (h, w, t), gpu.forward(x, t, ctx), ref, tol=TOL_MODEL)


@pytest.mark.parametrize("h,w", [(16, 12), (5, 7), (24, 20)])
def test_implicit_convolution_equals_im2col_bit_for_bit(small, h, w):
    """round 5, option conv_implicit: the 3 x 3 convolutions gather their A operand from the activation rows inside the GEMM kernel
    (borders, stride 2, the split over K) instead of reading an im2col matrix -- the same fragments in the same k order, so a
    resnet (two convolutions, stride 1) and a downsampler (stride 2) give the same bits either way"""
    import torch
    from r3g import ffi
    L = ffi.lib()
    x, temb, ctx = small.inputs(64, h, w, 40 + h)
    outs = {}
    try:
        for mode in (1, 0, 1):
            ffi.check(L.r3g_set_option(b"conv_implicit", mode))
            outs[mode] = (small.gpu.resnet("down_blocks.0.resnets.0", x, temb, 64).clone(),
                          small.gpu.downsample("down_blocks.0.downsamplers.0", x).clone())
    finally:
        ffi.check(L.r3g_set_option(b"conv_implicit", 1))
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    # and the whole forward (conv_in with its zero-padded 64 input channels, the up path's convolutions behind the nearest upsampling)
    from oracle import unet_torch as U
    oracle, gpu, xf, cf = _full(U.small_config(), 5, 2 * ((h + 1) // 2), 2 * ((w + 1) // 2))
    try:
        ffi.check(L.r3g_set_option(b"conv_implicit", 0))
        a = gpu.forward(xf, 321.0, cf).clone()
        ffi.check(L.r3g_set_option(b"conv_implicit", 1))
        b = gpu.forward(xf, 321.0, cf).clone()
    finally:
        ffi.check(L.r3g_set_option(b"conv_implicit", 1))
    assert torch.equal(a, b)


def test_full_unet_forward_sd21_dims():
    """the whole SD-2.1-dims UNet (865.9 M parameters, 4 levels, 64x64 latent, 77 x 1024 context), one forward"""
    import time
    import torch
    from oracle import unet_torch as U
    oracle, gpu, x, ctx = _full(U.sd21_config(), 11, 64, 64)
    with torch.no_grad():
        ref = oracle(x, 500.0, ctx)
    out = gpu.forward(x, 500.0, ctx)
    _check("unet SD-2.1 dims full forward 64x64", out, ref, tol=TOL_MODEL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        gpu.forward(x, 500.0, ctx)
    torch.cuda.synchronize()
    report("unet SD-2.1 dims full forward: milliseconds per evaluation", (time.perf_counter() - t0) / 3 * 1e3, 1e9)
