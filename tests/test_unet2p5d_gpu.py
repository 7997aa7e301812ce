"""GPU parity of the multiview UNet of the texture stage (include/r3g.h "several samples per call and upstream's 2.5D transformer
blocks") through the C ABI against the PyTorch-CPU fp32 restatement oracle/unet2p5d_torch.py: one 2.5D
transformer with its reference-pass counterpart, and whole evaluations (reference pass of the 4-channel copy, then the 12-channel
generator over 3 / 6 views with camera labels and branch scales).  Tolerances as for the plain UNet: 1e-2 on a block's branch,
2e-2 on a whole forward (bf16 GEMM operands, fp32 accumulation and residual stream)."""
import time

import pytest

from parity_support import rel_l2, report

pytestmark = pytest.mark.gpu


class Pair:
    def __init__(self, cfg, seed, n_views_max, n_ref_max, latent_hw):
        import torch
        from oracle import unet2p5d_torch as M
        from r3g.multiview import MultiviewUNet
        self.cfg = cfg
        m = M.build(cfg, seed=seed)
        sd = {k: (v.to(torch.bfloat16).float() if v.ndim >= 2 else v.clone()) for k, v in m.state_dict().items()}
        m.load_state_dict(sd, strict=True)
        self.oracle = m
        self.gpu = MultiviewUNet(sd, cfg, n_views_max=n_views_max, n_ref_max=n_ref_max, latent_hw=latent_hw)


@pytest.fixture(scope="module")
def small():
    from oracle import unet_torch as U
    return Pair(U.small_config(), 2, 6, 2, 16 * 16)


def test_one_2p5d_transformer_with_reference_and_multiview_attention(small):
    import torch
    m, gpu = small.oracle, small.gpu
    name = "down_blocks.0.attentions.1"
    g = torch.Generator().manual_seed(4)
    xr = torch.randn(2, 64, 8, 8, generator=g)                 # two reference images
    x = torch.randn(3, 64, 8, 8, generator=g)                  # three views
    cond = {}
    with torch.no_grad():
        m.ctl_dual.clear(); m.ctl_dual.update(mode="w", num_in_batch=2, condition_embed_dict=cond)
        ref_o = m.unet_dual.down_blocks[0].attentions[1](xr, m.unet.learned_text_clip_ref.expand(2, -1, -1))
        m.ctl.clear(); m.ctl.update(mode="r", num_in_batch=3, condition_embed_dict=cond, mva_scale=0.7, ref_scale=1.4)
        want = m.unet.down_blocks[0].attentions[1](x, m.unet.learned_text_clip_gen.expand(3, -1, -1))
        m.ctl.update(mva_scale=0.0, ref_scale=0.0)
        plain = m.unet.down_blocks[0].attentions[1](x, m.unet.learned_text_clip_gen.expand(3, -1, -1))
    ref_g = gpu.ref.transformer_mv(name, xr, gpu.text_ref, flags=1).cpu()
    e = rel_l2(ref_g - xr, ref_o - xr)
    report("unet2p5d.transformer reference copy (2 samples, plain block)", e, 1e-2)
    assert e < 1e-2
    gpu.gen.set_condition(name, gpu.ref)
    got = gpu.gen.transformer_mv(name, x, gpu.text_gen, flags=2, mva_scale=0.7, ref_scale=1.4).cpu()
    e = rel_l2(got - x, want - x)
    report("unet2p5d.transformer 3 views, refview x1.4 + multiview x0.7", e, 1e-2)
    assert e < 1e-2
    # the two extra branches really contribute: against the oracle with both scales at zero the result is far off
    assert rel_l2(got - x, plain - x) > 0.1
    got0 = gpu.gen.transformer_mv(name, x, gpu.text_gen, flags=0, mva_scale=0.0).cpu()
    e = rel_l2(got0 - x, plain - x)
    report("unet2p5d.transformer 3 views, both branches off", e, 1e-2)
    assert e < 1e-2


@pytest.mark.parametrize("n_views,n_ref,hw,scales", [(3, 1, 8, (1.0, 1.0)), (6, 2, 16, (0.8, 1.3))])
def test_whole_evaluation_small(small, n_views, n_ref, hw, scales):
    import torch
    m, gpu = small.oracle, small.gpu
    g = torch.Generator().manual_seed(10 + n_views)
    ref = torch.randn(n_ref, 4, hw, hw, generator=g)
    x, nm, ps = (torch.randn(n_views, 4, hw, hw, generator=g) for _ in range(3))
    cam_ref, cam = list(range(n_ref)), list(range(n_views))
    with torch.no_grad():
        cond = m.reference_pass(ref, torch.tensor(cam_ref))
        want = m(x, 481.0, nm, ps, cond, torch.tensor(cam), mva_scale=scales[0], ref_scale=scales[1])
        indep = m(x, 481.0, nm, ps, cond, torch.tensor(cam), mva_scale=0.0, ref_scale=0.0)
    gpu.reference_pass(ref, cam_ref)
    got = gpu(x, 481.0, nm, ps, cam, mva_scale=scales[0], ref_scale=scales[1]).cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    e = rel_l2(got, want)
    report("unet2p5d.forward %d views + %d reference, %dx%d latents" % (n_views, n_ref, hw, hw), e, 2e-2)
    assert e < 2e-2
    assert rel_l2(got, indep) > 5 * e                          # the cross-view / reference branches are not a rounding effect


@pytest.mark.parametrize("n_views,hw", [(6, 16), (3, 8), (1, 8)])   # (1, 8): one view per half -- no multiview attention in either form
def test_guidance_pair_in_one_launch_set_equals_the_two_calls(small, n_views, hw):
    """r3g_unet_forward_mv flag 4 (round 5): the conditional evaluation (learned context, reference attention) and the unconditional one
    (zero context, no reference attention) of a guided step as ONE launch set of 2 n samples -- every sample's prediction is the
    two-call one (the same kernels per row; only the split-K convolutions may add their slices in another order)"""
    import torch
    from r3g import unet as RU
    gpu = small.gpu
    g = torch.Generator().manual_seed(50 + n_views)
    ref = torch.randn(1, 4, hw, hw, generator=g)
    x, nm, ps = (torch.randn(n_views, 4, hw, hw, generator=g) for _ in range(3))
    gpu.reference_pass(ref, [0])
    dev = gpu.gen.device
    rows = RU.to_rows(torch.cat([x, nm, ps], dim=1).to(dev))
    ctx = gpu.text_gen[0].to(dev, torch.bfloat16).contiguous()
    ctx_u = torch.zeros_like(ctx)
    labels = [v + gpu.max_num_ref_image for v in range(n_views)]
    kw = dict(mva_scale=0.8, ref_scale=1.3)
    cond = gpu.gen.forward_mv_rows(rows, n_views, hw, hw, 481.0, ctx, class_labels=labels, flags=2, **kw).clone()
    unc = gpu.gen.forward_mv_rows(rows, n_views, hw, hw, 481.0, ctx_u, class_labels=labels, flags=0, **kw).clone()
    both = gpu.gen.forward_mv_rows(torch.cat([rows, rows], dim=0).contiguous(), 2 * n_views, hw, hw, 481.0,
                                   torch.cat([ctx, ctx_u], dim=0).contiguous(), class_labels=labels + labels, flags=2 | 4, **kw)
    n = rows.shape[0]
    assert torch.isfinite(both).all()
    ec, eu = rel_l2(both[:n].cpu(), cond.cpu()), rel_l2(both[n:].cpu(), unc.cpu())
    report("unet2p5d.guidance pair in one launch set vs two calls, %d views: conditional half" % n_views, ec, 1e-5)
    report("unet2p5d.guidance pair in one launch set vs two calls, %d views: unconditional half" % n_views, eu, 1e-5)
    assert ec <= 1e-5 and eu <= 1e-5
    assert rel_l2(cond.cpu(), unc.cpu()) > 1e-2                  # the two evaluations do differ: the halves were not mixed up
    # the pair again gives the same bits (the context's group stride, the padded query rows between the groups)
    again = gpu.gen.forward_mv_rows(torch.cat([rows, rows], dim=0).contiguous(), 2 * n_views, hw, hw, 481.0,
                                    torch.cat([ctx, ctx_u], dim=0).contiguous(), class_labels=labels + labels, flags=2 | 4, **kw)
    assert torch.equal(again, both)


def test_single_view_without_reference_is_the_plain_forward(small):
    import torch
    m, gpu = small.oracle, small.gpu
    g = torch.Generator().manual_seed(31)
    x, nm, ps = (torch.randn(1, 4, 8, 8, generator=g) for _ in range(3))
    with torch.no_grad():
        m.ctl.clear()
        want = m.unet(torch.cat([x, nm, ps], dim=1), 77.0, m.unet.learned_text_clip_gen)
    gpu.has_reference = False
    got = gpu(x, 77.0, nm, ps).cpu()
    e = rel_l2(got, want)
    report("unet2p5d.forward 1 view, no reference, no camera", e, 2e-2)
    assert e < 2e-2


def test_errors(small):
    import torch
    from r3g import ffi
    gpu = small.gpu
    x = torch.zeros(3, 4, 8, 8)
    with pytest.raises(ffi.R3GError):
        gpu.gen.forward_mv(torch.zeros(3, 12, 8, 8), 1.0, gpu.text_gen, class_labels=[0, 1, 999])      # camera index out of range
    with pytest.raises(ValueError):
        gpu.gen.forward_mv(torch.zeros(3, 12, 8, 8), 1.0, gpu.text_gen, flags=4)                       # the pair needs two contexts: rows API
    with pytest.raises(ffi.R3GError):                                                                   # the pair with the reference-pass flag
        gpu.gen.forward_mv_rows(torch.zeros(4 * 64, 12, device=gpu.gen.device), 4, 8, 8, 1.0,
                                torch.zeros(2 * gpu.text_gen.shape[1], gpu.text_gen.shape[2], dtype=torch.bfloat16, device=gpu.gen.device), flags=4 | 1)
    with pytest.raises(ffi.R3GError):
        gpu.gen.forward_mv(torch.zeros(13, 12, 16, 16), 1.0, gpu.text_gen)                             # 13 x 256 rows > the arena's 2 x 6 x 256 (a guidance pair of 6 views)
    with pytest.raises(ffi.R3GError):
        gpu.gen.condition("down_blocks.0.attentions.0")                                                 # the generator never ran with flag 1


def test_sd21_dims_six_views_timing():
    """the real sizes: SD-2.1 UNet (12-channel conv_in, 2.5D blocks) + the
    4-channel reference copy (1.83 G parameters together), 6 views of 32 x 32 latents (256 x 256 views) and, for the time only, 64 x 64 (512 x 512 views)"""
    import torch
    from oracle import unet_torch as U
    p = Pair(U.sd21_config(), 1, 6, 1, 64 * 64)
    g = torch.Generator().manual_seed(5)
    ref = torch.randn(1, 4, 16, 16, generator=g)
    x, nm, ps = (torch.randn(6, 4, 16, 16, generator=g) for _ in range(3))
    with torch.no_grad():
        cond = p.oracle.reference_pass(ref, torch.tensor([0]))
        want = p.oracle(x, 640.0, nm, ps, cond, torch.tensor(range(6)))
    p.gpu.reference_pass(ref, [0])
    got = p.gpu(x, 640.0, nm, ps, list(range(6))).cpu()
    e = rel_l2(got, want)
    report("unet2p5d.forward sd21 dims, 6 views + 1 reference, 16x16 latents", e, 2e-2)
    assert e < 2e-2
    for hw in (32, 64):
        ref = torch.randn(1, 4, hw, hw, generator=g)
        x, nm, ps = (torch.randn(6, 4, hw, hw, generator=g) for _ in range(3))
        p.gpu.reference_pass(ref, [0])
        out = p.gpu(x, 640.0, nm, ps, list(range(6)))
        ts = []
        for _ in range(5):         # one evaluation is a few hundred launches from Python: the median, not one sample
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = p.gpu(x, 640.0, nm, ps, list(range(6)))
            torch.cuda.synchronize()
            ts.append(1000.0 * (time.perf_counter() - t0))
        assert torch.isfinite(out).all()
        report("unet2p5d.sd21 dims: milliseconds per evaluation (median of 5), 6 views of %dx%d latents" % (hw, hw), sorted(ts)[2], 1e6)
