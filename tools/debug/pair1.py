"""debug: guidance pair of ONE view vs the two calls -- where do the halves part?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-re-gen_amd"), os.path.join(ROOT, "tests")]
import torch
from parity_support import rel_l2
from test_unet2p5d_gpu import Pair
from oracle import unet_torch as U
from r3g import unet as RU
small = Pair(U.small_config(), 2, 6, 2, 16 * 16)
gpu = small.gpu
for n_views, hw in ((1, 8), (1, 16), (2, 8)):
    g = torch.Generator().manual_seed(50 + n_views)
    ref = torch.randn(1, 4, hw, hw, generator=g)
    x, nm, ps = (torch.randn(n_views, 4, hw, hw, generator=g) for _ in range(3))
    gpu.reference_pass(ref, [0])
    dev = gpu.gen.device
    rows = RU.to_rows(torch.cat([x, nm, ps], dim=1).to(dev))
    ctx = gpu.text_gen[0].to(dev, torch.bfloat16).contiguous()
    ctx_u = torch.zeros_like(ctx)
    for labels in (None, [v + gpu.max_num_ref_image for v in range(n_views)]):
        for ctxu_name, cu in (("zeros", ctx_u), ("same", ctx)):
            kw = dict(mva_scale=0.8, ref_scale=1.3)
            cond = gpu.gen.forward_mv_rows(rows, n_views, hw, hw, 481.0, ctx, class_labels=labels, flags=2, **kw).clone()
            unc = gpu.gen.forward_mv_rows(rows, n_views, hw, hw, 481.0, cu, class_labels=labels, flags=0, **kw).clone()
            unc_b = gpu.gen.forward_mv_rows(rows, n_views, hw, hw, 481.0, cu, class_labels=labels, flags=0, **kw).clone()
            both = gpu.gen.forward_mv_rows(torch.cat([rows, rows], 0).contiguous(), 2 * n_views, hw, hw, 481.0, torch.cat([ctx, cu], 0).contiguous(),
                                           class_labels=None if labels is None else labels + labels, flags=2 | 4, **kw)
            n = rows.shape[0]
            print("views", n_views, "hw", hw, "labels", labels is not None, "uncond ctx", ctxu_name, "| cond half", "%.3e" % rel_l2(both[:n].cpu(), cond.cpu()),
                  "uncond half", "%.3e" % rel_l2(both[n:].cpu(), unc.cpu()), "| two-call repeat", "%.3e" % rel_l2(unc_b.cpu(), unc.cpu()), flush=True)
# one transformer block alone
name = "down_blocks.0.attentions.1"
g = torch.Generator().manual_seed(4)
for hw in (8, 4, 2, 1):
    x1 = torch.randn(1, 64, hw, hw, generator=g)
    xr = torch.randn(1, 64, hw, hw, generator=g)
    gpu.ref.transformer_mv(name, xr, gpu.text_ref, flags=1)
    gpu.gen.set_condition(name, gpu.ref)
    c = gpu.gen.transformer_mv(name, x1, gpu.text_gen, flags=2, mva_scale=0.7, ref_scale=1.4).cpu()
    u = gpu.gen.transformer_mv(name, x1, gpu.text_gen, flags=0, mva_scale=0.7, ref_scale=1.4).cpu()
    print("transformer hw", hw, "cond vs uncond differ", "%.3e" % rel_l2(c, u))
