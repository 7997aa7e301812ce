"""The multiview-UNet oracle (oracle/unet2p5d_torch.py): structural properties that hold whatever upstream's exact code is --
the wrapper reduces to the plain UNet when its extra branches are off, views interact only through attn_multiview, the
reference states enter only through attn_refview -- and the host-side state-dict split / re-layouts of r3g.unet."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))

from oracle import unet2p5d_torch as M, unet_torch as U  # noqa: E402
from r3g import unet as runet  # noqa: E402


@pytest.fixture(scope="module")
def model():
    m = M.build(U.small_config(), seed=2)
    g = torch.Generator().manual_seed(0)
    ref = torch.randn(1, 4, 8, 8, generator=g)
    cond = m.reference_pass(ref, torch.tensor([0]))
    x, nm, ps = (torch.randn(3, 4, 8, 8, generator=g) for _ in range(3))
    return m, cond, x, nm, ps


def test_reference_pass_keeps_one_state_per_transformer(model):
    m, cond, *_ = model
    assert sorted(cond) == sorted(runet.transformer_prefixes(2, 2))
    assert cond["down_blocks.0.attentions.0"].shape == (1, 64, 64) and cond["mid_block.attentions.0"].shape == (1, 16, 128)


def test_branches_off_is_the_plain_unet(model):
    m, cond, x, nm, ps = model
    inp = torch.cat([x, nm, ps], dim=1)[:1]
    m.ctl.clear()                                   # no mode, one sample per batch: the wrapper must be transparent
    with torch.no_grad():
        got = m.unet(inp, 300.0, m.unet.learned_text_clip_gen)
        plain = U.UNet2DConditionModel(dict(U.small_config(), in_channels=12, out_channels=4)).eval()
        gen_sd, _, _ = runet.split_2p5d_state_dict(m.state_dict())
        plain.load_state_dict({k: v for k, v in gen_sd.items() if "attn_multiview" not in k and "attn_refview" not in k
                               and not k.startswith("class_embedding")}, strict=True)
        want = plain(inp, 300.0, m.unet.learned_text_clip_gen)
    assert torch.allclose(got, want, atol=1e-5)


def test_views_interact_only_through_multiview_attention(model):
    m, cond, x, nm, ps = model
    cam = torch.tensor([0, 1, 2])
    full = m(x, 500.0, nm, ps, cond, cam)
    alone = torch.cat([m(x[i:i + 1], 500.0, nm[i:i + 1], ps[i:i + 1], cond, cam[i:i + 1]) for i in range(3)])
    off = m(x, 500.0, nm, ps, cond, cam, mva_scale=0.0)
    assert torch.allclose(off, alone, atol=1e-5)                    # mva_scale 0: every view as if it were alone
    assert (full - alone).abs().max() > 1e-3                        # and the branch does something
    perm = torch.tensor([2, 0, 1])                                  # equivariance: permuting the views permutes the outputs
    assert torch.allclose(m(x[perm], 500.0, nm[perm], ps[perm], cond, cam[perm]), full[perm], atol=1e-5)


def test_reference_enters_only_through_reference_attention(model):
    m, cond, x, nm, ps = model
    other = {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(3)) for k, v in cond.items()}
    a = m(x, 500.0, nm, ps, cond, None, ref_scale=0.0)
    b = m(x, 500.0, nm, ps, other, None, ref_scale=0.0)
    assert torch.equal(a, b)
    assert (m(x, 500.0, nm, ps, cond) - m(x, 500.0, nm, ps, other)).abs().max() > 1e-3


def test_camera_embedding_is_added_to_the_time_embedding(model):
    m, cond, x, nm, ps = model
    a = m(x, 500.0, nm, ps, cond, torch.tensor([0, 1, 2]))
    with torch.no_grad():
        m.unet.class_embedding.weight[5:8] += 1.0                   # generated views use rows offset by the 5 reference slots
        b = m(x, 500.0, nm, ps, cond, torch.tensor([0, 1, 2]))
        m.unet.class_embedding.weight[5:8] -= 1.0
    assert (a - b).abs().max() > 1e-3


def test_state_dict_split_and_prepared_layouts(model):
    m, *_ = model
    sd = m.state_dict()
    gen, ref, extra = runet.split_2p5d_state_dict(sd)
    assert set(extra) == {"learned_text_clip_gen", "learned_text_clip_ref"}
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight" in gen            # ".transformer." is gone
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn_multiview.to_q.weight" in gen
    assert not any("attn_multiview" in k or "attn_refview" in k for k in ref) and "conv_in.weight" in ref
    assert gen["conv_in.weight"].shape[1] == 12 and ref["conv_in.weight"].shape[1] == 4
    w = runet.prepare_weights(gen, "cpu")
    b = "down_blocks.0.attentions.0.transformer_blocks.0."
    q, k, v = (gen[b + "attn_multiview.to_%s.weight" % n] for n in "qkv")
    t, code = w[b + "attn_multiview.to_qkv.weight"]
    assert code == 1 and torch.equal(t.float(), torch.cat([q, k, v]).to(torch.bfloat16).float())
    t, _ = w[b + "attn_refview.to_kv.weight"]
    rk, rv = gen[b + "attn_refview.to_k.weight"], gen[b + "attn_refview.to_v.weight"]
    assert t.shape == (128, 64) and torch.equal(t[:64].float(), rk.to(torch.bfloat16).float()) \
        and torch.equal(t[64:].float(), rv.to(torch.bfloat16).float())                            # one head: k rows, then v rows
    assert (b + "attn_refview.to_q.weight") in w and (b + "attn_multiview.to_q.weight") not in w
    t, code = w["class_embedding.weight"]
    assert code == 0 and t.dtype == torch.float32 and t.shape == (49, 256)
    assert not any("learned_text_clip" in k for k in w)
