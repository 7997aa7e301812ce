"""The AutoencoderKL oracle (oracle/aekl_torch.py) against what is published about the SD VAE (parameter count, key names)
and against torch.nn.functional; and the algebra + weight re-layouts of the HIP path (tests/aekl_rows_mirror.py mirrors
unet.cpp's data flow on rows with the weights exactly as r3g.unet.prepare_aekl_weights hands them over) against the oracle."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import aekl_torch as A  # noqa: E402
from r3g import unet as runet  # noqa: E402
from aekl_rows_mirror import Mirror  # noqa: E402


@pytest.fixture(scope="module")
def small():
    cfg = A.small_config()
    return cfg, A.build(cfg, seed=5)


def test_sd_dims_have_the_published_parameter_count_and_names():
    m = A.build(A.sd_config())
    assert sum(p.numel() for p in m.parameters()) == 83_653_863          # diffusers AutoencoderKL of SD 1.x / 2.x
    assert sum(p.numel() for p in m.decoder.parameters()) == 49_490_179
    keys = set(m.state_dict())
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.resnets.1.conv2.bias", "encoder.down_blocks.2.downsamplers.0.conv.weight",
              "encoder.down_blocks.1.resnets.0.conv_shortcut.weight", "encoder.mid_block.attentions.0.group_norm.weight",
              "encoder.mid_block.attentions.0.to_q.bias", "encoder.mid_block.attentions.0.to_out.0.weight", "encoder.conv_norm_out.bias",
              "quant_conv.weight", "post_quant_conv.bias", "decoder.conv_in.weight", "decoder.mid_block.resnets.1.norm2.weight",
              "decoder.up_blocks.0.resnets.2.conv1.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.bias",
              "decoder.up_blocks.2.upsamplers.0.conv.weight", "decoder.conv_out.bias"):
        assert k in keys, k
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in keys      # the last level keeps its resolution
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in keys
    assert not any("time_emb_proj" in k for k in keys)


def test_mid_attention_is_single_head_scaled_dot_product_attention(small):
    cfg, m = small
    a = m.decoder.mid_block.attentions[0]
    x = torch.randn(1, 128, 8, 8, generator=torch.Generator().manual_seed(1))
    t = a.group_norm(x).view(1, 128, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(a.to_q(t)[:, None], a.to_k(t)[:, None], a.to_v(t)[:, None])[:, 0]      # one head of dim 128
    ref = x + a.to_out[0](ref).transpose(1, 2).reshape(1, 128, 8, 8)
    assert torch.allclose(a(x), ref, atol=1e-5, rtol=1e-5)


def test_downsample_pads_one_pixel_after_not_before(small):
    cfg, m = small
    d = m.encoder.down_blocks[0].downsamplers[0]
    x = torch.randn(1, 64, 8, 8, generator=torch.Generator().manual_seed(2))
    y = d(x)
    assert y.shape == (1, 64, 4, 4)
    # output (0, 0) sees input rows / columns 0..2 (no padding before); output (3, 3) sees 6, 7 and one zero row / column
    w, b = d.conv.weight, d.conv.bias
    assert torch.allclose(y[0, :, 0, 0], (w * x[0, :, 0:3, 0:3][None]).sum((1, 2, 3)) + b, atol=1e-5)
    tail = torch.zeros(64, 3, 3)
    tail[:, :2, :2] = x[0, :, 6:8, 6:8]
    assert torch.allclose(y[0, :, 3, 3], (w * tail[None]).sum((1, 2, 3)) + b, atol=1e-5)


def test_shapes_and_the_mode_of_the_latent_distribution(small):
    cfg, m = small
    x = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    mom = m.encode_moments(x)
    assert mom.shape == (1, 8, 8, 8)
    assert torch.equal(m.encode_mode(x), mom[:, :4])
    assert m.decode(mom[:, :4]).shape == (1, 3, 32, 32)


def test_prepared_weights_have_the_library_layouts(small):
    cfg, m = small
    w = runet.prepare_aekl_weights(m.state_dict(), "cpu")
    sd = m.state_dict()
    t, code = w["decoder.conv_out.weight"]
    assert code == 1 and t.dtype == torch.bfloat16 and t.shape == (4, 9 * 64)         # 3 image channels -> 4 rows, the last zero
    assert torch.equal(t[3].float(), torch.zeros(9 * 64)) and w["decoder.conv_out.bias"][0].shape == (1, 4)
    assert w["decoder.conv_out.bias"][0][0, 3] == 0
    t, _ = w["encoder.conv_in.weight"]
    assert t.shape == (64, 9 * 64)                                                    # 3 input channels zero-padded to 64
    k = t.float().reshape(64, 3, 3, 64)
    assert torch.equal(k[..., 3:], torch.zeros(64, 3, 3, 61))
    assert torch.equal(k[..., :3], sd["encoder.conv_in.weight"].permute(0, 2, 3, 1).to(torch.bfloat16).float())
    t, _ = w["post_quant_conv.weight"]
    assert t.shape == (4, 64) and torch.equal(t[:, 4:].float(), torch.zeros(4, 60))
    assert torch.equal(t[:, :4].float(), sd["post_quant_conv.weight"][:, :, 0, 0].to(torch.bfloat16).float())
    assert w["quant_conv.weight"][0].shape == (8, 64)
    assert w["decoder.mid_block.attentions.0.to_v.weight"][0].shape == (128, 128)
    for name, (t, code) in w.items():
        assert t.ndim == 2 and (code == 0 or t.shape[1] % 64 == 0) and (code == 0 or t.shape[0] % 4 == 0), name


def _rows(x):
    return runet.to_rows(x)


def test_row_mirror_of_the_hip_data_flow_matches_the_oracle(small):
    """fp32 mirror: the one-sided im2col, V^T = W_v xn^T, b_v behind the softmax, K / N zero padding -- all exact algebra"""
    cfg, m = small
    # the library reads bf16 matrices: give the oracle the same (bf16-representable) weights so that only the algebra is compared
    sd = {k: (v.to(torch.bfloat16).float() if v.ndim >= 2 else v) for k, v in m.state_dict().items()}
    m2 = A.AutoencoderKL(cfg).eval()
    m2.load_state_dict(sd, strict=True)
    mir = Mirror(runet.prepare_aekl_weights(sd, "cpu"), cfg, bf16=False)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 3, 32, 32, generator=g)
    mom, h, w = mir.encode(_rows(x), 32, 32)
    ref = m2.encode_moments(x)
    assert (h, w) == (8, 8)
    assert torch.allclose(runet.from_rows(mom[:, :8], 8, 8), ref, atol=2e-4, rtol=1e-4)
    z = torch.randn(1, 4, 8, 8, generator=g)
    img, H, W = mir.decode(_rows(z), 8, 8)
    assert (H, W) == (32, 32) and torch.equal(img[:, 3], torch.zeros(32 * 32))
    assert torch.allclose(runet.from_rows(img[:, :3], 32, 32), m2.decode(z), atol=2e-4, rtol=1e-4)


def test_bf16_operands_stay_within_the_tolerance_the_gpu_test_uses(small):
    """the mirror with every GEMM operand rounded to bf16 (what the MFMA path does) against the fp32 oracle: the relative
    error the GPU parity test (tests/test_aekl_gpu.py, 2e-2) has to expect, with a factor of margin"""
    cfg, m = small
    sd = m.state_dict()
    mir = Mirror(runet.prepare_aekl_weights(sd, "cpu"), cfg, bf16=True)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 3, 32, 32, generator=g)
    z = torch.randn(1, 4, 8, 8, generator=g)
    mom, _, _ = mir.encode(_rows(x), 32, 32)
    img, _, _ = mir.decode(_rows(z), 8, 8)
    e_enc = (runet.from_rows(mom[:, :8], 8, 8) - m.encode_moments(x)).norm() / m.encode_moments(x).norm()
    e_dec = (runet.from_rows(img[:, :3], 32, 32) - m.decode(z)).norm() / m.decode(z).norm()
    assert e_enc < 1e-2 and e_dec < 1e-2, (float(e_enc), float(e_dec))
