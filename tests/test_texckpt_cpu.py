"""Loading the two texture models from checkpoint directories, CPU side: the folder layouts are parsed into the configurations and
state dicts the HIP classes take (those classes are replaced by recorders here -- they need an MI355X), and from_pretrained picks
the two sub-folders up the way upstream's does."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import tex_ckpt_support as CK  # noqa: E402


class _Rec:
    calls = []

    def __init__(self, *a, **k):
        type(self).calls.append((type(self).__name__, a, k))
        self.device = "cpu"
        self.factor, self.latent_channels, self.image_channels = 4, 4, 3
        self.max_num_ref_image = 5


@pytest.fixture()
def patched(monkeypatch, tmp_path):
    import r3g.delight
    import r3g.multiview
    import r3g.unet
    _Rec.calls = []
    for mod, name in ((r3g.delight, "InstructPix2Pix"), (r3g.multiview, "MultiviewUNet"), (r3g.unet, "AutoencoderKLBlocks")):
        monkeypatch.setattr(mod, name, type(name, (_Rec,), {}))
    pieces = CK.write_checkpoints(str(tmp_path))
    return str(tmp_path), pieces


def test_delight_folder_is_parsed(patched):
    from hy3dgen.texgen.utils.dehighlight_utils import Light_Shadow_Remover
    root, (dl, _) = patched

    class Cfg:
        light_remover_ckpt_path = os.path.join(root, "hunyuan3d-delight-v2-0")
        device = 0
    r = Light_Shadow_Remover(Cfg)
    name, a, k = _Rec.calls[-1]
    assert name == "InstructPix2Pix"
    usd, vsd, ucfg, vcfg = a
    assert ucfg == dict(block_out_channels=(64, 128), layers_per_block=2, cross_attention_dim=128, ctx_tokens=13, temb_dim=256, groups=32)
    assert vcfg == dict(block_out_channels=(64, 64, 128), layers_per_block=1, latent_channels=4, image_channels=3, groups=32)
    assert usd["conv_in.weight"].shape[1] == 8 and "decoder.conv_out.bias" in vsd
    assert k["image_size"] == 512 and k["prediction_type"] == "epsilon" and abs(k["scaling_factor"] - 0.18215) < 1e-9
    assert torch.equal(r.prompt_embeds, dl["prompt_embeds"])
    os.remove(os.path.join(Cfg.light_remover_ckpt_path, "prompt_embeds_empty.safetensors"))
    with pytest.raises(FileNotFoundError, match="empty prompt"):
        Light_Shadow_Remover(Cfg)


def test_multiview_folder_is_parsed_from_a_torch_pickle(patched):
    from hy3dgen.texgen.utils.multiview_utils import Multiview_Diffusion_Net
    root, _ = patched

    class Cfg:
        multiview_ckpt_path = os.path.join(root, "hunyuan3d-paint-v2-0")
        device = 0
    net = Multiview_Diffusion_Net(Cfg)
    names = [c[0] for c in _Rec.calls]
    assert names[-2:] == ["AutoencoderKLBlocks", "MultiviewUNet"]
    _, a, k = _Rec.calls[-1]
    sd, ucfg = a
    assert "unet.learned_text_clip_gen" in sd and "unet_dual.conv_in.weight" in sd and ucfg["ctx_tokens"] == 13
    assert k["n_views_max"] == 6 and k["latent_hw"] == (512 // 4) ** 2           # the small VAE has three levels: factor 4
    assert net.pipeline.scheduler.timestep_spacing == "trailing" and net.pipeline.scaling_factor == 0.18215
    os.remove(os.path.join(Cfg.multiview_ckpt_path, "unet", "diffusion_pytorch_model.bin"))
    with pytest.raises(FileNotFoundError):
        Multiview_Diffusion_Net(Cfg)


def test_from_pretrained_picks_up_the_two_subfolders(patched):
    from hy3dgen.texgen import Hunyuan3DPaintPipeline
    from hy3dgen.texgen.utils.dehighlight_utils import Light_Shadow_Remover
    from hy3dgen.texgen.utils.multiview_utils import Multiview_Diffusion_Net
    root, _ = patched
    p = Hunyuan3DPaintPipeline.from_pretrained(root, device="cuda:0")
    assert isinstance(p.delight_model, Light_Shadow_Remover) and isinstance(p.multiview_model, Multiview_Diffusion_Net)
    assert p.source.startswith("delighted input; multiview diffusion model")
    mine = object()
    q = Hunyuan3DPaintPipeline.from_pretrained(root, multiview_model=mine)      # an explicit model wins
    assert q.multiview_model is mine and isinstance(q.delight_model, Light_Shadow_Remover)
    plain = Hunyuan3DPaintPipeline.from_pretrained("synthetic:mini")           # no such folder: the stage's default
    assert plain.delight_model is None and plain.multiview_model is None and plain.source.startswith("input view only")


def test_unsupported_unet_layouts_are_refused():
    from hy3dgen.texgen.utils.dehighlight_utils import unet_config_from_diffusers
    base = {"block_out_channels": [320, 640], "attention_head_dim": [5, 10], "use_linear_projection": True, "cross_attention_dim": 1024}
    assert unet_config_from_diffusers(base, 77)["temb_dim"] == 1280
    with pytest.raises(ValueError, match="head dim 64"):
        unet_config_from_diffusers(dict(base, attention_head_dim=8), 77)       # SD 1.x: 8 heads of 40 / 80
    with pytest.raises(ValueError, match="use_linear_projection"):
        unet_config_from_diffusers(dict(base, use_linear_projection=False), 77)
    with pytest.raises(ValueError, match="act_fn"):
        unet_config_from_diffusers(dict(base, act_fn="gelu"), 77)
    with pytest.raises(ValueError, match="layout of blocks"):
        unet_config_from_diffusers(dict(base, down_block_types=["DownBlock2D", "DownBlock2D"]), 77)
    # the real SD-2.1 unet/config.json passes
    sd21 = {"act_fn": "silu", "attention_head_dim": [5, 10, 20, 20], "block_out_channels": [320, 640, 1280, 1280], "center_input_sample": False,
            "cross_attention_dim": 1024, "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], "downsample_padding": 1,
            "dual_cross_attention": False, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "layers_per_block": 2,
            "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32, "num_class_embeds": None, "only_cross_attention": False,
            "out_channels": 4, "sample_size": 96, "up_block_types": ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3, "upcast_attention": True,
            "use_linear_projection": True}
    assert unet_config_from_diffusers(sd21, 77) == dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                                                        cross_attention_dim=1024, ctx_tokens=77, temb_dim=1280, groups=32)


def test_prompt_embedding_tool_writes_what_the_loader_reads(tmp_path):
    """tools/make_prompt_embeds.py on a tiny random CLIP text encoder saved the way diffusers pipelines store one"""
    import importlib.util
    import json
    from safetensors.torch import load_file
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "tok_src"))
    vocab = {"<|startoftext|>": 0, "<|endoftext|>": 1, "!</w>": 2, "a</w>": 3}
    with open(os.path.join(d, "tok_src", "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(d, "tok_src", "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    tok = CLIPTokenizer(os.path.join(d, "tok_src", "vocab.json"), os.path.join(d, "tok_src", "merges.txt"), pad_token="!", model_max_length=77)
    tok.save_pretrained(os.path.join(d, "tokenizer"))
    cfg = CLIPTextConfig(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                         max_position_embeddings=77, bos_token_id=0, eos_token_id=1, pad_token_id=2)
    torch.manual_seed(0)
    enc = CLIPTextModel(cfg).eval()
    enc.save_pretrained(os.path.join(d, "text_encoder"))
    spec = importlib.util.spec_from_file_location("make_prompt_embeds", os.path.join(ROOT, "tools", "make_prompt_embeds.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    assert tool.main([d]) == 0
    got = load_file(os.path.join(d, "prompt_embeds_empty.safetensors"))["prompt_embeds"]
    ids = tok("", padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert ids[0, 0] == 0 and ids[0, 1] == 1 and ids.shape == (1, 77)          # <start>, <end>, padding
    with torch.no_grad():
        want = enc(ids)[0]
    assert got.shape == (1, 77, 64) and torch.allclose(got, want, atol=1e-6)
