"""Test support: write synthetic checkpoint directories in the layouts the texture models are loaded from -- a diffusers
InstructPix2Pix folder (hunyuan3d-delight-v2-0) and upstream's multiview folder (hunyuan3d-paint-v2-0, [UPSTREAM-RECALLED] layout) --
from the oracle's small random models."""
import json
import os

import torch


def _dump(path, obj):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(obj, f)


def _vae(folder, vcfg, sd):
    from safetensors.torch import save_file
    _dump(os.path.join(folder, "vae", "config.json"),
          {"block_out_channels": list(vcfg["block_out_channels"]), "layers_per_block": vcfg["layers_per_block"],
           "latent_channels": vcfg["latent_channels"], "in_channels": 3, "out_channels": 3, "norm_num_groups": vcfg["groups"],
           "scaling_factor": 0.18215})
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(folder, "vae", "diffusion_pytorch_model.safetensors"))


def _unet_cfg(ucfg, in_channels):
    return {"block_out_channels": list(ucfg["block_out_channels"]), "attention_head_dim": list(ucfg["heads"]),
            "use_linear_projection": True, "cross_attention_dim": ucfg["cross_attention_dim"],
            "layers_per_block": ucfg["layers_per_block"], "norm_num_groups": ucfg["groups"], "in_channels": in_channels,
            "out_channels": 4}


def write_checkpoints(root, seed=0):
    """-> (delight oracle pieces, multiview oracle pieces): dicts with the torch modules / tensors the folders were written from"""
    from safetensors.torch import save_file
    from oracle import aekl_torch as A, unet2p5d_torch as M, unet_torch as U
    rnd = lambda sd: {k: (v.to(torch.bfloat16).float() if v.ndim >= 2 else v.clone()) for k, v in sd.items()}
    vcfg = A.small_config()
    vae = A.build(vcfg, seed=seed + 5)
    vsd = rnd(vae.state_dict())
    vae.load_state_dict(vsd, strict=True)
    # ---- delighting model
    d = os.path.join(root, "hunyuan3d-delight-v2-0")
    ucfg = dict(U.small_config(), in_channels=8, out_channels=4)
    usd = rnd(U.synthetic_state_dict(ucfg, seed=seed + 3, full=True))
    _dump(os.path.join(d, "unet", "config.json"), _unet_cfg(ucfg, 8))
    save_file({k: v.contiguous() for k, v in usd.items()}, os.path.join(d, "unet", "diffusion_pytorch_model.safetensors"))
    _vae(d, vcfg, vsd)
    _dump(os.path.join(d, "scheduler", "scheduler_config.json"), {"prediction_type": "epsilon", "beta_schedule": "scaled_linear"})
    pe = torch.randn(1, ucfg["ctx_tokens"], ucfg["cross_attention_dim"], generator=torch.Generator().manual_seed(seed)).to(torch.bfloat16).float()
    save_file({"prompt_embeds": pe}, os.path.join(d, "prompt_embeds_empty.safetensors"))
    # ---- multiview model: the UNet's weights as the older torch pickle
    m = os.path.join(root, "hunyuan3d-paint-v2-0")
    mcfg = U.small_config()
    mv = M.build(mcfg, seed=seed + 2)
    msd = rnd(mv.state_dict())
    mv.load_state_dict(msd, strict=True)
    _dump(os.path.join(m, "unet", "config.json"), _unet_cfg(mcfg, 4))
    torch.save(msd, os.path.join(m, "unet", "diffusion_pytorch_model.bin"))
    _vae(m, vcfg, vsd)
    _dump(os.path.join(m, "scheduler", "scheduler_config.json"), {"prediction_type": "epsilon", "timestep_spacing": "trailing"})
    return ({"unet": U.load(ucfg, usd, full=True), "vae": vae, "prompt_embeds": pe, "ucfg": ucfg, "vcfg": vcfg},
            {"unet": mv, "vae": vae, "ucfg": mcfg, "vcfg": vcfg})
