"""hy3dgen.texgen.utils.dehighlight_utils -- MI355X mirror of upstream's delighting step (`Light_Shadow_Remover`), the first
thing `Hunyuan3DPaintPipeline.__call__` does with the object's image (behind reference src/2d_to_3d_models/run.py:97).

[UPSTREAM-RECALLED] upstream builds a diffusers StableDiffusionInstructPix2PixPipeline from `config.light_remover_ckpt_path`,
replaces its scheduler by EulerAncestralDiscreteScheduler.from_config, and its __call__ is: resize to 512 x 512; for RGBA erode
the alpha by a 3 x 3 kernel and paint what falls outside white; run the pipeline (prompt "", generator torch.manual_seed(42),
50 steps, image_guidance_scale 1.5, guidance_scale 1.0); match the result's per-channel mean / standard deviation over the
object to the input's (`recorrect_rgb`, kept only if it lowers the mean squared difference to the input); composite over white.
The class and method names, the call signature and those constants are upstream's; neither hy3dgen nor diffusers is in the
container, so none of this is pinned against them.

Here the diffusion model itself runs on the HIP blocks (r3g.delight.InstructPix2Pix: UNet + SD VAE + Euler-ancestral loop);
the image bookkeeping around it (a 512 x 512 erode, four means and standard deviations) is host numpy.  The prompt embedding
of "" is an input (one constant tensor; the CLIP text encoder is not on this path)."""
import json
import os

import numpy as np


def read_weights(folder):
    """the state dict of a diffusers component folder: diffusion_pytorch_model.safetensors, or the older .bin (a torch pickle of
    tensors only)"""
    st = os.path.join(folder, "diffusion_pytorch_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(folder, "diffusion_pytorch_model.bin")
    if os.path.exists(pt):
        import torch
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError("no diffusion_pytorch_model.safetensors / .bin in %s" % folder)


def read_json(path, default=None):
    if not os.path.exists(path):
        if default is None:
            raise FileNotFoundError(path)
        return default
    with open(path) as f:
        return json.load(f)


# what the HIP UNet implements of diffusers' UNet2DConditionModel configuration space: a config.json that says otherwise is
# refused (a silently different architecture would produce plausible-looking garbage)
_UNET_FIXED = {
    "act_fn": "silu", "flip_sin_to_cos": True, "freq_shift": 0, "resnet_time_scale_shift": "default", "mid_block_scale_factor": 1,
    "class_embed_type": None, "addition_embed_type": None, "time_embedding_type": "positional", "dual_cross_attention": False,
    "only_cross_attention": False, "conv_in_kernel": 3, "conv_out_kernel": 3, "encoder_hid_dim": None,
    "mid_block_type": "UNetMidBlock2DCrossAttn", "transformer_layers_per_block": 1, "norm_eps": 1e-5, "downsample_padding": 1,
}


def unet_config_from_diffusers(uc, ctx_tokens):
    """diffusers unet/config.json -> the keys this path uses; refuses layouts it does not run"""
    ch = tuple(uc["block_out_channels"])
    heads = uc.get("attention_head_dim", 8)
    heads = tuple(heads) if isinstance(heads, (list, tuple)) else (heads,) * len(ch)
    if any(c // h_ != 64 for c, h_ in zip(ch, heads)):
        raise ValueError("this path runs head dim 64 (the SD 2.x layout: attention_head_dim = heads per block)")
    if not uc.get("use_linear_projection", False):
        raise ValueError("this path needs use_linear_projection (SD 2.x)")
    n = len(ch)
    want_down = ["CrossAttnDownBlock2D"] * (n - 1) + ["DownBlock2D"]
    want_up = ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * (n - 1)
    if list(uc.get("down_block_types", want_down)) != want_down or list(uc.get("up_block_types", want_up)) != want_up:
        raise ValueError("this path runs the SD layout of blocks (%s / %s)" % (want_down, want_up))
    bad = {k: uc[k] for k, v in _UNET_FIXED.items() if k in uc and uc[k] != v and not (v is None and uc[k] in (None, "None"))}
    if bad:
        raise ValueError("unet/config.json asks for what this path does not implement: %s" % bad)
    return dict(block_out_channels=ch, layers_per_block=uc.get("layers_per_block", 2), cross_attention_dim=uc["cross_attention_dim"],
                ctx_tokens=int(ctx_tokens), temb_dim=4 * ch[0], groups=uc.get("norm_num_groups", 32))


def vae_config_from_diffusers(vc):
    """diffusers vae/config.json (AutoencoderKL) -> the keys this path uses; refuses what the HIP VAE does not implement"""
    n = len(vc["block_out_channels"])
    if list(vc.get("down_block_types", ["DownEncoderBlock2D"] * n)) != ["DownEncoderBlock2D"] * n or \
            list(vc.get("up_block_types", ["UpDecoderBlock2D"] * n)) != ["UpDecoderBlock2D"] * n:
        raise ValueError("this path runs AutoencoderKL's DownEncoderBlock2D / UpDecoderBlock2D layout")
    if vc.get("act_fn", "silu") != "silu" or not vc.get("mid_block_add_attention", True) or vc.get("out_channels", 3) != vc.get("in_channels", 3):
        raise ValueError("vae/config.json asks for what this path does not implement (act_fn silu, mid-block attention, in = out channels)")
    return dict(block_out_channels=tuple(vc["block_out_channels"]), layers_per_block=vc.get("layers_per_block", 2),
                latent_channels=vc.get("latent_channels", 4), image_channels=vc.get("in_channels", 3),
                groups=vc.get("norm_num_groups", 32))


def erode3(alpha):
    """cv2.erode(alpha, np.ones((3, 3), np.uint8), iterations=1): minimum over the 3 x 3 neighbourhood; cv2's default border
    for erosion does not lower the result (the border counts as +infinity)"""
    a = np.asarray(alpha)
    p = np.pad(a, 1, mode="constant", constant_values=np.iinfo(a.dtype).max if a.dtype.kind in "ui" else np.inf)
    out = a.copy()
    for dy in range(3):
        for dx in range(3):
            out = np.minimum(out, p[dy:dy + a.shape[0], dx:dx + a.shape[1]])
    return out


def recorrect_rgb(src, target, alpha, scale=0.95):
    """upstream `recorrect_rgb`: per channel (src - scale mean_src) (std_tgt / std_src) + scale mean_tgt over alpha > 0.5,
    clamped to [0, 1]; kept only if it is closer (mean squared difference over the whole image) to the target than src is.
    src, target float [H, W, 3] in [0, 1], alpha float [H, W, 1] -> float [H, W, 4] (rgb | alpha)"""
    src, target, alpha = np.asarray(src, np.float64), np.asarray(target, np.float64), np.asarray(alpha, np.float64)
    m = alpha[..., 0] > 0.5
    out = np.zeros_like(src)
    for c in range(3):
        s, t = src[..., c][m], target[..., c][m]
        # torch.std is the unbiased estimator
        s_mean, s_std = s.mean(), s.std(ddof=1)
        t_mean, t_std = t.mean(), t.std(ddof=1)
        out[..., c] = np.clip((src[..., c] - scale * s_mean) * (t_std / s_std) + scale * t_mean, 0.0, 1.0)
    keep_src = np.mean((src - target) ** 2) < np.mean((out - target) ** 2)
    return np.concatenate([src if keep_src else out, alpha], axis=-1)


def empty_prompt_embedding(path, prompt=""):
    """The text conditioning of the delighting model.  Upstream's Light_Shadow_Remover calls its InstructPix2Pix pipeline with the
    prompt "" every time, so the CLIP text encoder's output is ONE constant per checkpoint: it is computed once, when the
    checkpoint is loaded, and never on the per-object path.  Sources, in this order:
      * `<path>/prompt_embeds_empty.safetensors` (key "prompt_embeds"), when it is there (tools/make_prompt_embeds.py writes it);
      * the checkpoint's own `tokenizer/` and `text_encoder/` (the stock snapshot layout) through `transformers` on the host,
        exactly as diffusers' `encode_prompt` does: `text_encoder(tokenizer(prompt, padding="max_length",
        max_length=model_max_length, truncation=True).input_ids)[0]` -- the last hidden state, fp32.
    -> f32 [1, tokens, dim]"""
    import torch
    pe = os.path.join(path, "prompt_embeds_empty.safetensors")
    if os.path.exists(pe) and prompt == "":
        from safetensors.torch import load_file
        emb = load_file(pe)["prompt_embeds"]
        return emb.reshape(1, emb.shape[-2], emb.shape[-1]).float()
    tok_dir, enc_dir = os.path.join(path, "tokenizer"), os.path.join(path, "text_encoder")
    if not (os.path.isdir(tok_dir) and os.path.isdir(enc_dir)):
        raise FileNotFoundError("%s has neither prompt_embeds_empty.safetensors nor tokenizer/ + text_encoder/: the delighting model "
                                "needs the text embedding of the empty prompt (INTEGRATION.md)" % path)
    from transformers import CLIPTextModel, CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(tok_dir)
    enc = CLIPTextModel.from_pretrained(enc_dir).to(torch.float32).eval()
    ids = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
    with torch.no_grad():
        emb = enc(ids)[0]
    return emb.reshape(1, emb.shape[-2], emb.shape[-1]).float().contiguous()


class Light_Shadow_Remover:
    cfg_image = 1.5      # image_guidance_scale (inert while cfg_text <= 1: diffusers then runs no guidance at all)
    cfg_text = 1.0       # guidance_scale
    size = 512
    steps = 50
    seed = 42

    def __init__(self, config=None, model=None, prompt_embeds=None):
        """model: an r3g.delight.InstructPix2Pix; prompt_embeds: the text encoder's output for the prompt "" [1, tokens, dim].
        With a config that carries `light_remover_ckpt_path` (upstream's attribute) both are read from that diffusers
        checkpoint directory (from_pretrained)."""
        if model is None:
            path = getattr(config, "light_remover_ckpt_path", None) if config is not None else None
            if not path:
                raise ValueError("Light_Shadow_Remover needs a model or config.light_remover_ckpt_path")
            model, prompt_embeds = self.load(path, device=getattr(config, "device", 0))
        if prompt_embeds is None:
            raise ValueError("Light_Shadow_Remover needs the text embedding of the empty prompt")
        self.model = model
        self.prompt_embeds = prompt_embeds

    @staticmethod
    def load(path, device=0):
        """a diffusers InstructPix2Pix checkpoint directory: unet/ and vae/ (config.json + diffusion_pytorch_model.safetensors
        or .bin), scheduler/scheduler_config.json (prediction_type) and `prompt_embeds_empty.safetensors` (key "prompt_embeds":
        the text encoder's last hidden state for the prompt "", computed once with the checkpoint's own tokenizer / text_encoder
        -- INTEGRATION.md)"""
        from r3g.delight import InstructPix2Pix
        prompt = empty_prompt_embedding(path)
        vc = read_json(os.path.join(path, "vae", "config.json"))
        unet_config = unet_config_from_diffusers(read_json(os.path.join(path, "unet", "config.json")), prompt.shape[1])
        sched_cfg = read_json(os.path.join(path, "scheduler", "scheduler_config.json"), default={})
        model = InstructPix2Pix(read_weights(os.path.join(path, "unet")), read_weights(os.path.join(path, "vae")), unet_config,
                                vae_config_from_diffusers(vc), image_size=Light_Shadow_Remover.size,
                                scaling_factor=vc.get("scaling_factor", 0.18215),
                                prediction_type=sched_cfg.get("prediction_type", "epsilon"), device=device)
        return model, prompt

    def prepare(self, image):
        """PIL image -> (rgb uint8 [S, S, 3] fed to the model, rgb_target float [S, S, 3], alpha float [S, S, 1])"""
        image = image.resize((self.size, self.size))
        arr = np.array(image)
        if image.mode == "RGBA":
            alpha = erode3(arr[:, :, 3])
            arr[alpha == 0, :3] = 255
            arr[:, :, 3] = alpha
            t = arr / 255.0
            return np.ascontiguousarray(arr[:, :, :3]), t[:, :, :3], t[:, :, 3:]
        arr = np.array(image.convert("RGB"))
        t = arr / 255.0
        return arr, t, np.ones_like(t[:, :, :1])

    def finish(self, out_rgb_u8, rgb_target, alpha):
        """the model's uint8 output -> colour-corrected, composited over white -> PIL RGB image"""
        from PIL import Image
        img = recorrect_rgb(out_rgb_u8 / 255.0, rgb_target, alpha)
        img = img[:, :, :3] * img[:, :, 3:] + (1.0 - img[:, :, 3:])
        return Image.fromarray((img * 255).astype(np.uint8))

    def __call__(self, image):
        import torch
        rgb, target, alpha = self.prepare(image)
        x = torch.from_numpy(rgb.astype(np.float32) / 255.0).permute(2, 0, 1)[None] * 2.0 - 1.0       # diffusers VaeImageProcessor
        out = self.model(x, self.prompt_embeds, num_inference_steps=self.steps, generator=torch.Generator().manual_seed(self.seed))
        out = (out[0] / 2 + 0.5).clamp(0, 1).permute(1, 2, 0).float().cpu().numpy()
        return self.finish((out * 255).round().astype(np.uint8), target, alpha)
