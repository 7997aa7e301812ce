"""Checkpoint layout of the Hunyuan3D-2 shape model (upstream state-dict names and shapes) and
checkpoint sources: safetensors on disk, or seeded synthetic weights generated directly on the GPU
(no network and no real weights are available where this was built; SURVEY.md 8d "synthetic weights").
"""
import os

import torch


def swiglu_hidden(hidden_size, mlp_ratio):
    return (int(int(hidden_size * mlp_ratio) * 2 / 3) + 7) // 8 * 8


def param_shapes(cfg):
    """name -> shape for every floating-point parameter the pipeline loads."""
    d, v, c = cfg["dit"], cfg["vae"], cfg["cond"]
    s = {}

    def lin(name, n, k, bias=True):
        s[name + ".weight"] = (n, k)
        if bias:
            s[name + ".bias"] = (n,)

    H, mh, hd = d["hidden_size"], int(d["hidden_size"] * d["mlp_ratio"]), d["hidden_size"] // d["num_heads"]
    lin("model.latent_in", H, d["in_channels"])
    lin("model.time_in.in_layer", H, 256)
    lin("model.time_in.out_layer", H, H)
    lin("model.cond_in", H, d["context_in_dim"])
    for i in range(d["depth"]):
        for st in ("img", "txt"):
            b = "model.double_blocks.%d.%s" % (i, st)
            lin(b + "_mod.lin", 6 * H, H)
            lin(b + "_attn.qkv", 3 * H, H, d["qkv_bias"])
            s[b + "_attn.norm.query_norm.scale"] = (hd,)
            s[b + "_attn.norm.key_norm.scale"] = (hd,)
            lin(b + "_attn.proj", H, H)
            lin(b + "_mlp.0", mh, H)
            lin(b + "_mlp.2", H, mh)
    for i in range(d["depth_single_blocks"]):
        b = "model.single_blocks.%d" % i
        lin(b + ".linear1", 3 * H + mh, H)
        lin(b + ".linear2", H, H + mh)
        s[b + ".norm.query_norm.scale"] = (hd,)
        s[b + ".norm.key_norm.scale"] = (hd,)
        lin(b + ".modulation.lin", 3 * H, H)
    lin("model.final_layer.linear", d["in_channels"], H)
    lin("model.final_layer.adaLN_modulation.1", 2 * H, H)

    W, whd = v["width"], v["width"] // v["heads"]
    lin("vae.post_kl", W, v["embed_dim"])

    def qknorm(base):
        if v["qk_norm"]:
            for n in ("q_norm", "k_norm"):
                s["%s.%s.weight" % (base, n)] = (whd,)
                s["%s.%s.bias" % (base, n)] = (whd,)

    def ln(name, n):
        s[name + ".weight"] = (n,)
        s[name + ".bias"] = (n,)

    for i in range(v["num_decoder_layers"]):
        b = "vae.transformer.resblocks.%d" % i
        lin(b + ".attn.c_qkv", 3 * W, W, v["qkv_bias"])
        lin(b + ".attn.c_proj", W, W)
        qknorm(b + ".attn.attention")
        ln(b + ".ln_1", W)
        lin(b + ".mlp.c_fc", 4 * W, W)
        lin(b + ".mlp.c_proj", W, 4 * W)
        ln(b + ".ln_2", W)
    g = "vae.geo_decoder"
    e = v.get("geo_decoder_mlp_expand_ratio", 4)
    lin(g + ".query_proj", W, 3 * (2 * v["num_freqs"] + 1))
    lin(g + ".cross_attn_decoder.attn.c_q", W, W, v["qkv_bias"])
    lin(g + ".cross_attn_decoder.attn.c_kv", 2 * W, W, v["qkv_bias"])
    lin(g + ".cross_attn_decoder.attn.c_proj", W, W)
    if v.get("geo_decoder_ln_post", True):
        qknorm(g + ".cross_attn_decoder.attn.attention")
    for n in ("ln_1", "ln_2", "ln_3"):
        ln(g + ".cross_attn_decoder." + n, W)
    lin(g + ".cross_attn_decoder.mlp.c_fc", e * W, W)
    lin(g + ".cross_attn_decoder.mlp.c_proj", W, e * W)
    if v.get("geo_decoder_ln_post", True):
        ln(g + ".ln_post", W)
    lin(g + ".output_proj", 1, W)

    Hc, P = c["hidden_size"], c["image_size"] // c["patch_size"]
    F = swiglu_hidden(Hc, c["mlp_ratio"])
    m = "conditioner.main_image_encoder.model"
    s[m + ".embeddings.cls_token"] = (1, 1, Hc)
    s[m + ".embeddings.mask_token"] = (1, Hc)
    s[m + ".embeddings.position_embeddings"] = (1, P * P + 1, Hc)
    s[m + ".embeddings.patch_embeddings.projection.weight"] = (Hc, 3, c["patch_size"], c["patch_size"])
    s[m + ".embeddings.patch_embeddings.projection.bias"] = (Hc,)
    for i in range(c["num_hidden_layers"]):
        b = "%s.encoder.layer.%d" % (m, i)
        ln(b + ".norm1", Hc)
        for n in ("query", "key", "value"):
            lin(b + ".attention.attention." + n, Hc, Hc)
        lin(b + ".attention.output.dense", Hc, Hc)
        s[b + ".layer_scale1.lambda1"] = (Hc,)
        ln(b + ".norm2", Hc)
        lin(b + ".mlp.weights_in", 2 * F, Hc)
        lin(b + ".mlp.weights_out", Hc, F)
        s[b + ".layer_scale2.lambda1"] = (Hc,)
    ln(m + ".layernorm", Hc)
    return s


def _is_scale(name):
    return (name.endswith(".scale") or name.endswith("lambda1") or
            (name.endswith(".weight") and any(t in name for t in (
                "norm1.", "norm2.", "q_norm.", "k_norm.", "ln_1.", "ln_2.", "ln_3.", "ln_post.", "layernorm."))))


def synthetic_state_dict(cfg, seed=0, device="cuda", std=None):
    """Seeded synthetic checkpoint generated on `device` (timing / plumbing).  Default: every layer at
    unit scale -- Linear ~ N(0, 1/fan_in), adaLN modulation layers ~ N(0, 9/fan_in) with bias N(0, 0.3^2), norm scales
    1 +- 10 %, biases N(0, 0.1^2), Dinov2 cls / position embeddings N(0, 0.5^2) -- so that every branch contributes O(1)
    to its residual stream and the occupancy field changes sign (a surface exists at every model size).
    std = 0.02 (a float) reproduces the first-round checkpoint: Linear ~ N(0, std^2), near-identity blocks."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        is_mod = ".lin." in name or "adaLN" in name
        emb = name.endswith(("cls_token", "mask_token", "position_embeddings"))
        if _is_scale(name):
            t = 1.0 + (0.1 if std is None else 0.05) * torch.randn(shape, generator=g, device=device)
        elif emb:
            t = (0.5 if std is None else std) * torch.randn(shape, generator=g, device=device)
        elif len(shape) >= 2:
            if std is None:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                scale = (3.0 if is_mod else 1.0) / fan_in ** 0.5
            else:
                scale = 0.2 if "output_proj" in name else std
            t = scale * torch.randn(shape, generator=g, device=device)
        else:
            t = ((0.3 if is_mod else 0.1) if std is None else 0.01) * torch.randn(shape, generator=g, device=device)
        sd[name] = t
    return sd


def load_safetensors_dir(path, variant=None, use_safetensors=True):
    """<path>/model[.variant].safetensors (upstream single-file layout with model./vae./conditioner. prefixes), or -- upstream's
    `use_safetensors=False` / older snapshots -- <path>/model[.variant].ckpt: a torch pickle of {"model": sd, "vae": sd,
    "conditioner": sd} (loaded with weights_only=True), flattened to the same prefixed names.  [UPSTREAM-RECALLED: the two file
    names and the three top-level keys are how hy3dgen/shapegen/pipelines.py `from_single_file` reads them.]"""
    stems = (["model.%s" % variant] if variant else []) + ["model", "model.fp16"]
    if use_safetensors:
        from safetensors.torch import load_file
        for n in stems:
            f = os.path.join(path, n + ".safetensors")
            if os.path.exists(f):
                return load_file(f)
    for n in stems:
        f = os.path.join(path, n + ".ckpt")
        if os.path.exists(f):
            return flatten_ckpt(torch.load(f, map_location="cpu", weights_only=True))
    raise FileNotFoundError("no model*.safetensors / model*.ckpt under " + path)


def flatten_ckpt(ckpt):
    """{"model": {...}, "vae": {...}, "conditioner": {...}} (optionally under "state_dict") -> {"model.x": t, ...}; a dict that
    is already flat passes through"""
    if isinstance(ckpt, dict) and isinstance(ckpt.get("state_dict"), dict):
        ckpt = ckpt["state_dict"]
    if not isinstance(ckpt, dict) or not ckpt:
        raise ValueError("checkpoint is not a state dict")
    if all(torch.is_tensor(v) for v in ckpt.values()):
        return dict(ckpt)
    flat = {}
    for part, sd in ckpt.items():
        if not isinstance(sd, dict):
            continue          # e.g. a stored step counter
        for k, v in sd.items():
            if torch.is_tensor(v):
                flat["%s.%s" % (part, k)] = v
    if not flat:
        raise ValueError("checkpoint holds no tensors under its top-level keys %s" % sorted(ckpt))
    return flat
