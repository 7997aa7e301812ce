"""The multiview diffusion UNet of upstream's texture stage on the HIP blocks: [UPSTREAM-RECALLED]
hy3dgen/texgen/hunyuanpaint/unet/modules.py `UNet2p5DConditionModel` -- an SD-2.1 UNet whose transformer blocks also attend
across the views of the object (attn_multiview) and to the hidden states of a reference pass over the input image (attn_refview),
with a camera embedding added to the time embedding and 12 input channels (latent | normal-map latent | position-map latent).
Two library contexts: the generator and the plain 4-channel reference copy (`unet_dual`); the reference pass keeps norm1's output
of every transformer in HBM and the generator reads it from there (no copy, no host round trip).

MultiviewUNet is the model (one evaluation = r3g_unet_forward_mv over all views); MultiviewPipeline is upstream's sampling loop
around it (VAE encoding of the image and of the rendered normal / position maps, classifier-free guidance, Euler-ancestral steps,
VAE decoding of the views)."""
import torch

from . import unet as _unet


class MultiviewUNet:
    max_num_ref_image = 5       # upstream: camera indices of generated views are offset by the reference slots

    def __init__(self, state_dict, unet_config, n_views_max=6, n_ref_max=1, latent_hw=64 * 64, device=0):
        """state_dict: upstream's UNet2p5DConditionModel ("unet.*", "unet_dual.*"); unet_config: block_out_channels,
        layers_per_block, cross_attention_dim, ctx_tokens, temb_dim, groups"""
        gen_sd, ref_sd, extra = _unet.split_2p5d_state_dict(state_dict)
        c = unet_config
        ch = tuple(c["block_out_channels"])
        self.n_levels, self.layers = len(ch), c["layers_per_block"]
        common = dict(max_channels=2 * max(ch), temb_dim=c["temb_dim"], ctx_dim=c["cross_attention_dim"], groups=c["groups"],
                      device=device, block_out_channels=ch, layers_per_block=c["layers_per_block"], out_channels=4)
        # the generator's arena holds the TWO evaluations of a guidance step side by side (2 x n_views_max samples, <= 16)
        self.pair_capacity = 2 * n_views_max <= 16
        self.gen = _unet.UnetBlocks(gen_sd, max_hw=(2 if self.pair_capacity else 1) * n_views_max * latent_hw,
                                    ctx_tokens=max(c["ctx_tokens"], n_ref_max * latent_hw), in_channels=12, **common)
        self.ref = _unet.UnetBlocks(ref_sd, max_hw=n_ref_max * latent_hw, ctx_tokens=c["ctx_tokens"], in_channels=4, **common)
        self.text_gen = extra["learned_text_clip_gen"].detach().reshape(1, -1, c["cross_attention_dim"])
        self.text_ref = extra["learned_text_clip_ref"].detach().reshape(1, -1, c["cross_attention_dim"])
        self.has_reference = False

    def close(self):
        """give both contexts' HBM back (the generator first: it points into the reference copy's kept states)"""
        self.gen.close()
        self.ref.close()

    def reference_pass(self, ref_latents, camera_info_ref=None):
        """ref_latents NCHW [n_ref, 4, h, w] of ONE object: the reference copy runs once at timestep 0 and keeps every
        transformer's normalised hidden states; the generator is pointed at them"""
        self.ref.forward_mv(ref_latents, 0.0, self.text_ref, class_labels=camera_info_ref, flags=1)
        for p in _unet.transformer_prefixes(self.n_levels, self.layers):
            self.gen.set_condition(p, self.ref)
        self.has_reference = True

    def __call__(self, sample, timestep, normal_imgs, position_imgs, camera_info_gen=None, mva_scale=1.0, ref_scale=1.0):
        """sample / normal_imgs / position_imgs NCHW [n_views, 4, h, w] -> noise prediction [n_views, 4, h, w]"""
        x = torch.cat([sample, normal_imgs, position_imgs], dim=1)
        labels = None if camera_info_gen is None else [int(v) + self.max_num_ref_image for v in camera_info_gen]
        return self.gen.forward_mv(x, timestep, self.text_gen, class_labels=labels, flags=2 if self.has_reference else 0,
                                   mva_scale=mva_scale, ref_scale=ref_scale)


class MultiviewPipeline:
    """The sampling loop around the multiview UNet: [UPSTREAM-RECALLED] hy3dgen/texgen/hunyuanpaint/pipeline.py
    `HunyuanPaintPipeline` as `Multiview_Diffusion_Net` drives it -- the delighted image, the normal maps and the position maps of
    the views are VAE-encoded (a sample of the latent distribution times the scaling factor), the reference copy of the UNet runs
    once on the image's latents, then `num_inference_steps` (30) Euler-ancestral steps ("trailing" spacing) on the views' latents,
    every step one evaluation with the reference attention on and, for guidance_scale > 1, one with it off (upstream's
    unconditional branch: zero reference latents with ref_scale 0), combined as uncond + g (cond - uncond); the views are decoded
    one by one.  Latents, conditioning latents and noise stay in HBM as rows for the whole loop."""

    def __init__(self, unet, vae, scaling_factor=0.18215, prediction_type="epsilon", uncond_context="zeros", paired_guidance=True):
        """uncond_context: the text context of the UNCONDITIONAL branch of classifier-free guidance.  "zeros" (default):
        [UPSTREAM-RECALLED] HunyuanPaintPipeline sets negative_prompt_embeds = zeros_like(prompt_embeds) -- the branch runs without
        the reference attention AND on an all-zero context; "learned": the learned embedding in both branches (rounds 2-3).
        Unpinned: to be checked against the public source"""
        from . import sched as _sched
        if uncond_context not in ("zeros", "learned"):
            raise ValueError("uncond_context must be 'zeros' or 'learned'")
        self.uncond_context = uncond_context
        # paired_guidance: both evaluations of a guided step as ONE launch set of 2 n samples (round 5; r3g_unet_forward_mv flag 4)
        # where the generator's arena has the room; False: two launch sets of n samples (rounds 3-4)
        self.paired_guidance = bool(paired_guidance)
        self.unet, self.vae = unet, vae
        self.scaling_factor = float(scaling_factor)
        self.scheduler = _sched.EulerAncestralDiscrete(prediction_type=prediction_type, timestep_spacing="trailing")
        self.device = vae.device

    def encode(self, images, noise):
        """images NCHW [n, 3, H, W] in [-1, 1]; noise NCHW [n, z, h, w] (N(0,1)) -> latent_dist.sample() * scaling_factor"""
        zc = self.vae.latent_channels
        out = []
        for i in range(images.shape[0]):
            mom = self.vae.encode(images[i:i + 1])
            mean, logvar = mom[:, :zc], mom[:, zc:].clamp(-30.0, 20.0)
            out.append((mean + torch.exp(0.5 * logvar) * noise[i:i + 1].to(self.device)) * self.scaling_factor)
        return torch.cat(out, dim=0)

    def __call__(self, ref_images, normal_imgs, position_imgs, camera_info_gen, camera_info_ref=None, num_inference_steps=30,
                 guidance_scale=2.0, generator=None, noise=None, output="image"):
        """ref_images [n_ref, 3, H, W], normal_imgs / position_imgs [n, 3, H, W] in [-1, 1] -> images NCHW [n, 3, H, W].
        noise: dict with "ref" [n_ref, z, h, w], "normal", "position", "latents" [n, z, h, w] and "steps" (one [n, z, h, w] per
        step); drawn from `generator` in that order when absent"""
        n, n_ref = normal_imgs.shape[0], ref_images.shape[0]
        f, zc = self.vae.factor, self.vae.latent_channels
        H, W = normal_imgs.shape[2:]
        h, w = H // f, W // f
        steps = int(num_inference_steps)
        if noise is None:
            draw = lambda k: torch.randn((k, zc, h, w), generator=generator, dtype=torch.float32)
            noise = {"ref": draw(n_ref), "normal": draw(n), "position": draw(n), "latents": draw(n)}
            noise["steps"] = [draw(n) for _ in range(steps)]
        dev = self.device
        ref_latents = self.encode(ref_images, noise["ref"])
        cond_rows = _unet.to_rows(torch.cat([self.encode(normal_imgs, noise["normal"]),
                                             self.encode(position_imgs, noise["position"])], dim=1))       # [n h w][2 z]
        self.unet.reference_pass(ref_latents, camera_info_ref)
        labels = None if camera_info_gen is None else [int(v) + self.unet.max_num_ref_image for v in camera_info_gen]
        ctx = self.unet.text_gen[0].to(dev, torch.bfloat16).contiguous()
        ctx_u = torch.zeros_like(ctx) if self.uncond_context == "zeros" else ctx
        sch = self.scheduler.set_timesteps(steps)
        x = _unet.to_rows(noise["latents"].to(dev)) * sch.init_noise_sigma
        step_noise = [_unet.to_rows(e.to(dev)) for e in noise["steps"]]
        guided = guidance_scale > 1.0
        paired = guided and self.paired_guidance and getattr(self.unet, "pair_capacity", False) and 2 * n <= 16
        rows1 = n * h * w
        inp2 = torch.empty(((2 if paired else 1) * rows1, 3 * zc), dtype=torch.float32, device=dev)
        inp = inp2[:rows1]
        eps2 = torch.empty((2 * rows1, zc), dtype=torch.float32, device=dev)
        eps_c, eps_u = eps2[:rows1], eps2[rows1:]
        ctx2 = torch.cat([ctx, ctx_u], dim=0).contiguous()
        labels2 = None if labels is None else list(labels) + list(labels)
        gen = self.unet.gen
        for i in range(steps):
            t = float(sch.timesteps[i])
            sch.model_input(x, cond_rows, i, out=inp)
            if paired:
                inp2[rows1:].copy_(inp)
                gen.forward_mv_rows(inp2, 2 * n, h, w, t, ctx2, class_labels=labels2, flags=2 | 4, out=eps2)
            else:
                gen.forward_mv_rows(inp, n, h, w, t, ctx, class_labels=labels, flags=2, out=eps_c)
                if guided:
                    gen.forward_mv_rows(inp, n, h, w, t, ctx_u, class_labels=labels, flags=0, out=eps_u)
            if guided:
                sch.cfg_combine(eps_u, eps_c, guidance_scale, out=eps_c)
            sch.step(x, eps_c, step_noise[i], i)
        if output == "latent":
            return _unet.from_rows(x, h, w, n)
        x = x / self.scaling_factor
        views = [self.vae.decode_rows(x[k * h * w:(k + 1) * h * w].contiguous(), h, w)[:, :self.vae.image_channels] for k in range(n)]
        return torch.cat([_unet.from_rows(v, H, W) for v in views], dim=0)
