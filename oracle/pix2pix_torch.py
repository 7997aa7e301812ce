"""ORACLE (test infrastructure, not product code): PyTorch-CPU fp32 restatement of the sampling loop of upstream's delighting
model -- diffusers `StableDiffusionInstructPix2PixPipeline.__call__` with an `EulerAncestralDiscreteScheduler`
([UPSTREAM-RECALLED] hy3dgen/texgen/utils/dehighlight_utils.py `Light_Shadow_Remover`: prompt "", guidance_scale 1.0,
image_guidance_scale 1.5, 50 steps, generator seeded 42; with guidance_scale 1.0 diffusers' pipeline does NOT run
classifier-free guidance -- `do_classifier_free_guidance = guidance_scale > 1.0 and image_guidance_scale >= 1.0` -- so every
step is ONE UNet evaluation on cat(latents, image_latents)).  Behind reference src/2d_to_3d_models/run.py:97.

PARITY UNPINNED for the pipeline as a whole (no diffusers, no checkpoints, no golden outputs in the container).  The
scheduler's sigma table IS pinned: the published k-diffusion / SD constants sigma_max = 14.6146, sigma_min = 0.0292 of the
scaled-linear schedule (tests/test_pix2pix_cpu.py).
"""
import numpy as np
import torch


def train_sigmas(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """sqrt((1 - abar_t) / abar_t) of SD's scaled-linear schedule, float64 closed form (independent of the product's table)"""
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
    abar = np.cumprod(1.0 - betas)
    return np.sqrt((1.0 - abar) / abar)


def euler_ancestral_tables(num_inference_steps, num_train_timesteps=1000):
    """(timesteps [n] descending, sigmas [n + 1] with a final 0): timestep_spacing "linspace", sigmas interpolated linearly"""
    ts = np.linspace(0, num_train_timesteps - 1, num_inference_steps)[::-1].copy()
    s = np.interp(ts, np.arange(num_train_timesteps), train_sigmas(num_train_timesteps))
    return ts.astype(np.float32), np.concatenate([s, [0.0]]).astype(np.float32)


def euler_ancestral_step(sample, model_out, noise, sigma_from, sigma_to, prediction_type="epsilon"):
    sf, st = float(sigma_from), float(sigma_to)
    if prediction_type == "epsilon":
        x0 = sample - sf * model_out
    else:
        x0 = model_out * (-sf / (sf ** 2 + 1) ** 0.5) + sample / (sf ** 2 + 1)
    up = (st ** 2 * (sf ** 2 - st ** 2) / sf ** 2) ** 0.5
    down = (st ** 2 - up ** 2) ** 0.5
    d = (sample - x0) / sf
    return sample + d * (down - sf) + noise * up


@torch.no_grad()
def instruct_pix2pix(unet, vae, prompt_embeds, image, num_inference_steps, latents, step_noise, scaling_factor=0.18215,
                     prediction_type="epsilon", output="image"):
    """image NCHW in [-1, 1]; latents: the initial N(0,1) draw [1, z, h, w]; step_noise: list of N(0,1) draws, one per step.
    -> decoded image NCHW.  (image_latents are the MODE of the encoder's distribution and are NOT multiplied by the scaling
    factor -- InstructPix2Pix's convention; the result is divided by it before decoding.)"""
    ts, sig = euler_ancestral_tables(num_inference_steps)
    image_latents = vae.encode_mode(image)
    x = latents * float(sig.max())
    for i, t in enumerate(ts):
        inp = torch.cat([x / (float(sig[i]) ** 2 + 1) ** 0.5, image_latents], dim=1)
        eps = unet(inp, float(t), prompt_embeds)
        x = euler_ancestral_step(x, eps, step_noise[i], sig[i], sig[i + 1], prediction_type)
    if output == "latent":
        return x
    if output == "both":           # (final latents, decoded image) of ONE run of the loop
        return x, vae.decode(x / scaling_factor)
    return vae.decode(x / scaling_factor)
