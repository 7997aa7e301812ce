"""Host side of the sampling loop of upstream's delighting model ([UPSTREAM-RECALLED] hy3dgen/texgen/utils/dehighlight_utils.py:
a StableDiffusionInstructPix2PixPipeline whose scheduler is replaced by EulerAncestralDiscreteScheduler.from_config): the sigma
table of diffusers' EulerAncestralDiscreteScheduler (a few dozen floats, host arithmetic) and thin calls of the two elementwise
HIP steps (include/r3g.h "sampling loop of upstream's delighting model")."""
import ctypes

import numpy as np
import torch

from . import ffi as _l


class EulerAncestralDiscrete:
    """diffusers EulerAncestralDiscreteScheduler(num_train_timesteps 1000, beta_start 0.00085, beta_end 0.012,
    beta_schedule "scaled_linear", timestep_spacing "linspace"): SD's scheduler_config.json values"""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="epsilon", timestep_spacing="linspace"):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError("beta_schedule must be 'scaled_linear' or 'linear'")
        if prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError("prediction_type must be 'epsilon' or 'v_prediction'")
        if timestep_spacing not in ("linspace", "trailing"):
            raise ValueError("timestep_spacing must be 'linspace' or 'trailing'")
        self.prediction_type = prediction_type
        self.timestep_spacing = timestep_spacing      # upstream: "linspace" for the delighting model, "trailing" for the multiview one
        self.num_train_timesteps = int(num_train_timesteps)
        ac = torch.cumprod(1.0 - betas, dim=0)
        self._train_sigmas = (((1 - ac) / ac) ** 0.5).numpy()          # float32, ascending in t
        self.timesteps = None
        self.sigmas = None

    def set_timesteps(self, num_inference_steps):
        n = int(num_inference_steps)
        if n < 1:
            raise ValueError("num_inference_steps must be positive")
        if self.timestep_spacing == "linspace":
            t = np.linspace(0, self.num_train_timesteps - 1, n, dtype=np.float32)[::-1].copy()
        else:       # diffusers "trailing": round(arange(N, 0, -N / n)) - 1
            t = np.round(np.arange(self.num_train_timesteps, 0, -self.num_train_timesteps / n)).astype(np.float32) - 1
        s = np.interp(t, np.arange(0, len(self._train_sigmas)), self._train_sigmas)
        self.sigmas = np.concatenate([s, [0.0]]).astype(np.float32)
        self.timesteps = t
        return self

    @property
    def init_noise_sigma(self):
        return float(self.sigmas.max())

    # ---- the device steps
    def model_input(self, latent_rows, cond_rows, i, out=None):
        """rows f32 [pixels][c], [pixels][k] -> [pixels][c + k] = (latent / sqrt(sigma_i^2 + 1) | conditioning latents)"""
        n, c = latent_rows.shape
        k = cond_rows.shape[1]
        if out is None:
            out = torch.empty((n, c + k), dtype=torch.float32, device=latent_rows.device)
        with torch.cuda.device(latent_rows.device):
            _l.check(_l.lib().r3g_sched_model_input(latent_rows.data_ptr(), c, cond_rows.data_ptr(), k, n,
                                                    ctypes.c_float(float(self.sigmas[i])), out.data_ptr(),
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    @staticmethod
    def cfg_combine(uncond_rows, cond_rows, guidance_scale, out=None):
        """uncond + guidance_scale (cond - uncond)"""
        if out is None:
            out = torch.empty_like(cond_rows)
        with torch.cuda.device(cond_rows.device):
            _l.check(_l.lib().r3g_sched_cfg_combine(uncond_rows.data_ptr(), cond_rows.data_ptr(), cond_rows.numel(),
                                                    ctypes.c_float(float(guidance_scale)), out.data_ptr(),
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def step(self, sample_rows, model_out_rows, noise_rows, i):
        """in place on sample_rows: one EulerAncestralDiscreteScheduler.step from sigma_i to sigma_{i+1}"""
        with torch.cuda.device(sample_rows.device):
            _l.check(_l.lib().r3g_sched_euler_ancestral_step(sample_rows.data_ptr(), model_out_rows.data_ptr(), noise_rows.data_ptr(),
                                                             sample_rows.numel(), ctypes.c_float(float(self.sigmas[i])),
                                                             ctypes.c_float(float(self.sigmas[i + 1])),
                                                             1 if self.prediction_type == "v_prediction" else 0,
                                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return sample_rows
