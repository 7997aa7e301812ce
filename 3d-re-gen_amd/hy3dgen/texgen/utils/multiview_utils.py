"""hy3dgen.texgen.utils.multiview_utils -- MI355X mirror of upstream's multiview generation step (`Multiview_Diffusion_Net`),
what `Hunyuan3DPaintPipeline.__call__` produces its six views with (behind reference src/2d_to_3d_models/run.py:97).

[UPSTREAM-RECALLED] upstream loads `config.multiview_ckpt_path` as a diffusers pipeline with the custom `hunyuanpaint` pipeline,
replaces the scheduler by EulerAncestralDiscreteScheduler.from_config(..., timestep_spacing='trailing'), and its __call__ is
`(input_images, control_images, camera_info)`: everything resized to the view size 512, control_images = the normal maps of the
views followed by their position maps, seed 0, camera_info_ref [[0]], 30 steps; it returns the views as PIL images.
`camera_info` is computed by the caller from the views' elevation / azimuth (camera_index below).  Names, signature and those
constants are upstream's as recalled; nothing here is pinned against upstream.

The model itself runs on the HIP blocks (r3g.multiview.MultiviewPipeline: 2.5D UNet + SD VAE + Euler-ancestral loop)."""
import numpy as np


def camera_index(elev, azim):
    """upstream's camera_info entry of a view: the azimuth in 30 degree steps (rotated by 9, 12 per ring; 4 per ring at the
    poles) plus the ring's offset -- rings at elevation -20 / 0 / 20 / -90 / 90"""
    ring = {-20: (1, 0), 0: (1, 12), 20: (1, 24), -90: (3, 36), 90: (3, 40)}
    if int(elev) not in ring:
        raise ValueError("the camera embedding knows the elevations -20, 0, 20, -90, 90")
    div, off = ring[int(elev)]
    return (((int(azim) // 30) + 9) % 12) // div + off


class Multiview_Diffusion_Net:
    view_size = 512
    steps = 30
    seed = 0
    wants_control_images = True          # Hunyuan3DPaintPipeline renders normal / position maps for this model

    def __init__(self, config=None, pipeline=None):
        """pipeline: an r3g.multiview.MultiviewPipeline; or a config that carries `multiview_ckpt_path` (upstream's attribute):
        the checkpoint directory is read here (load)"""
        if pipeline is None:
            path = getattr(config, "multiview_ckpt_path", None) if config is not None else None
            if not path:
                raise ValueError("Multiview_Diffusion_Net needs a pipeline or config.multiview_ckpt_path")
            pipeline = self.load(path, device=getattr(config, "device", 0))
        self.pipeline = pipeline

    @classmethod
    def load(cls, path, device=0, n_views_max=6):
        """upstream's multiview checkpoint directory ([UPSTREAM-RECALLED] layout: a diffusers pipeline folder whose unet/ holds
        the UNet2p5DConditionModel state dict -- "unet.*", "unet_dual.*" -- next to the wrapped UNet's config.json, a vae/ and a
        scheduler/): -> r3g.multiview.MultiviewPipeline sized for `n_views_max` views of view_size x view_size"""
        import os
        from r3g.multiview import MultiviewPipeline, MultiviewUNet
        from r3g.unet import AutoencoderKLBlocks
        from .dehighlight_utils import read_json, read_weights, unet_config_from_diffusers, vae_config_from_diffusers
        sd = read_weights(os.path.join(path, "unet"))
        text = [v for k, v in sd.items() if k.endswith("learned_text_clip_gen")]
        if not text:
            raise ValueError("%s/unet holds no 'unet.learned_text_clip_gen': not a UNet2p5DConditionModel state dict" % path)
        unet_config = unet_config_from_diffusers(read_json(os.path.join(path, "unet", "config.json")), text[0].shape[-2])
        vc = read_json(os.path.join(path, "vae", "config.json"))
        vcfg = vae_config_from_diffusers(vc)
        factor = 2 ** (len(vcfg["block_out_channels"]) - 1)
        lat = cls.view_size // factor
        vae = AutoencoderKLBlocks(read_weights(os.path.join(path, "vae")), block_out_channels=vcfg["block_out_channels"],
                                  layers_per_block=vcfg["layers_per_block"], latent_channels=vcfg["latent_channels"],
                                  image_channels=vcfg["image_channels"], groups=vcfg["groups"],
                                  max_image_hw=cls.view_size * cls.view_size, device=device)
        unet = MultiviewUNet(sd, unet_config, n_views_max=n_views_max, n_ref_max=1, latent_hw=lat * lat, device=device)
        sched_cfg = read_json(os.path.join(path, "scheduler", "scheduler_config.json"), default={})
        return MultiviewPipeline(unet, vae, scaling_factor=vc.get("scaling_factor", 0.18215),
                                 prediction_type=sched_cfg.get("prediction_type", "epsilon"))

    @staticmethod
    def _tensor(images, size):
        import torch
        from PIL import Image
        arrs = []
        for im in images:
            if not isinstance(im, Image.Image):
                im = Image.fromarray((np.clip(np.asarray(im, np.float32), 0, 1) * 255 + 0.5).astype(np.uint8))
            if im.mode in ("RGBA", "LA"):
                # an image prompt that still carries alpha (no delighting model in front: that step composites): over white,
                # as Light_Shadow_Remover does -- whatever RGB sits under alpha = 0 must not become the reference image
                rgba = np.asarray(im.convert("RGBA"), np.float32) / 255.0
                a = rgba[:, :, 3:4]
                im = Image.fromarray(((rgba[:, :, :3] * a + (1.0 - a)) * 255.0 + 0.5).astype(np.uint8), "RGB")
            im = im.convert("RGB").resize((size, size))
            arrs.append(np.asarray(im, np.float32) / 255.0)
        return torch.from_numpy(np.stack(arrs)).permute(0, 3, 1, 2) * 2.0 - 1.0

    def __call__(self, input_images, control_images, camera_info):
        import torch
        from PIL import Image
        if not isinstance(input_images, (list, tuple)):
            input_images = [input_images]
        num_view = len(control_images) // 2
        if len(camera_info) != num_view:
            raise ValueError("one camera index per view")
        s = self.view_size
        ref = self._tensor(input_images, s)
        normal = self._tensor(control_images[:num_view], s)
        position = self._tensor(control_images[num_view:], s)
        out = self.pipeline(ref, normal, position, list(camera_info), camera_info_ref=[0] * len(input_images),
                            num_inference_steps=self.steps, generator=torch.Generator().manual_seed(self.seed))
        out = (out / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().cpu().numpy()
        return [Image.fromarray((v * 255).round().astype(np.uint8)) for v in out]
