"""GPU parity of the multiview generation step of the texture stage: the two new scheduler kernels, the whole
sampling loop (r3g.multiview.MultiviewPipeline: VAE encodes, reference pass, N x (model input, 2.5D UNet with and without the
reference attention, guidance, Euler-ancestral step), VAE decodes) against oracle/mvpaint_torch.py with the same noise, and
upstream's flow through Hunyuan3DPaintPipeline with both diffusion models plugged in.  Tolerance of the loop: 5e-2 rel-L2
(guidance 2 doubles the difference of two evaluations that are each at 0.8e-2)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from parity_support import rel_l2, report  # noqa: E402

pytestmark = pytest.mark.gpu


def test_model_input_and_guidance_kernels():
    import torch
    from r3g import sched
    g = torch.Generator().manual_seed(0)
    lat, cond = torch.randn(777, 4, generator=g), torch.randn(777, 8, generator=g)
    s = sched.EulerAncestralDiscrete(timestep_spacing="trailing").set_timesteps(5)
    got = s.model_input(lat.cuda(), cond.cuda(), 1).cpu()
    want = torch.cat([lat / (float(s.sigmas[1]) ** 2 + 1) ** 0.5, cond], dim=1)
    assert got.shape == (777, 12) and torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    u, c = torch.randn(5000, generator=g), torch.randn(5000, generator=g)
    got = s.cfg_combine(u.cuda(), c.cuda(), 2.0).cpu()
    assert torch.allclose(got, u + 2.0 * (c - u), rtol=1e-6, atol=1e-6)


class Pair:
    def __init__(self):
        import torch
        from oracle import aekl_torch as A, unet2p5d_torch as M, unet_torch as U
        from r3g import unet as RU
        from r3g.multiview import MultiviewPipeline, MultiviewUNet
        rnd = lambda sd: {k: (v.to(torch.bfloat16).float() if v.ndim >= 2 else v.clone()) for k, v in sd.items()}
        self.ucfg, self.vcfg = U.small_config(), A.small_config()
        self.unet = M.build(self.ucfg, seed=2)
        usd = rnd(self.unet.state_dict())
        self.unet.load_state_dict(usd, strict=True)
        self.vae = A.build(self.vcfg, seed=5)
        vsd = rnd(self.vae.state_dict())
        self.vae.load_state_dict(vsd, strict=True)
        size = 64
        self.gpu_vae = RU.AutoencoderKLBlocks(vsd, block_out_channels=self.vcfg["block_out_channels"],
                                              layers_per_block=self.vcfg["layers_per_block"], max_image_hw=size * size)
        self.gpu_unet = MultiviewUNet(usd, self.ucfg, n_views_max=6, n_ref_max=1, latent_hw=(size // 4) ** 2)
        self.pipe = MultiviewPipeline(self.gpu_unet, self.gpu_vae)


@pytest.fixture(scope="module")
def pair():
    return Pair()


def _inputs(n, n_ref, size, steps, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    h = size // 4
    img = lambda k: torch.rand(k, 3, size, size, generator=g) * 2 - 1
    dr = lambda k: torch.randn(k, 4, h, h, generator=g)
    noise = {"ref": dr(n_ref), "normal": dr(n), "position": dr(n), "latents": dr(n), "steps": [dr(n) for _ in range(steps)]}
    return img(n_ref), img(n), img(n), noise


@pytest.mark.parametrize("n,size,steps,g", [(3, 32, 4, 2.0), (6, 64, 8, 2.0), (6, 32, 30, 1.0)])
def test_sampling_loop_small(pair, n, size, steps, g):
    import torch
    from oracle import mvpaint_torch as MP
    ref, nm, ps, noise = _inputs(n, 1, size, steps, 50 + steps)
    cams = list(range(n))
    want_z, want = MP.multiview_paint(pair.unet, pair.vae, ref, nm, ps, cams, [0], steps, noise, guidance_scale=g, output="both")
    got_z = pair.pipe(ref, nm, ps, cams, [0], num_inference_steps=steps, guidance_scale=g, noise=noise, output="latent").cpu()
    got = pair.pipe(ref, nm, ps, cams, [0], num_inference_steps=steps, guidance_scale=g, noise=noise).cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    ez, ei = rel_l2(got_z, want_z), rel_l2(got, want)
    report("mvpaint.loop %d views %dx%d, %d steps, guidance %.0f: final latents" % (n, size, size, steps, g), ez, 5e-2)
    report("mvpaint.loop %d views %dx%d, %d steps, guidance %.0f: decoded views" % (n, size, size, steps, g), ei, 5e-2)
    assert ez < 5e-2 and ei < 5e-2, (ez, ei)


def test_paired_guidance_is_the_two_launch_sets(pair):
    """the guided loop with both evaluations of a step in ONE launch set (the default since round 5) against the same loop with two
    launch sets per step (rounds 3-4): the same latents after 8 steps up to the fp32 addition order of the split-K convolutions"""
    import torch
    from r3g.multiview import MultiviewPipeline
    ref, nm, ps, noise = _inputs(6, 1, 64, 8, 77)
    cams = list(range(6))
    assert pair.pipe.paired_guidance and pair.gpu_unet.pair_capacity
    one = pair.pipe(ref, nm, ps, cams, [0], num_inference_steps=8, guidance_scale=2.0, noise=noise, output="latent").cpu()
    two_sets = MultiviewPipeline(pair.gpu_unet, pair.gpu_vae, paired_guidance=False)
    two = two_sets(ref, nm, ps, cams, [0], num_inference_steps=8, guidance_scale=2.0, noise=noise, output="latent").cpu()
    e = rel_l2(one, two)
    report("mvpaint.loop 6 views 64x64, 8 steps: one launch set per guided step vs two", e, 1e-4)
    assert e <= 1e-4


def test_upstream_flow_through_the_paint_pipeline(pair):
    """delight -> unwrap -> normal / position maps of six views -> multiview diffusion -> bake -> inpaint, every model on the HIP
    blocks (random weights: the colours mean nothing, the plumbing is what is checked)"""
    import torch
    from hy3dgen.texgen import Hunyuan3DPaintPipeline
    from hy3dgen.texgen.utils.dehighlight_utils import Light_Shadow_Remover
    from hy3dgen.texgen.utils.multiview_utils import Multiview_Diffusion_Net
    from oracle import unet_torch as U
    from r3g.delight import InstructPix2Pix
    from r3g.mesh import Mesh
    import tex_support as ts
    from PIL import Image

    class Net(Multiview_Diffusion_Net):
        view_size, steps = 64, 3

    class Delight(Light_Shadow_Remover):
        size, steps = 64, 2

    calls = {}

    class Spy(Net):
        def __call__(self, input_images, control_images, camera_info):
            calls["n_control"], calls["cams"] = len(control_images), list(camera_info)
            calls["normal0"] = np.asarray(control_images[0])
            calls["position0"] = np.asarray(control_images[6])
            return super().__call__(input_images, control_images, camera_info)

    ucfg = dict(U.small_config(), in_channels=8, out_channels=4)
    usd = {k: (v.to(torch.bfloat16).float() if v.ndim >= 2 else v) for k, v in U.synthetic_state_dict(ucfg, seed=3, full=True).items()}
    vsd = {k: (v.to(torch.bfloat16).float() if v.ndim >= 2 else v) for k, v in pair.vae.state_dict().items()}
    delight = Delight(model=InstructPix2Pix(usd, vsd, ucfg, pair.vcfg, image_size=64),
                      prompt_embeds=torch.randn(1, ucfg["ctx_tokens"], ucfg["cross_attention_dim"]))
    v, f = ts.icosphere(3)
    img = np.zeros((96, 96, 4), np.uint8)
    img[..., :3] = 180
    yy, xx = np.mgrid[0:96, 0:96]
    img[..., 3] = np.where((xx - 47.5) ** 2 + (yy - 47.5) ** 2 < 40 ** 2, 255, 0)
    pipe = Hunyuan3DPaintPipeline(texture_size=256, render_size=128, multiview_model=Spy(pipeline=pair.pipe), delight_model=delight)
    out = pipe(Mesh(v, f), image=Image.fromarray(img, "RGBA"))
    assert calls["n_control"] == 12 and calls["cams"] == [21, 12, 15, 18, 43, 37]
    n0, p0 = calls["normal0"], calls["position0"]
    assert n0.shape == (64, 64, 3) and (n0[0, 0] == 255).all() and (p0[0, 0] == 255).all()       # white outside the silhouette
    c = n0[32, 32].astype(int)                                                                    # front view, centre: n = +z
    assert abs(c[0] - 128) < 12 and abs(c[1] - 128) < 12 and c[2] > 240
    assert abs(int(p0[32, 32][2]) - int(255 * (1 / (2 * 1.05) + 0.5))) < 6                          # z = 1 on the unit sphere
    assert out.texture.shape == (256, 256, 3) and out.metadata["texture_source"].startswith("delighted input; multiview diffusion model")
    st = pipe.last_stats
    assert st["texels_painted_by_views"] > 0.8 * st["texels_covered"]        # six views see (nearly) the whole sphere


def test_both_models_load_from_checkpoint_folders_and_match_their_oracles(tmp_path, monkeypatch):
    """Hunyuan3DPaintPipeline.from_pretrained(<folder with hunyuan3d-delight-v2-0/ and hunyuan3d-paint-v2-0/>): the state dicts
    travel file -> loader -> re-layout -> HBM, and what runs there matches the oracle built from the same tensors"""
    import torch
    from PIL import Image
    import tex_ckpt_support as CK
    import tex_support as ts
    from hy3dgen.texgen import Hunyuan3DPaintPipeline
    from hy3dgen.texgen.utils.dehighlight_utils import Light_Shadow_Remover
    from hy3dgen.texgen.utils.multiview_utils import Multiview_Diffusion_Net
    from oracle import mvpaint_torch as MP, pix2pix_torch as P
    from r3g.mesh import Mesh
    dl, mv = CK.write_checkpoints(str(tmp_path))
    monkeypatch.setattr(Light_Shadow_Remover, "size", 64)
    monkeypatch.setattr(Light_Shadow_Remover, "steps", 2)
    monkeypatch.setattr(Multiview_Diffusion_Net, "view_size", 64)
    monkeypatch.setattr(Multiview_Diffusion_Net, "steps", 2)
    pipe = Hunyuan3DPaintPipeline.from_pretrained(str(tmp_path), texture_size=256, render_size=128)
    assert pipe.source.startswith("delighted input; multiview diffusion model")
    # the delighting model as loaded
    g = torch.Generator().manual_seed(1)
    img = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    lat = torch.randn(1, 4, 16, 16, generator=g)
    noise = [torch.randn(1, 4, 16, 16, generator=g) for _ in range(3)]
    want = P.instruct_pix2pix(dl["unet"], dl["vae"], dl["prompt_embeds"], img, 3, lat, noise)
    got = pipe.delight_model.model(img, pipe.delight_model.prompt_embeds, num_inference_steps=3, latents=lat, step_noise=noise).cpu()
    e = rel_l2(got, want)
    report("texckpt.delight model loaded from its folder, 3 steps", e, 3e-2)
    assert e < 3e-2
    # the multiview model as loaded (its UNet from a torch pickle)
    ref, nm, ps, nz = _inputs(6, 1, 64, 3, 7)
    want = MP.multiview_paint(mv["unet"], mv["vae"], ref, nm, ps, list(range(6)), [0], 3, nz)
    got = pipe.multiview_model.pipeline(ref, nm, ps, list(range(6)), [0], num_inference_steps=3, noise=nz).cpu()
    e = rel_l2(got, want)
    report("texckpt.multiview model loaded from its folder, 3 guided steps", e, 5e-2)
    assert e < 5e-2
    # and the stage's call
    v, f = ts.icosphere(3)
    rgba = np.zeros((96, 96, 4), np.uint8)
    rgba[..., :3] = 200
    yy, xx = np.mgrid[0:96, 0:96]
    rgba[..., 3] = np.where((xx - 47.5) ** 2 + (yy - 47.5) ** 2 < 40 ** 2, 255, 0)
    out = pipe(Mesh(v, f), image=Image.fromarray(rgba, "RGBA"))
    assert out.texture.shape == (256, 256, 3) and pipe.last_stats["texels_painted_by_views"] > 0.8 * pipe.last_stats["texels_covered"]
