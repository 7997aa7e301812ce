"""Multiview generation step of the texture stage, CPU side: camera indices, the "trailing" sigma table, the oracle loop, and
the image bookkeeping of Multiview_Diffusion_Net around a stub pipeline."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))

from oracle import mvpaint_torch as MP  # noqa: E402
from r3g import sched  # noqa: E402
from hy3dgen.texgen.utils import multiview_utils as MU  # noqa: E402


def test_camera_indices_fit_the_embedding_table():
    from hy3dgen.texgen.pipelines import DEFAULT_VIEWS
    views = [(e, a) for e, a, _ in DEFAULT_VIEWS]
    assert views == [(0, 0), (0, 90), (0, 180), (0, 270), (90, 0), (-90, 180)]     # [UPSTREAM-RECALLED] the view from below at azimuth 180
    idx = [MU.camera_index(e, a) for e, a in views]
    assert idx == [21, 12, 15, 18, 43, 37] and len(set(idx)) == 6
    every = {MU.camera_index(e, a) for e in (-20, 0, 20) for a in range(0, 360, 30)} | \
            {MU.camera_index(e, a) for e in (-90, 90) for a in range(0, 360, 90)}
    assert every == set(range(44))                       # 12 x 3 rings + 4 x 2 poles: upstream's max_num_gen_image
    with pytest.raises(ValueError):
        MU.camera_index(45, 0)


@pytest.mark.parametrize("n", [4, 30])
def test_trailing_table(n):
    ts, sg = MP.trailing_tables(n)
    e = sched.EulerAncestralDiscrete(timestep_spacing="trailing").set_timesteps(n)
    assert np.array_equal(ts, e.timesteps) and ts[0] == 999.0 and ts[-1] == round(1000 / n) - 1
    assert np.allclose(sg, e.sigmas, rtol=2e-5) and abs(e.init_noise_sigma - 14.6146) < 5e-5
    with pytest.raises(ValueError):
        sched.EulerAncestralDiscrete(timestep_spacing="leading")


def _models():
    from oracle import aekl_torch as A, unet2p5d_torch as M, unet_torch as U
    return M.build(U.small_config(), seed=2), A.build(A.small_config(), seed=5)


def _inputs(n, n_ref, size, steps, seed):
    g = torch.Generator().manual_seed(seed)
    h = size // 4
    img = lambda k: torch.rand(k, 3, size, size, generator=g) * 2 - 1
    dr = lambda k: torch.randn(k, 4, h, h, generator=g)
    noise = {"ref": dr(n_ref), "normal": dr(n), "position": dr(n), "latents": dr(n), "steps": [dr(n) for _ in range(steps)]}
    return img(n_ref), img(n), img(n), noise


def test_oracle_loop_guidance_one_skips_the_unconditional_branch():
    m, vae = _models()
    ref, nm, ps, noise = _inputs(3, 1, 32, 3, 0)
    a = MP.multiview_paint(m, vae, ref, nm, ps, [0, 1, 2], [0], 3, noise, guidance_scale=1.0, output="latent")
    calls = []
    orig = m.forward
    m.forward = lambda *x, **k: (calls.append(k.get("ref_scale", 1.0)), orig(*x, **k))[1]
    b = MP.multiview_paint(m, vae, ref, nm, ps, [0, 1, 2], [0], 3, noise, guidance_scale=2.0, output="latent")
    m.forward = orig
    assert calls == [1.0, 0.0] * 3 and a.shape == (3, 4, 8, 8) and (a - b).abs().max() > 1e-3
    img = MP.multiview_paint(m, vae, ref, nm, ps, [0, 1, 2], [0], 3, noise, guidance_scale=2.0)
    assert img.shape == (3, 3, 32, 32) and torch.isfinite(img).all()


class _StubPipeline:
    def __call__(self, ref, normal, position, cams, camera_info_ref, num_inference_steps, generator):
        assert ref.shape == (1, 3, 512, 512) and normal.shape == (6, 3, 512, 512) and position.shape == normal.shape
        assert cams == [21, 12, 15, 18, 43, 37] and camera_info_ref == [0] and num_inference_steps == 30
        assert generator.initial_seed() == 0 and float(normal.min()) >= -1 and float(normal.max()) <= 1
        return normal                                    # echo the normal maps


def test_multiview_diffusion_net_bookkeeping():
    from PIL import Image
    rng = np.random.default_rng(0)
    ctl = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8), "RGB") for _ in range(12)]
    net = MU.Multiview_Diffusion_Net(pipeline=_StubPipeline())
    out = net(Image.fromarray(rng.integers(0, 255, (80, 80, 3), dtype=np.uint8), "RGB"), ctl, [21, 12, 15, 18, 43, 37])
    assert len(out) == 6 and all(o.size == (512, 512) and o.mode == "RGB" for o in out)
    assert np.abs(np.asarray(out[2]).astype(int) - np.asarray(ctl[2].resize((512, 512))).astype(int)).max() <= 1
    with pytest.raises(ValueError):
        net(ctl[0], ctl, [0, 1])                          # one camera index per view
    with pytest.raises(ValueError):
        MU.Multiview_Diffusion_Net()


def test_recenter_image_of_the_paint_pipeline():
    from PIL import Image
    from hy3dgen.texgen import Hunyuan3DPaintPipeline as PP
    arr = np.zeros((100, 120, 4), np.uint8)
    arr[20:60, 30:110] = (10, 200, 30, 255)                       # a 40 x 80 object
    out = PP.recenter_image(Image.fromarray(arr, "RGBA"))
    assert out.mode == "RGBA" and out.size == (112, 112)           # 80 + 2 x 16 wide, 40 + 2 x 8 high -> square of 112
    o = np.asarray(out)
    ys, xs = np.nonzero(o[..., 3])
    assert (xs.min(), xs.max() + 1, ys.min(), ys.max() + 1) == (16, 96, 36, 76)      # centred, border 16 left and right
    assert tuple(o[50, 50]) == (10, 200, 30, 255) and tuple(o[0, 0]) == (255, 255, 255, 0)
    rgb = Image.new("RGB", (8, 8))
    assert PP.recenter_image(rgb) is rgb and PP.recenter_image(Image.new("L", (8, 8))).mode == "RGB"
    with pytest.raises(ValueError):
        PP.recenter_image(Image.fromarray(np.zeros((8, 8, 4), np.uint8), "RGBA"))
