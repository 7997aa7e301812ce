"""Delighting step of the texture stage, CPU side: the Euler-ancestral sigma table (pinned against the published constants of
SD's scaled-linear schedule), the oracle's step, the host-side image bookkeeping of Light_Shadow_Remover, and the product's
scheduler table against the oracle's independent closed form."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))

from oracle import pix2pix_torch as P  # noqa: E402
from r3g import sched  # noqa: E402
from hy3dgen.texgen.utils import dehighlight_utils as D  # noqa: E402


def test_sigma_table_has_the_published_range():
    s = P.train_sigmas()
    assert abs(s[-1] - 14.6146) < 5e-5 and abs(s[0] - 0.0292) < 5e-5          # k-diffusion / SD: sigma_max, sigma_min
    assert np.all(np.diff(s) > 0)


@pytest.mark.parametrize("n", [1, 4, 50])
def test_product_table_matches_the_oracles_closed_form(n):
    ts, sg = P.euler_ancestral_tables(n)
    e = sched.EulerAncestralDiscrete().set_timesteps(n)
    assert np.array_equal(ts, e.timesteps) and ts[-1] == 0.0 and (n == 1 or ts[0] == 999.0)      # np.linspace(0, 999, 1) = [0]
    assert np.allclose(sg, e.sigmas, rtol=2e-5, atol=0)  # the product follows diffusers (float32 betas: 1 - abar_0 cancels)
    assert e.sigmas[-1] == 0.0 and len(e.sigmas) == n + 1
    assert n == 1 or abs(e.init_noise_sigma - 14.6146) < 5e-5


def test_last_step_lands_on_the_predicted_original_and_no_noise_enters():
    g = torch.Generator().manual_seed(0)
    x, m, z = (torch.randn(1, 4, 8, 8, generator=g) for _ in range(3))
    for pred in ("epsilon", "v_prediction"):
        out = P.euler_ancestral_step(x, m, z, 0.7, 0.0, pred)
        x0 = x - 0.7 * m if pred == "epsilon" else m * (-0.7 / (0.49 + 1) ** 0.5) + x / 1.49
        assert torch.allclose(out, x0, atol=1e-6)


def test_step_keeps_the_marginal_variance():
    """x = x0 + sigma_from n with the exact eps = n: the step must land on x0 + sigma_down n + sigma_up z, whose variance around
    x0 is sigma_to^2 -- the ancestral split sigma_down^2 + sigma_up^2 = sigma_to^2"""
    g = torch.Generator().manual_seed(1)
    x0, n, z = (torch.randn(4096, generator=g, dtype=torch.float64) for _ in range(3))
    sf, st = 3.0, 1.2
    out = P.euler_ancestral_step(x0 + sf * n, n, z, sf, st)
    up = (st ** 2 * (sf ** 2 - st ** 2) / sf ** 2) ** 0.5
    down = (st ** 2 - up ** 2) ** 0.5
    assert torch.allclose(out, x0 + down * n + up * z, atol=1e-9)
    assert abs(float((out - x0).var()) - st ** 2) < 0.1


def test_erode3_is_the_3x3_minimum_with_a_neutral_border():
    rng = np.random.default_rng(0)
    a = (rng.random((17, 23)) > 0.3).astype(np.uint8) * 255
    ref = a.copy()
    for y in range(17):
        for x in range(23):
            ref[y, x] = a[max(0, y - 1):y + 2, max(0, x - 1):x + 2].min()
    assert np.array_equal(D.erode3(a), ref)
    assert np.array_equal(D.erode3(np.full((4, 4), 255, np.uint8)), np.full((4, 4), 255, np.uint8))


def test_recorrect_rgb_matches_statistics_and_keeps_the_better_image():
    rng = np.random.default_rng(1)
    target = rng.random((32, 32, 3))
    alpha = np.zeros((32, 32, 1))
    alpha[4:28, 6:30] = 1.0
    src = np.clip(0.5 * target + 0.2, 0, 1)                         # an affine colour cast: the correction undoes most of it
    out = D.recorrect_rgb(src, target, alpha)
    assert out.shape == (32, 32, 4) and np.array_equal(out[..., 3:], alpha)
    m = alpha[..., 0] > 0.5
    assert np.mean((out[..., :3] - target) ** 2) < np.mean((src - target) ** 2)
    for c in range(3):
        assert abs(out[..., c][m].std(ddof=1) - target[..., c][m].std(ddof=1)) < 0.02
    same = D.recorrect_rgb(target, target, alpha)                   # nothing to correct: the source is kept as it is
    assert np.array_equal(same[..., :3], target)


class _Echo:
    """stands in for r3g.delight.InstructPix2Pix on the CPU: returns its input image"""

    def __call__(self, x, prompt_embeds, num_inference_steps, generator):
        assert x.shape == (1, 3, 512, 512) and float(x.min()) >= -1.0 and float(x.max()) <= 1.0
        assert num_inference_steps == 50 and generator.initial_seed() == 42
        return x


def test_light_shadow_remover_bookkeeping_around_the_model():
    from PIL import Image
    rng = np.random.default_rng(2)
    arr = np.zeros((256, 256, 4), np.uint8)
    arr[..., :3] = rng.integers(0, 255, (256, 256, 3))
    arr[64:192, 64:192, 3] = 255
    r = D.Light_Shadow_Remover(model=_Echo(), prompt_embeds=torch.zeros(1, 77, 1024))
    out = r(Image.fromarray(arr, "RGBA"))
    assert out.mode == "RGB" and out.size == (512, 512)
    o = np.asarray(out)
    assert np.all(o[:100, :100] == 255)                              # outside the object: white
    rgb, target, alpha = r.prepare(Image.fromarray(arr, "RGBA"))
    assert alpha.shape == (512, 512, 1) and set(np.unique(alpha)) <= {0.0, 1.0} or alpha.max() <= 1.0
    inside = alpha[..., 0] == 1.0
    assert np.all(np.abs(o[inside].astype(int) - rgb[inside].astype(int)) <= 1)      # echo model: the object keeps its colours
    with pytest.raises(ValueError):
        D.Light_Shadow_Remover(model=None)
    with pytest.raises(ValueError):
        D.Light_Shadow_Remover(model=_Echo())                        # no embedding of the empty prompt


def test_oracle_pipeline_runs_and_is_deterministic_in_its_noise():
    from oracle import aekl_torch as A, unet_torch as U
    ucfg = dict(U.small_config(), in_channels=8, out_channels=4)
    unet = U.load(ucfg, U.synthetic_state_dict(ucfg, seed=3, full=True), full=True)
    vae = A.build(A.small_config(), seed=5)
    g = torch.Generator().manual_seed(1)
    img = torch.rand(1, 3, 32, 32, generator=g) * 2 - 1
    pe = torch.randn(1, ucfg["ctx_tokens"], ucfg["cross_attention_dim"], generator=g)
    lat = torch.randn(1, 4, 8, 8, generator=g)
    noise = [torch.randn(1, 4, 8, 8, generator=g) for _ in range(3)]
    a = P.instruct_pix2pix(unet, vae, pe, img, 3, lat, noise)
    b = P.instruct_pix2pix(unet, vae, pe, img, 3, lat, noise)
    assert a.shape == (1, 3, 32, 32) and torch.equal(a, b) and torch.isfinite(a).all()
    z = P.instruct_pix2pix(unet, vae, pe, img, 3, lat, noise, output="latent")
    assert z.shape == (1, 4, 8, 8) and torch.allclose(vae.decode(z / 0.18215), a)
