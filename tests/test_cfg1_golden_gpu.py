"""BASELINE.json configs[1] at FULL depth against the fp32 oracle, numerically: the headline configuration end to end.

tests/golden/cfg1_full_depth.npz was written by tools/make_cfg1_golden.py on the CPU box (oracle/hy3d_torch.py in fp32, about
half an hour of 8 cores, once): full Hunyuan3D-2 dims -- DINOv2-g 40 layers, 16 double + 32 single DiT blocks on 3072 + 1370
tokens, 50 Euler steps x CFG 2 at guidance 5, 16 VAE layers -- on the bench's synthetic crop 0, the reference's noise seed
(src/config.yaml:29) and the seeded unit-scale checkpoint of the parity tests (every branch moves its residual stream by O(1)).
It holds the conditioner tokens, the latents after steps 10 / 20 / 30 / 40 / 50, rows of the shape-VAE output and 4096
consecutive grid logits from the centre of the 257^3 grid.  The reference call this pins: src/2d_to_3d_models/run.py:77-84
with src/config.yaml:165-169.

Tolerances: 50-step latents <= 3e-2 rel-L2; grid logits <= 1e-2 of the largest |logit| of the slice.  48 blocks x
49 evaluations x guidance 5 at full width is where bf16 drift would show: the per-decade errors are printed.
"""
import os

import numpy as np
import pytest

from parity_support import TOL, bf16_round_matrices, rel_l2, report

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_full_depth.npz")


@pytest.fixture(scope="module")
def run():
    """the HIP path on the golden's inputs: conditioner tokens, latents after every decade of steps, VAE rows, the logit slice"""
    import torch
    from bench import synthetic_crop
    from hy3dgen.shapegen import Hunyuan3DDiTFlowMatchingPipeline
    from oracle import hy3d_torch as H
    from r3g import ffi
    g = np.load(GOLDEN)
    cfg = H.full_config()
    sd = bf16_round_matrices(H.synthetic_state_dict(cfg, seed=int(g["ckpt_seed"])))
    pipe = Hunyuan3DDiTFlowMatchingPipeline(cfg, sd, "cuda:0")
    del sd
    L = ffi.lib()
    steps, guidance, R = int(g["steps"]), float(g["guidance"]), int(g["octree_resolution"])
    out = {"golden": g}
    with torch.no_grad():
        cond2 = pipe.encode_cond(pipe.prepare_image(synthetic_crop(0))["image"])
        out["cond"] = cond2[0].float().cpu()
        assert float(cond2[1].abs().max()) == 0.0
        lat = pipe.prepare_latents(torch.manual_seed(int(g["noise_seed"])))
        try:
            for lo in range(0, steps, 10):      # the 50-step schedule in five segments, each continuing on the last one's latents
                ffi.check(L.r3g_set_option(b"flow_first_step", lo))
                ffi.check(L.r3g_set_option(b"flow_last_step", lo + 10))
                lat = pipe.model.flow_sample(lat, cond2, steps, guidance, cfg["sched"]["shift"], uncond_uniform=True)
                out["lat_%02d" % (lo + 10)] = lat.clone().cpu()
        finally:
            ffi.check(L.r3g_set_option(b"flow_first_step", 0))
            ffi.check(L.r3g_set_option(b"flow_last_step", -1))
        # the same schedule in one call (what the pipeline does): the segments above are the same launches
        whole = pipe.model.flow_sample(pipe.prepare_latents(torch.manual_seed(int(g["noise_seed"]))), cond2, steps, guidance,
                                       cfg["sched"]["shift"], uncond_uniform=True)
        out["whole_equals_segments"] = bool(torch.equal(whole.cpu(), out["lat_%02d" % steps]))
        z = pipe.model.vae_decode(lat, return_z=True)
        out["vae"] = z.cpu()
        start, count = int(g["logit_start"]), int(g["logits"].shape[0])
        n = R + 1
        grid = torch.zeros((n, n, n), dtype=torch.float32, device="cuda")
        pipe.model.grid_query(cfg["box_v"], R, out=grid, start=start, count=count)
        out["logits"] = grid.reshape(-1)[start:start + count].cpu()
        # the same 50 steps with the DiT's residual stream in fp32 (option dit_resid_f16 = 0: rounds 1-3; the default since
        # round 4 is fp16, the reference's own activation type)
        try:
            ffi.check(L.r3g_set_option(b"dit_resid_f16", 0))
            lat32 = pipe.model.flow_sample(pipe.prepare_latents(torch.manual_seed(int(g["noise_seed"]))), cond2, steps, guidance,
                                           cfg["sched"]["shift"], uncond_uniform=True)
            out["lat_50_f32_stream"] = lat32.clone().cpu()
        finally:
            ffi.check(L.r3g_set_option(b"dit_resid_f16", 1))
    return out


def test_conditioner_tokens_at_full_depth(run):
    import torch
    ref = torch.from_numpy(run["golden"]["cond_rows"])
    err = rel_l2(run["cond"][::10], ref)
    report("configs[1] full depth: DINOv2-g tokens (40 layers)", err, TOL["conditioner"])
    assert err <= TOL["conditioner"]


def test_fifty_step_latents_at_full_depth(run):
    import torch
    assert run["whole_equals_segments"]
    worst = 0.0
    for k in (10, 20, 30, 40, 50):
        ref = torch.from_numpy(run["golden"]["lat_%02d" % k])
        got = run["lat_%02d" % k]
        assert torch.isfinite(got).all()
        err = rel_l2(got, ref)
        report("configs[1] full depth: latents after step %d of 50 (CFG 5)" % k, err, TOL["flow_sample_50"])
        worst = max(worst, err)
    assert worst <= TOL["flow_sample_50"]
    # the sampler moved the latents by O(1): the tolerance is not met by standing still
    first, last = torch.from_numpy(run["golden"]["lat_10"]), torch.from_numpy(run["golden"]["lat_50"])
    assert rel_l2(last, first) > 0.3


def test_vae_and_grid_logits_at_full_depth(run):
    import torch
    ref = torch.from_numpy(run["golden"]["vae_rows"])
    err = rel_l2(run["vae"][::24], ref)
    report("configs[1] full depth: shape-VAE output (16 layers) after the 50-step sample", err, TOL["flow_sample_50"])
    assert err <= TOL["flow_sample_50"]
    lref = torch.from_numpy(run["golden"]["logits"])
    d = float((run["logits"] - lref).abs().max() / lref.abs().max())
    report("configs[1] full depth: 4096 grid logits of the 257^3 grid, max |d| / max |logit|", d, TOL["grid_logits"])
    assert d <= TOL["grid_logits"]


def test_fp32_residual_stream_at_full_depth(run):
    """option dit_resid_f16 = 0: the residual stream of the 48 blocks in fp32 (rounds 1-3) instead of fp16 (the default: what the
    reference's fp16 pipeline holds) -- the same 50 steps against the same fp32 golden, and the two runs against each other"""
    import torch
    ref = torch.from_numpy(run["golden"]["lat_50"])
    got = run["lat_50_f32_stream"]
    assert torch.isfinite(got).all()
    err = rel_l2(got, ref)
    report("configs[1] full depth: latents after 50 steps with an fp32 residual stream (dit_resid_f16 = 0)", err, TOL["flow_sample_50"])
    report("configs[1] full depth:   the fp16-stream run (default) against the fp32-stream run", rel_l2(run["lat_50"], got), TOL["flow_sample_50"])
    assert err <= TOL["flow_sample_50"]
