"""tests/golden/cfg1_full_depth.npz (BASELINE.json configs[1] at full depth through the fp32 oracle, tools/make_cfg1_golden.py):
what can be said about the fixture without a GPU and without the half hour that made it -- it carries the generating script's
constants, has the shapes tests/test_cfg1_golden_gpu.py reads, and looks like a 50-step sample (finite, O(1), moving)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fixture_matches_its_generator_and_its_reader():
    spec = importlib.util.spec_from_file_location("make_cfg1_golden", os.path.join(ROOT, "tools", "make_cfg1_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_full_depth.npz"))
    assert (int(g["ckpt_seed"]), int(g["noise_seed"]), int(g["steps"]), float(g["guidance"]), int(g["octree_resolution"])) == \
        (gen.CKPT_SEED, gen.NOISE_SEED, gen.STEPS, gen.GUIDANCE, gen.R) == (11, 1234567, 50, 5.0, 256)
    assert int(g["logit_start"]) == gen.logit_start(256) == 128 * 257 * 257 + 128 * 257 and g["logits"].shape == (gen.LOGIT_COUNT,)
    assert g["cond_rows"].shape == (137, 1536) and g["vae_rows"].shape == (128, 1024)       # every 10th of 1370, every 24th of 3072
    lats = [g["lat_%02d" % k] for k in gen.KEEP]
    assert all(x.shape == (3072, 64) and x.dtype == np.float32 and np.isfinite(x).all() for x in lats)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel(lats[-1], lats[0]) > 0.3                              # the sampler moves the latents by O(1) ...
    assert all(0.02 < rel(lats[i + 1], lats[i]) < 2.0 for i in range(4))     # ... in every decade of steps
    assert 0.1 < float(np.abs(lats[-1]).mean()) < 10.0
    assert np.isfinite(g["logits"]).all() and float(np.abs(g["logits"]).max()) > 0.1
    assert (g["logits"] > 0).any() and (g["logits"] < 0).any()       # the slice crosses the surface: MC has work there
