#!/usr/bin/env python3
"""tools/make_cfg1_golden.py -- the fp32 oracle on BASELINE.json configs[1] at FULL depth, written as a fixture.

    python tools/make_cfg1_golden.py [--work /tmp/cfg1_golden_work] [--threads 8]

Runs oracle/hy3d_torch.py (PyTorch CPU fp32, test infrastructure) on exactly what the headline bench line computes per
object -- full Hunyuan3D-2 dims (16 double + 32 single DiT blocks, 3072 latent + 1370 context tokens, DINOv2-g 40 layers,
16 VAE layers), 50 Euler steps x CFG 2 at guidance 5, the reference's seed 1234567 (src/config.yaml:29; call
src/2d_to_3d_models/run.py:77-84 with src/config.yaml:165-169) -- on the bench's synthetic crop 0 and the seeded unit-scale
checkpoint of the parity tests (oracle.synthetic_state_dict(full_config(), seed=CKPT_SEED), matrices rounded to bf16 once),
and writes tests/golden/cfg1_full_depth.npz:

    cond_rows    every 10th conditioner token                      f32 [137, 1536]
    lat_XX       latents after Euler step XX = 10, 20, 30, 40, 50  f32 [3072, 64]
    vae_rows     every 24th row of the shape-VAE output            f32 [128, 1024]
    logit_start, logits : 4096 consecutive grid points of the 257^3 grid (centre of the volume)  f32 [4096]

Hours of host time on 8 cores (about 900 TFLOP in fp32), once; every Euler step is checkpointed under --work, a restarted run
resumes.  tests/test_cfg1_golden_gpu.py compares the HIP path with the file (50-step latents <= 3e-2 rel-L2,
grid logits <= 1e-2 of the largest |logit| here).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

CKPT_SEED = 11
NOISE_SEED = 1234567          # src/config.yaml:29
STEPS, GUIDANCE, R = 50, 5.0, 256
KEEP = (10, 20, 30, 40, 50)
LOGIT_COUNT = 4096


def logit_start(R):
    n = R + 1
    return (n // 2) * n * n + (n // 2) * n          # the row through the centre of the volume and the rows after it


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/cfg1_golden_work")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "cfg1_full_depth.npz"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    os.makedirs(a.work, exist_ok=True)
    from bench import synthetic_crop
    from oracle import hy3d_torch as H
    from parity_support import bf16_round_matrices

    def log(msg):
        print("[%s] %s" % (time.strftime("%H:%M:%S"), msg), flush=True)

    cfg = H.full_config()
    t0 = time.time()
    sd = bf16_round_matrices(H.synthetic_state_dict(cfg, seed=CKPT_SEED))
    pipe = H.load_state_dict(H.ShapePipeline(cfg), sd)
    del sd
    log("checkpoint built in %.0f s" % (time.time() - t0))

    cond_f = os.path.join(a.work, "cond2.pt")
    if os.path.exists(cond_f):
        cond2 = torch.load(cond_f)
    else:
        t0 = time.time()
        img, _ = H.preprocess_image(synthetic_crop(0), **cfg["proc"])
        cond2 = pipe.encode_cond(img)
        torch.save(cond2, cond_f)
        log("conditioner in %.0f s" % (time.time() - t0))

    lat = H.prepare_latents((1,) + pipe.vae.latent_shape, torch.manual_seed(NOISE_SEED))
    first = 0
    for i in range(STEPS - 1, -1, -1):
        f = os.path.join(a.work, "lat_%02d.pt" % (i + 1))
        if os.path.exists(f):
            lat, first = torch.load(f), i + 1
            break
    log("resuming after step %d" % first)
    tick = [time.time()]

    def cb(i, x):
        torch.save(x, os.path.join(a.work, "lat_%02d.pt" % (i + 1)))
        log("step %d done in %.0f s" % (i + 1, time.time() - tick[0]))
        tick[0] = time.time()
    lat = pipe.sample(cond2, lat, STEPS, GUIDANCE, first_step=first, callback=cb)

    t0 = time.time()
    with torch.no_grad():
        z = pipe.vae(lat / pipe.vae.scale_factor)
        start = logit_start(R)
        pts = torch.from_numpy(H.dense_grid_points(cfg["box_v"], R)[start:start + LOGIT_COUNT].copy())
        logits = pipe.vae.geo_decoder(queries=pts[None], latents=z)[0].reshape(-1).float()
    log("VAE + %d grid points in %.0f s" % (LOGIT_COUNT, time.time() - t0))

    out = {"cond_rows": cond2[0, ::10].numpy().astype(np.float32),
           "vae_rows": z[0, ::24].numpy().astype(np.float32),
           "logit_start": np.int64(start), "logits": logits.numpy().astype(np.float32),
           "ckpt_seed": np.int64(CKPT_SEED), "noise_seed": np.int64(NOISE_SEED), "steps": np.int64(STEPS),
           "guidance": np.float64(GUIDANCE), "octree_resolution": np.int64(R)}
    for k in KEEP:
        out["lat_%02d" % k] = torch.load(os.path.join(a.work, "lat_%02d.pt" % k))[0].numpy().astype(np.float32)
    np.savez(a.out, **out)
    log("wrote %s" % a.out)


if __name__ == "__main__":
    main()
