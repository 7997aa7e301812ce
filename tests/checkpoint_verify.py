"""TEST INFRASTRUCTURE (imports oracle/): one-command verification of a Hunyuan3D-2 snapshot against both implementations.

The DiT / VAE / geo-decoder parity of this repository is PARITY-UNPINNED (oracle/hy3d_torch.py header): `hy3dgen` and its weights
are not in the container, so the HIP path is checked against a restatement on synthetic checkpoints.  The day a real snapshot is
reachable, this is the command that decides (reference call sites src/2d_to_3d_models/run.py:122-124, 204-206):

    python tools/verify_checkpoint.py <snapshot dir> [--subfolder hunyuan3d-dit-v2-0] [--variant fp16] [--steps 4] [--full]

  1. KEYS     the files load through the PRODUCT's loader (r3g/weights.py); the names are diffed against the oracle's modules
              (strict=True: any missing / unexpected key is printed and fails the run -- a restatement that misremembers upstream's
              module tree shows up here first) and against the product's own table (r3g.weights.param_shapes: names AND shapes).
  2. BLOCKS   per-block branch contributions of the DiT, HIP vs the oracle block applied to the HIP path's own stream (tolerance
              parity_support.TOL["block_delta"]), and max |residual stream| per block -- the figure that decides whether the fp16
              residual stream (the default, csrc/model.cpp g_dit_resid_f16) has headroom under REAL weight scales (limit 65504).
  3. SAMPLING `--steps` Euler steps x CFG from the reference's seed: latents HIP (fp16 stream AND fp32 stream) vs oracle fp32,
              tolerance TOL["flow_sample"] (TOL["flow_sample_50"] from 50 steps); the fp16-guard counter must stay 0.
  4. DECODE   shape-VAE output and grid logits on a slab of the grid, TOL["vae_latents"] / TOL["grid_logits"].
  5. TEXTURE  when <snapshot>/hunyuan3d-delight-v2-0 and hunyuan3d-paint-v2-0 exist: both loaders in strict mode (every tensor the HIP
              UNets / VAEs need must be found under its diffusers name with its shape; anything else is listed and fails the run).
`--full` = 50 steps and every block (hours of host time for the fp32 oracle at full depth); the default is sized for minutes.
Exit code 0 = every check inside its tolerance; the JSON report goes to stdout (last line).  tests/test_verify_checkpoint_*.py run
it on a synthetic snapshot written in upstream's layout.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-re-gen_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP16_MAX = 65504.0


def key_report(cfg, sd):
    """names (and shapes) of the checkpoint against the oracle's module tree and the product's table"""
    from oracle import hy3d_torch as H
    from r3g import weights as W
    pipe = H.ShapePipeline(cfg)
    want = {}
    for prefix, mod in (("model.", pipe.model), ("vae.", pipe.vae), ("conditioner.", pipe.conditioner)):
        for k, v in mod.state_dict().items():
            want[prefix + k] = tuple(v.shape)
    have = {k: tuple(v.shape) for k, v in sd.items()}
    missing = sorted(set(want) - set(have))
    unexpected = sorted(set(have) - set(want))
    shape_diff = sorted(k for k in set(want) & set(have) if want[k] != have[k])
    table = {k: tuple(v) for k, v in W.param_shapes(cfg).items()}
    rep = {"oracle_missing": missing, "oracle_unexpected": unexpected,
           "oracle_shape_mismatch": [(k, have[k], want[k]) for k in shape_diff],
           "product_table_missing": sorted(set(table) - set(have)), "product_table_unexpected": sorted(set(have) - set(table)),
           "product_shape_mismatch": [(k, have[k], table[k]) for k in sorted(set(table) & set(have)) if table[k] != have[k]],
           "tensors": len(have), "dtypes": sorted({str(v.dtype) for v in sd.values()})}
    rep["ok"] = not (missing or unexpected or shape_diff or rep["product_table_missing"] or rep["product_table_unexpected"] or
                     rep["product_shape_mismatch"])
    return pipe, rep


def verify_shape(path, subfolder, variant, steps, blocks, slab, keys_only, log):
    import torch
    import yaml
    import parity_support as P
    from hy3dgen.shapegen.pipelines import config_from_yaml
    from oracle import hy3d_torch as H
    from r3g import weights as W
    rep = {"snapshot": path, "subfolder": subfolder}
    sub = os.path.join(os.path.expanduser(path), subfolder)
    with open(os.path.join(sub, "config.yaml")) as f:
        cfg = config_from_yaml(yaml.safe_load(f))
    sd = W.load_safetensors_dir(sub, variant)
    sd = {k: v.float() if torch.is_floating_point(v) else v for k, v in sd.items()}
    oracle, rep["keys"] = key_report(cfg, sd)
    log("keys: %d tensors, dtypes %s -> %s" % (rep["keys"]["tensors"], rep["keys"]["dtypes"], "OK" if rep["keys"]["ok"] else "MISMATCH"))
    for name in ("oracle_missing", "oracle_unexpected", "product_table_missing", "product_table_unexpected"):
        for k in rep["keys"][name][:40]:
            log("  %s: %s" % (name, k))
    if not rep["keys"]["ok"] or keys_only:
        rep["ok"] = rep["keys"]["ok"]
        return rep
    # what rounding the checkpoint's matrices to bf16 (the MFMA operand type) costs, then both sides on the SAME rounded tensors
    rounded = P.bf16_round_matrices(sd)
    num = sum(float(((rounded[k] - sd[k]).double() ** 2).sum()) for k in sd if torch.is_floating_point(sd[k]) and sd[k].ndim >= 2)
    den = sum(float((sd[k].double() ** 2).sum()) for k in sd if torch.is_floating_point(sd[k]) and sd[k].ndim >= 2)
    rep["bf16_rounding_of_the_matrices_rel_l2"] = (num / max(den, 1e-300)) ** 0.5
    H.load_state_dict(oracle, rounded)          # strict=True
    if not torch.cuda.is_available():
        raise SystemExit("verify_checkpoint: the numeric checks need the MI355X (no CPU path in the product); --keys-only runs anywhere")
    from r3g import ffi
    from r3g import model as M
    L = ffi.lib()
    gpu = M.ShapeModel(cfg, rounded, 0, grid_chunk=4096)
    ok = True
    # ---- 2. per-block deltas + residual magnitudes (fp32 stream: dit_forward is the plain path)
    x, t, cond = P.dit_inputs(cfg, 3)
    m = oracle.model
    nd, ns = len(m.double_blocks), len(m.single_blocks)
    todo = list(range(nd + ns)) if blocks < 0 else sorted(set(list(range(min(blocks, nd))) + [nd + i for i in range(min(blocks, ns))]))
    _, _, vec = P.dit_prologue(m, x, t, cond)
    n_cond = cond.shape[1]
    rep["blocks"] = []
    for k in todo:
        gpu.dit_forward(x, t, cond, min(k, nd), max(0, k - nd))
        s0 = gpu.dit_stream(x.shape[0]).cpu()
        gpu.dit_forward(x, t, cond, min(k + 1, nd), max(0, k + 1 - nd))
        s1 = gpu.dit_stream(x.shape[0]).cpu()
        ref = P.dit_block_apply(m, k, s0, vec, n_cond) - s0
        err = P.block_delta_error(s1 - s0, ref, n_cond, k < nd)
        amax = float(s1.abs().max())
        rep["blocks"].append({"block": k, "kind": "double" if k < nd else "single", "delta_rel_l2": err, "max_abs_residual": amax,
                              "fp16_headroom": FP16_MAX / max(amax, 1e-30)})
        log("block %2d (%s): delta %.3e [%.1e]   max |stream| %.4g (fp16 headroom x%.1f)" % (
            k, "double" if k < nd else "single", err, P.TOL["block_delta"], amax, FP16_MAX / max(amax, 1e-30)))
        ok &= err <= P.TOL["block_delta"] and torch.isfinite(s1).all().item()
    rep["max_abs_residual"] = max(b["max_abs_residual"] for b in rep["blocks"])
    rep["fp16_stream_has_headroom"] = rep["max_abs_residual"] * 8 < FP16_MAX      # (a factor 8 in hand: other inputs, later steps)
    # ---- 3. sampling: fp16 stream (default) and fp32 stream against the oracle
    tol = P.TOL["flow_sample_50"] if steps >= 50 else P.TOL["flow_sample"]
    g = torch.Generator().manual_seed(1234567)                   # src/config.yaml:29
    lat0 = torch.randn(1, cfg["vae"]["num_latents"], cfg["dit"]["in_channels"], generator=g)
    cond2 = torch.stack([cond[0], torch.zeros_like(cond[0])]).to(torch.bfloat16)
    t0 = time.time()
    want = oracle.sample(cond2.float(), lat0.clone(), steps, 5.0)[0]
    log("oracle: %d Euler steps x CFG in %.0f s" % (steps, time.time() - t0))
    rep["sampling"] = {"steps": steps, "tolerance": tol}
    before = ffi.counter("dit_f16_fallbacks")
    for name, f16 in (("fp16_stream", 1), ("fp32_stream", 0)):
        ffi.check(L.r3g_set_option(b"dit_resid_f16", f16))
        try:
            got = gpu.flow_sample(lat0[0].clone(), cond2, steps, 5.0).cpu()
        finally:
            ffi.check(L.r3g_set_option(b"dit_resid_f16", 1))
        e = P.rel_l2(got, want)
        rep["sampling"][name] = e
        log("latents after %d steps, %s: %.3e [%.1e]" % (steps, name, e, tol))
        ok &= e <= tol and torch.isfinite(got).all().item()
    rep["sampling"]["dit_f16_fallbacks"] = ffi.counter("dit_f16_fallbacks") - before
    ok &= rep["sampling"]["dit_f16_fallbacks"] == 0
    # ---- 4. VAE + a slab of the grid
    z = gpu.vae_decode(got.to("cuda"), return_z=True).cpu()
    with torch.no_grad():
        zr = oracle.vae(got[None] if got.ndim == 2 else got)
    e = P.rel_l2(z, zr[0])
    rep["vae_rel_l2"] = e
    log("shape-VAE output: %.3e [%.1e]" % (e, P.TOL["vae_latents"]))
    ok &= e <= P.TOL["vae_latents"]
    R = 64
    n = R + 1
    start, count = (n // 2) * n * n, min(slab, n * n * (n - n // 2))
    out = torch.zeros((n, n, n), dtype=torch.float32, device="cuda")
    gpu.grid_query(1.01, R, out=out, start=start, count=count)
    pts = torch.from_numpy(H.dense_grid_points(1.01, R)[start:start + count])[None]
    with torch.no_grad():
        lr = oracle.vae.geo_decoder(queries=pts, latents=zr)[0, :, 0]
    lg = out.reshape(-1)[start:start + count].cpu()
    e = float((lg - lr).abs().max() / lr.abs().max())
    rep["grid_logits_max_err_over_max"] = e
    log("grid logits on %d points: %.3e [%.1e]" % (count, e, P.TOL["grid_logits"]))
    ok &= e <= P.TOL["grid_logits"]
    rep["ok"] = bool(ok)
    return rep


def verify_texture(path, log):
    """both texture loaders, strict, when their sub-folders exist (src/2d_to_3d_models/run.py:126-128, 207-209)"""
    rep = {}
    root = os.path.expanduser(path)
    present = [d for d in ("hunyuan3d-delight-v2-0", "hunyuan3d-paint-v2-0") if os.path.isdir(os.path.join(root, d))]
    rep["folders"] = present
    if not present:
        log("texture: no hunyuan3d-delight-v2-0 / hunyuan3d-paint-v2-0 folder under the snapshot: skipped")
        rep["ok"] = True
        rep["skipped"] = True
        return rep
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("verify_checkpoint: the texture checks need the MI355X")
    from hy3dgen.texgen import Hunyuan3DPaintPipeline
    try:
        pipe = Hunyuan3DPaintPipeline.from_pretrained(root, strict=True)
        rep["loaded"] = True
        rep["problems"] = list(getattr(pipe, "load_problems", []) or [])
        rep["ok"] = not rep["problems"] and bool(getattr(pipe, "implemented", True))
        log("texture: both models loaded strictly: %s" % ("OK" if rep["ok"] else rep["problems"]))
    except Exception as e:
        rep["loaded"] = False
        rep["error"] = "%s: %s" % (type(e).__name__, e)
        rep["ok"] = False
        log("texture: strict load FAILED: %s" % rep["error"])
    return rep


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("snapshot")
    ap.add_argument("--subfolder", default="hunyuan3d-dit-v2-0")
    ap.add_argument("--variant", default="fp16")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--blocks", type=int, default=2, help="double and single blocks checked one by one from the front (-1: all)")
    ap.add_argument("--slab", type=int, default=8192, help="grid points of the 65^3 grid whose logits are compared")
    ap.add_argument("--full", action="store_true", help="50 steps, every block")
    ap.add_argument("--keys-only", action="store_true", help="step 1 only (runs without a GPU)")
    ap.add_argument("--no-texture", action="store_true")
    a = ap.parse_args(argv)
    if a.full:
        a.steps, a.blocks = 50, -1

    def log(msg):
        print("[verify] " + msg, file=sys.stderr, flush=True)
    rep = {"shape": verify_shape(a.snapshot, a.subfolder, a.variant, a.steps, a.blocks, a.slab, a.keys_only, log)}
    if not a.no_texture and not a.keys_only:
        rep["texture"] = verify_texture(a.snapshot, log)
    rep["ok"] = all(v.get("ok", False) for v in rep.values() if isinstance(v, dict))
    print(json.dumps(rep))
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
