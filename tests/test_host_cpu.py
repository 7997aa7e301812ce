"""CPU tests of the host-side logic: checkpoint layout vs the oracle's modules, weight re-layout, the Trimesh-like
mesh + GLB writer, the cleaners, and the restated image preprocessing (product vs oracle)."""
import numpy as np
import pytest
import torch


def test_checkpoint_layout_matches_restated_modules():
    from oracle import hy3d_torch as H
    from r3g import weights as W
    cfg = H.tiny_config()
    pipe = H.ShapePipeline(cfg)
    ref = {}
    for prefix, mod in (("model.", pipe.model), ("vae.", pipe.vae), ("conditioner.", pipe.conditioner)):
        for k, p in mod.state_dict().items():
            ref[prefix + k] = tuple(p.shape)
    mine = {k: tuple(v) for k, v in W.param_shapes(cfg).items()}
    assert mine == ref
    H.load_state_dict(H.ShapePipeline(cfg), W.synthetic_state_dict(cfg, 0, "cpu"))  # strict=True inside


def test_full_config_parameter_count_is_the_published_one():
    from hy3dgen.shapegen.pipelines import builtin_config
    from r3g import weights as W
    shapes = W.param_shapes(builtin_config("full"))
    n_dit = sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("model."))
    assert 1.05e9 < n_dit < 1.15e9       # "1.1B" DiT
    n_dino = sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("conditioner."))
    assert 1.1e9 < n_dino < 1.2e9        # dinov2-giant


def test_weight_relayout():
    from oracle import hy3d_torch as H
    from r3g import model as M
    cfg = H.tiny_config()
    sd = H.synthetic_state_dict(cfg, 1)
    w, scalars = M.prepare_weights(sd, "cpu")
    k = "model.latent_in.weight"
    assert w[k][0].shape == (128, 64) and w[k][1] == 1                  # K 16 -> 64, zero padded
    assert torch.equal(w[k][0][:, 16:].float(), torch.zeros(128, 48))
    q = "conditioner.main_image_encoder.model.encoder.layer.0.attention.attention"
    fused = w[q + ".qkv.weight"][0].float()
    assert torch.equal(fused[192:384], sd[q + ".key.weight"].to(torch.bfloat16).float())
    assert "vae.geo_decoder.output_proj.bias" in scalars
    c = M.make_config(cfg)
    assert c.cond_ffn_hidden == 512 and c.dit_mlp_hidden == 512


def test_mesh_glb_roundtrip(tmp_path):
    from r3g.mesh import Mesh, load_glb
    rng = np.random.default_rng(0)
    v = rng.standard_normal((50, 3)).astype(np.float32)
    f = rng.integers(0, 50, (80, 3))
    m = Mesh(v, f)
    p = tmp_path / "a" / "a.glb"
    p.parent.mkdir()
    m.export(str(p))
    back = load_glb(str(p))
    assert np.array_equal(back.vertices.astype(np.float32), v) and np.array_equal(back.faces, f)
    data = p.read_bytes()
    assert data[:4] == b"glTF" and len(data) % 4 == 0


def test_cleaner_oracle():
    """the numpy restatement of the cleaners (oracle/mesh_clean.py) behaves as the three classes are meant to"""
    from oracle import mc, mesh_clean
    from mc_volumes import golden_volume
    vol, level = golden_volume("A")
    v, f = mc.hy3d_mesh(vol, level)
    v = v.astype(np.float32)
    # add a floater: one far-away triangle, and a degenerate face
    v2 = np.concatenate([v, np.array([[5, 5, 5], [5, 5, 5.1], [5, 5.1, 5]], np.float32)])
    f2 = np.concatenate([f, [[len(v), len(v) + 1, len(v) + 2]], [[0, 0, 1]]])
    a, b = mesh_clean.remove_floaters(v2, f2)
    a, b = mesh_clean.remove_degenerate(a, b)
    assert np.array_equal(a, v) and np.array_equal(b, f)
    rv, rf = mesh_clean.reduce_faces(v, f, max_faces=3000)
    assert 0 < len(rf) <= 3000 and rf.max() == len(rv) - 1
    assert abs(np.linalg.norm(rv, axis=1).mean() - np.linalg.norm(v, axis=1).mean()) < 0.05
    same_v, same_f = mesh_clean.reduce_faces(v, f, max_faces=len(f))
    assert np.array_equal(same_v, v) and np.array_equal(same_f, f)


def test_mesh_host_device_views():
    from r3g.mesh import Mesh
    m = Mesh(np.zeros((3, 3)), [[0, 1, 2]])
    assert m.n_vertices == 3 and m.n_faces == 1 and not m.is_empty
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m.device_buffers()      # the cleaners have no CPU path


def test_preprocess_product_equals_oracle():
    from PIL import Image
    from hy3dgen.shapegen.preprocessors import ImageProcessorV2, conditioner_transform
    from oracle import hy3d_torch as H
    rng = np.random.default_rng(1)
    img = np.zeros((120, 90, 4), np.uint8)
    img[30:100, 20:70, :3] = rng.integers(0, 255, (70, 50, 3))
    img[30:100, 20:70, 3] = rng.integers(1, 255, (70, 50))
    pil = Image.fromarray(img, "RGBA")
    a = ImageProcessorV2(64, 0.15)(pil)
    b_img, b_mask = H.preprocess_image(pil, 64, 0.15)
    assert torch.equal(a["image"], b_img) and torch.equal(a["mask"], b_mask)
    x = conditioner_transform(a["image"], 70)
    y = H.DinoImageEncoder.transform((b_img + 1) / 2, 70, H.DinoImageEncoder.mean, H.DinoImageEncoder.std)
    assert torch.allclose(x, y, atol=1e-6)
    with pytest.raises(ValueError):
        ImageProcessorV2(64, 0.15)(Image.fromarray(np.zeros((8, 8, 4), np.uint8), "RGBA"))


def test_cleaner_oracle_properties():
    """size-independent properties of the restated cleaners on random triangle soups: outputs index densely into
    their vertex arrays, are idempotent, never grow, and the face budget holds"""
    from oracle import mesh_clean
    rng = np.random.default_rng(11)
    for nv, nf in ((30, 200), (400, 3000), (2000, 9000)):
        v = rng.standard_normal((nv, 3)).astype(np.float32)
        f = rng.integers(0, nv, (nf, 3)).astype(np.int32)
        f[::5, 2] = f[::5, 0]
        for fn, args in ((mesh_clean.remove_degenerate, ()), (mesh_clean.remove_floaters, (0.2,)),
                         (mesh_clean.reduce_faces, (nf // 8,))):
            a, b = fn(v, f, *args)
            assert a.dtype == np.float32 and b.dtype == np.int32
            assert len(b) <= nf and len(a) <= nv
            if len(b):
                assert b.min() == 0 and b.max() == len(a) - 1 and len(np.unique(b)) == len(a)
            a2, b2 = fn(a, b, *args)
            assert np.array_equal(a2, a) and np.array_equal(b2, b)
        _, rf = mesh_clean.reduce_faces(v, f, nf // 8)
        assert len(rf) <= nf // 8
        dv, df = mesh_clean.remove_degenerate(v, f)
        assert ((df[:, 0] != df[:, 1]) & (df[:, 1] != df[:, 2]) & (df[:, 0] != df[:, 2])).all()


def _snapshot_doc(cfg):
    """an upstream-style config.yaml document for `cfg` (the keys config_from_yaml reads)"""
    return {"model": {"target": "hy3dgen.shapegen.models.Hunyuan3DDiT", "params": dict(cfg["dit"])},
            "vae": {"target": "hy3dgen.shapegen.models.ShapeVAE", "params": dict(cfg["vae"])},
            "conditioner": {"params": {"main_image_encoder": {"kwargs": {"config": dict(cfg["cond"]),
                                                                           "image_size": cfg["cond"]["image_size"]}}}},
            "scheduler": {"params": dict(cfg["sched"])},
            "image_processor": {"params": dict(cfg["proc"])}}


def test_snapshot_config_and_weight_files(tmp_path):
    """`from_pretrained(<dir>)`'s host half: config.yaml -> cfg, model[.variant].safetensors lookup"""
    import yaml
    from safetensors.torch import save_file
    from hy3dgen.shapegen.pipelines import config_from_yaml
    from oracle import hy3d_torch as H
    from r3g import weights as W
    cfg = H.tiny_config()
    got = config_from_yaml(yaml.safe_load(yaml.safe_dump(_snapshot_doc(cfg))))
    for sec in ("dit", "vae", "cond", "sched", "proc"):
        for k, v in cfg[sec].items():
            if k in got[sec]:
                assert got[sec][k] == v, (sec, k)
    assert got["dit"]["hidden_size"] == 128 and got["vae"]["num_latents"] == 256 and got["cond"]["image_size"] == 70
    sd = {k: v.contiguous() for k, v in W.synthetic_state_dict(cfg, 0, "cpu").items()}
    save_file(sd, str(tmp_path / "model.fp16.safetensors"))
    back = W.load_safetensors_dir(str(tmp_path), "fp16")
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    assert set(W.load_safetensors_dir(str(tmp_path), None)) == set(sd)        # falls back to model.fp16.safetensors
    with pytest.raises(FileNotFoundError):
        W.load_safetensors_dir(str(tmp_path / "nope"), "fp16")
    # upstream's other container: model[.variant].ckpt = a pickle of {"model", "vae", "conditioner"} state dicts
    nested = {}
    for k, v in sd.items():
        part, rest = k.split(".", 1)
        nested.setdefault(part, {})[rest] = v
    assert set(nested) == {"model", "vae", "conditioner"}
    ck = tmp_path / "ck"
    ck.mkdir()
    torch.save(nested, str(ck / "model.fp16.ckpt"))
    for kw in (dict(use_safetensors=False), dict()):              # asked for, or found because no safetensors file is there
        back = W.load_safetensors_dir(str(ck), "fp16", **kw)
        assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    with pytest.raises(FileNotFoundError):                         # use_safetensors=False does not fall back to safetensors
        W.load_safetensors_dir(str(tmp_path), "fp16", use_safetensors=False)
    assert set(W.flatten_ckpt({"state_dict": nested, "global_step": 7})) == set(sd) and W.flatten_ckpt(sd).keys() == sd.keys()
    with pytest.raises(ValueError):
        W.flatten_ckpt({"model": {"step": 3}})


def test_bench_workload_definition():
    """bench.py's synthetic workload: the FLOP budget of configs[1] and the seeded crops"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from hy3dgen.shapegen.pipelines import builtin_config
    f = bench.flops_per_object(builtin_config("full"), 50, 256)
    assert abs(f / 1.498e15 - 1.0) < 2e-3                     # DiT 925.1 T + VAE 1.86 T + grid query 571.4 T
    assert abs(bench.flops_per_object(builtin_config("full"), 1, 256) - (f - 49 * 2 * 9.251e12)) / f < 1e-3
    a, b = bench.synthetic_crop(3), bench.synthetic_crop(3)
    assert a.mode == "RGBA" and a.size == (512, 512) and a.tobytes() == b.tobytes()      # seeded, reproducible
    assert bench.synthetic_crop(4).tobytes() != a.tobytes()
    for i in range(6):
        al = np.asarray(bench.synthetic_crop(i))[..., 3]
        cov = (al > 0).mean()
        assert 0.25 <= cov <= 0.85, (i, cov)
        rgb = np.asarray(bench.synthetic_crop(i))[..., :3]
        assert (rgb[al == 0] == 255).all()                     # white where transparent


def test_committed_bench_line_has_the_contract_fields():
    """the last bench line measured on the MI355X box (profiles/r*_bench.json) against the driver's contract: one JSON object,
    the named keys with the right types, a roofline and a cpu_baseline object; frac = achieved / peak"""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_bench.json")))
    assert files
    lines = [ln for ln in open(files[-1]).read().split("\n") if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 1e-6       # one object per step on one GPU
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert d["value"] / c["value"] > 100        # the device path is not the oracle in disguise
    if os.path.basename(files[-1]) >= "r03_bench.json":
        # round 3: the timed object's mesh was checked against the C marching-cubes oracle inside the run, the HBM-bound
        # kernel family has its own roofline entry, the counter record says whether it belongs to the library that ran
        assert d["mc_parity"] == "exact"
        m = d["roofline_mc"]
        assert m["bound"] == "hbm" and m["unit"] == "GB/s" and abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-9 and 0 < m["frac"] < 1
        assert m["object_like_field"]["frac"] > m["frac"]
        assert r["traffic_stale"] in (True, False) and d["config"]["objects_per_launch"] >= 1
        assert d["steps"] % 1 == 0 and d["config"]["objects_total"] == d["steps"]


def test_profiles_index_names_files_that_exist_and_the_headline_line_is_not_stale():
    """profiles/README.md's tables are the index the review reads: every `rNN_*` file they name is in the directory, and the newest
    committed bench line was taken with the traffic record of the library it ran (roofline.traffic_stale is False)"""
    import glob
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "profiles", "README.md")).read()
    round5 = text.split("## Round 5", 1)[1].split("## Round 4", 1)[0]
    named = set(re.findall(r"`(r05_[A-Za-z0-9_.]+\.(?:md|json|txt))`", round5)) | {"traffic.json"}
    assert len(named) >= 15
    missing = sorted(n for n in named if not os.path.exists(os.path.join(root, "profiles", n)))
    assert not missing, missing
    files = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_bench.json")))
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    assert d["roofline"]["traffic_stale"] is False
    assert d["roofline"]["traffic_source"].startswith("profiles/traffic.json")

