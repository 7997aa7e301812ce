"""The drop-in stage script end to end on the MI355X: crops in, `<out>/<stem>/<stem>.glb` out, through the hy3dgen
mirror, libr3g.so, the cleaners and the GLB writer (seeded synthetic weights at unit scale, the reference's 50 denoising
steps, 65^3 grid), plus BASELINE.json configs[0] (mini dims, 4 steps, 64^3 grid) against the CPU oracle, which is timed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage_script_writes_glbs(tmp_path):
    sys.path.insert(0, ROOT)
    from bench import synthetic_crop
    from r3g.mesh import load_glb
    inp, out = tmp_path / "prepped", tmp_path / "out"
    inp.mkdir()
    for i in range(2):
        synthetic_crop(i).save(inp / ("obj__(%d, %d).png" % (i, i)))
    synthetic_crop(5).save(inp / "floor__(1, 1).png")     # must be skipped
    cfg = {"mini": False, "num_inf_steps_hy": 50, "octree_resolution_hy": 64, "num_chunks_hy": 16000, "seed": 1234567,
           "remesh": False, "input_folder_hy": str(inp), "output_folder_hy": str(out), "use_banana": False,
           "prepped_for_hunyuan": str(tmp_path / "unused"), "jobs_per_gpu": 1, "use_all_available_cuda": False,
           "r3g_weights": "synthetic:{model}"}
    cfgp = tmp_path / "config.yaml"
    cfgp.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "3d-re-gen_amd", "stage", "run.py"), "--config", str(cfgp)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert sorted(os.listdir(out)) == ["obj__(0, 0)", "obj__(1, 1)"]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["ok"] == 2
    for stem in os.listdir(out):
        m = load_glb(str(out / stem / (stem + ".glb")))
        assert len(m.faces) > 0 and len(m.faces) <= 40000          # FaceReducer bound
        from gltf_validate import validate_glb                     # independent glTF 2.0 checks (tests/gltf_validate.py)
        got = validate_glb((out / stem / (stem + ".glb")).read_bytes())
        assert np.array_equal(got["indices"].astype(np.int64), m.faces) and len(got["positions"]) == len(m.vertices)
        assert got["image"] is not None and "TEXCOORD_0" in got["attributes"]      # texgen baked a base-colour texture
        assert np.isfinite(m.vertices).all() and np.abs(m.vertices).max() <= 1.02
    assert rep["textured"] is True and "input view only" in rep["texture_source"]


def test_configs1_literally_eight_crops_full_model(tmp_path):
    """BASELINE.json configs[1] as written: 1 scene / 8 object crops, Hunyuan3D-2 base dims at FULL depth (16 + 32 blocks,
    DINOv2-g 40 layers, VAE 16 layers), 50 steps x CFG 2, 256^3 octree grid, through the drop-in stage script (reference
    call src/2d_to_3d_models/run.py:77-84 with src/config.yaml's values); every GLB is checked.  The texture stage
    (run.py:97, built at :126-128) runs upstream's WHOLE flow here: the two diffusion models are loaded by the stage's default
    factory from checkpoint sub-folders (written from the oracle's small random models: there are no real weights), so the
    baked views are the multiview model's, not the input crop."""
    sys.path.insert(0, ROOT)
    from bench import synthetic_crop
    from r3g.mesh import load_glb
    from gltf_validate import validate_glb
    from tex_ckpt_support import write_checkpoints
    inp, out = tmp_path / "prepped", tmp_path / "out"
    inp.mkdir()
    write_checkpoints(str(tmp_path / "texture_weights"), seed=3)
    for i in range(8):
        synthetic_crop(i).save(inp / ("obj__(%d, %d).png" % (i, 10 * i)))
    cfg = {"mini": False, "num_inf_steps_hy": 50, "octree_resolution_hy": 256, "num_chunks_hy": 16000, "seed": 1234567,
           "remesh": False, "input_folder_hy": str(tmp_path / "unused"), "output_folder_hy": str(out), "use_banana": True,
           "prepped_for_hunyuan": str(inp), "jobs_per_gpu": 1, "use_all_available_cuda": False,
           "r3g_weights": "synthetic:{model}", "r3g_texture_weights": str(tmp_path / "texture_weights"),
           "r3g_require_textures": True}
    cfgp = tmp_path / "config.yaml"
    cfgp.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    import time
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "3d-re-gen_amd", "stage", "run.py"), "--config", str(cfgp)],
                       capture_output=True, text=True, timeout=1200, env=env)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert sorted(os.listdir(out)) == sorted("obj__(%d, %d)" % (i, 10 * i) for i in range(8))
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["objects"] == 8 and rep["ok"] == 8 and rep["failed"] == []
    assert rep["textured"] is True and "multiview diffusion model" in rep["texture_source"] and "delighted" in rep["texture_source"]
    distinct = set()
    for stem in os.listdir(out):
        data = (out / stem / (stem + ".glb")).read_bytes()
        got = validate_glb(data)
        m = load_glb(str(out / stem / (stem + ".glb")))
        assert 1000 < len(m.faces) <= 40000                                      # FaceReducer's bound, and a real surface
        assert np.array_equal(got["indices"].astype(np.int64), m.faces) and len(got["positions"]) == len(m.vertices)
        assert int(m.faces.max()) < len(m.vertices) and int(m.faces.min()) >= 0
        assert np.isfinite(m.vertices).all() and np.abs(m.vertices).max() <= 1.02   # inside upstream's box_v = 1.01 box
        assert got["image"] is not None and "TEXCOORD_0" in got["attributes"]
        distinct.add(hash(m.vertices.tobytes()))
    assert len(distinct) == 8                                    # eight different crops gave eight different meshes
    from parity_support import report
    report("configs[1] literal: stage script wall seconds for 8 crops (incl. model load)", wall, 1e9)
    report("configs[1] literal: mean seconds per object inside the stage", float(np.mean(rep["seconds"])), 1e9)
    report("configs[1] literal:   of it the cleaners (1.2 M-face noise meshes -> 40 000 faces)", rep["mean_seconds_cleaners"], 1e9)
    report("configs[1] literal:   of it the texture flow (both diffusion models, CI dims, 50 + 30 x 2 evaluations)",
           rep["mean_seconds_texture"], 1e9)
    report("configs[1] literal:   shape model + cleaners = what round 3 reported as the stage's seconds per object",
           float(np.mean(rep["seconds"])) - rep["mean_seconds_texture"], 1e9)


def test_octree_resolution_512_end_to_end():
    """SURVEY config 4's grid size (513^3 = 135 M points, a 15 M-vertex mesh) through grid query, marching cubes and the
    cleaners, on tiny model dims so that it takes seconds; marching cubes still bit-exact against the oracle."""
    import torch
    from PIL import Image
    from hy3dgen.shapegen import DegenerateFaceRemover, FaceReducer, FloaterRemover, Hunyuan3DDiTFlowMatchingPipeline
    from oracle import hy3d_torch as H, mc as omc
    cfg = H.tiny_config()
    sd = {k: (t.to(torch.bfloat16).float() if t.ndim >= 2 else t) for k, t in H.synthetic_state_dict(cfg, seed=5).items()}
    pipe = Hunyuan3DDiTFlowMatchingPipeline(cfg, sd, "cuda:0")
    rng = np.random.default_rng(0)
    arr = np.zeros((80, 80, 4), np.uint8)
    arr[20:60, 15:65, :3] = rng.integers(0, 255, (40, 50, 3))
    arr[20:60, 15:65, 3] = 255
    mesh = pipe(image=Image.fromarray(arr, "RGBA"), num_inference_steps=3, octree_resolution=512,
                generator=torch.manual_seed(1234567))[0]
    grid = pipe.last_grid.cpu().numpy()
    assert grid.shape == (513, 513, 513)
    ov, of = omc.hy3d_mesh(grid, 0.0, 1.01, 512)
    assert mesh.n_faces == len(of) > 1000000
    assert np.array_equal(mesh.faces, of.astype(np.int64))
    assert np.array_equal(mesh.vertices.astype(np.float32).view(np.uint32), ov.view(np.uint32))
    small = FaceReducer()(DegenerateFaceRemover()(FloaterRemover()(mesh)))
    assert 0 < small.n_faces <= 40000


def test_config1_mini_dims_against_the_timed_cpu_oracle():
    """BASELINE.json configs[0]: one crop, Hunyuan3D-2-mini dims (8 + 16 blocks, 512 latents), 4 flow-matching steps, 64^3
    octree grid.  The CPU PyTorch oracle runs the whole configuration (its wall time is recorded: the directly timed CPU
    baseline BASELINE.md section 4 asks for), the MI355X path runs it through the hy3dgen mirror; grids agree within the
    stated tolerance and the mesh is exactly the marching-cubes oracle's mesh of the GPU's own grid."""
    import time
    import torch
    sys.path.insert(0, ROOT)
    from bench import synthetic_crop
    from hy3dgen.shapegen import Hunyuan3DDiTFlowMatchingPipeline
    from oracle import hy3d_torch as H, mc as omc
    from parity_support import bf16_round_matrices, report
    cfg = H.mini_config()
    sd = bf16_round_matrices(H.synthetic_state_dict(cfg, seed=2))
    crop = synthetic_crop(0)
    oracle = H.load_state_dict(H.ShapePipeline(cfg), sd)
    t0 = time.perf_counter()
    _, grid_ref = oracle(crop, num_inference_steps=4, octree_resolution=64, num_chunks=16000,
                         generator=torch.manual_seed(1234567))
    cpu_s = time.perf_counter() - t0
    del oracle
    pipe = Hunyuan3DDiTFlowMatchingPipeline(cfg, sd, "cuda:0")
    del sd
    t0 = time.perf_counter()
    mesh = pipe(image=crop, num_inference_steps=4, octree_resolution=64, num_chunks=16000,
                generator=torch.manual_seed(1234567), output_type="trimesh")[0]
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    assert mesh is not None and mesh.n_faces > 100           # the field has a surface at mini dims
    grid = pipe.last_grid.cpu().numpy()
    ov, of = omc.hy3d_mesh(grid, 0.0, 1.01, 64)
    assert np.array_equal(mesh.faces, of.astype(np.int64))
    assert np.array_equal(mesh.vertices.astype(np.float32).view(np.uint32), ov.view(np.uint32))
    d = np.abs(grid - grid_ref.numpy()).max() / np.abs(grid_ref.numpy()).max()
    report("config 1 (mini, 4 steps, 65^3) grid vs CPU oracle", float(d), 3e-2)
    report("config 1 CPU oracle seconds (fp32, %d threads)" % torch.get_num_threads(), cpu_s, 1e9)
    report("config 1 MI355X seconds (first call, cold)", gpu_s, 1e9)
    assert d <= 3e-2
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "config1.json"), "w") as fh:
        json.dump({"config": "configs[0]: 1 crop, mini dims, 4 steps, 65^3 grid", "cpu_oracle_seconds": cpu_s,
                   "cpu_threads": torch.get_num_threads(), "cpu_dtype": "fp32", "mi355x_seconds_cold": gpu_s,
                   "grid_max_rel_err": float(d), "mesh_faces": int(mesh.n_faces)}, fh)
