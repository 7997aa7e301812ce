"""The drop-in stage script end to end on the MI355X: crops in, `<out>/<stem>/<stem>.glb` out, through the hy3dgen
mirror, libr3g.so, the cleaners and the GLB writer (seeded synthetic full-dims weights -- the mini-dims synthetic field happens to be negative everywhere --, and
its 2-step field too --, the reference's 50 denoising steps, 65^3 grid)."""
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
        assert np.isfinite(m.vertices).all() and np.abs(m.vertices).max() <= 1.02


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
