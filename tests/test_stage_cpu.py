"""Stage contract and the object-parallel multi-process path (gloo, world size 2) on CPU,
with the GPU pipeline replaced by a stand-in factory (the real pipeline needs the MI355X)."""
import importlib.util
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import yaml
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, "3d-re-gen_amd", "stage", "run.py")


def load_stage():
    spec = importlib.util.spec_from_file_location("r3g_stage_run", STAGE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fake_factory(config, device):
    from r3g.mesh import Mesh

    class Shape:
        calls = []

        def __call__(self, image=None, num_inference_steps=None, octree_resolution=None, num_chunks=None,
                     generator=None, output_type=None):
            assert image.mode == "RGBA" and output_type == "trimesh"
            Shape.calls.append((num_inference_steps, octree_resolution, num_chunks, generator.initial_seed()))
            v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
            f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])
            return [Mesh(v, f)]
    return Shape(), (lambda mesh, image=None: mesh), [lambda m: m]


def make_scene(tmp_path, names):
    inp, out = tmp_path / "prepped", tmp_path / "out"
    inp.mkdir()
    out.mkdir()
    (out / "stale").mkdir()
    (out / "stale" / "old.glb").write_bytes(b"x")
    for n in names:
        if n.lower().endswith((".png", ".jpg", ".jpeg")):
            mode = "RGBA" if n.lower().endswith(".png") else "RGB"
            Image.fromarray(np.full((16, 16, 4 if mode == "RGBA" else 3), 200, np.uint8), mode).save(inp / n)
        else:
            (inp / n).write_text("x")
    cfg = {"mini": False, "num_inf_steps_hy": 50, "octree_resolution_hy": 256, "num_chunks_hy": 16000, "seed": 1234567,
           "remesh": False, "input_folder_hy": str(tmp_path / "unused"), "output_folder_hy": str(out), "use_banana": True,
           "prepped_for_hunyuan": str(inp), "jobs_per_gpu": 1, "use_all_available_cuda": False}
    p = tmp_path / "config.yaml"
    p.write_text(yaml.safe_dump(cfg))
    return str(p), inp, out


def test_filesystem_contract(tmp_path, capsys):
    stage = load_stage()
    names = ["chair__(10, 20).png", "lamp__(3, 4).jpg", "Wall__(0, 0).png", "floor.png", "ceiling_light.PNG", "notes.txt"]
    cfg, inp, out = make_scene(tmp_path, names)
    assert stage.main(["--config", cfg], factory=fake_factory) == 0
    made = sorted(os.listdir(out))
    assert made == ["chair__(10, 20)", "lamp__(3, 4)"]            # skip list + non-images; stale content removed
    for stem in made:
        data = (out / stem / (stem + ".glb")).read_bytes()
        assert data[:4] == b"glTF"
    rep = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert rep["objects"] == 2 and rep["ok"] == 2


def test_reference_yaml_keys_are_forwarded(tmp_path):
    stage = load_stage()
    cfg, _, _ = make_scene(tmp_path, ["a.png"])
    seen = {}

    def factory(config, device):
        s, t, c = fake_factory(config, device)
        seen["shape"] = s
        return s, t, c
    stage.main(["--config", cfg], factory=factory)
    assert type(seen["shape"]).calls[-1] == (50, 256, 16000, 1234567)


def test_no_images_is_an_error(tmp_path):
    stage = load_stage()
    cfg, _, _ = make_scene(tmp_path, ["wall.png"])
    with pytest.raises(FileNotFoundError):
        stage.main(["--config", cfg], factory=fake_factory)
    with pytest.raises(FileNotFoundError):
        stage.main(["--config", str(tmp_path / "missing.yaml")], factory=fake_factory)


def test_partition_is_a_disjoint_cover():
    stage = load_stage()
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            parts = [stage.partition(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_object_parallel_two_ranks_gloo(tmp_path):
    names = ["obj__(%d, %d).png" % (i, i) for i in range(5)]
    cfg, inp, out = make_scene(tmp_path, names)
    driver = tmp_path / "drv.py"
    driver.write_text(textwrap.dedent("""
        import importlib.util, os, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        spec = importlib.util.spec_from_file_location("stage_run", %r)
        stage = importlib.util.module_from_spec(spec); spec.loader.exec_module(stage)
        from test_stage_cpu import fake_factory
        def factory(config, device):
            s, t, c = fake_factory(config, device)
            open(os.path.join(%r, "loaded_rank%%s" %% os.environ["RANK"]), "a").write("x")
            return s, t, c
        sys.exit(stage.main(["--config", %r], factory=factory))
    """ % (os.path.join(ROOT, "3d-re-gen_amd"), os.path.join(ROOT, "tests"), STAGE, str(tmp_path), cfg)))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(driver)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(out)) == sorted(n[:-4] for n in names)       # every object exactly once, stale dir gone
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["objects"] == 5 and rep["ok"] == 5
    # the model is loaded once per RANK (the reference reloads it per image)
    assert open(tmp_path / "loaded_rank0").read() == "x" and open(tmp_path / "loaded_rank1").read() == "x"
