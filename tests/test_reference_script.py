"""The reference's OWN stage script, unmodified, against the hy3dgen mirror (SURVEY.md 8b level B3).

`/root/reference/src/2d_to_3d_models/run.py` is executed as the orchestrator executes it (reference run.py:61-122):
`python <script> --config <yaml>` with PYTHONPATH = src : src/utils : <where the orchestrator expects Hunyuan3D-2> -- the
last entry being this repo's package directory.  `utils.global_utils` is the reference's own module; `trimesh` is the
compat module shipped with the package (trimesh is not installed here); `huggingface_hub.snapshot_download` is patched to
return a local synthetic snapshot (no network).  The reference only exists in the build container and the container has no GPU, so
the three device touch points of the mirror are replaced by CPU stand-ins (tests/ref_shim.py); everything else that runs
is the product's code.  On a machine with a GPU the same script runs on the real path (tests/test_stage_gpu.py covers the
mirror's own stage script there)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRIPT = os.path.join(REF, "src", "2d_to_3d_models", "run.py")

pytestmark = pytest.mark.skipif(not os.path.exists(SCRIPT), reason="the reference checkout only exists in the build container")


def _scene(tmp_path, remesh):
    import torch
    from safetensors.torch import save_file
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import hy3d_torch as H
    from test_host_cpu import _snapshot_doc
    cfg = H.tiny_config()
    snap = tmp_path / "snapshot" / "hunyuan3d-dit-v2-0"
    snap.mkdir(parents=True)
    (snap / "config.yaml").write_text(yaml.safe_dump(_snapshot_doc(cfg)))
    sd = {k: v.contiguous() for k, v in H.synthetic_state_dict(cfg, seed=5).items()}
    save_file(sd, str(snap / "model.fp16.safetensors"))
    inp, out = tmp_path / "prepped", tmp_path / "out"
    inp.mkdir()
    out.mkdir()
    (out / "stale.glb").write_bytes(b"old")
    rng = np.random.default_rng(0)
    for name in ("chair__(10, 20).png", "wall__(1, 2).png"):
        img = np.zeros((96, 80, 4), np.uint8)
        img[20:70, 15:60, :3] = rng.integers(0, 255, (50, 45, 3))
        img[20:70, 15:60, 3] = 255
        Image.fromarray(img, "RGBA").save(inp / name)
    conf = {"mini": False, "num_inf_steps_hy": 3, "octree_resolution_hy": 20, "num_chunks_hy": 999, "seed": 1234567,
            "remesh": remesh, "remesh_target_num_faces": 300, "input_folder_hy": str(tmp_path / "unused"),
            "output_folder_hy": str(out), "use_banana": True, "prepped_for_hunyuan": str(inp), "jobs_per_gpu": 1,
            "use_all_available_cuda": False}
    cpath = tmp_path / "config.yaml"
    cpath.write_text(yaml.safe_dump(conf))
    return str(cpath), str(tmp_path / "snapshot"), out


@pytest.mark.parametrize("remesh", [False, True])
def test_reference_stage_script_runs_unmodified_against_the_mirror(tmp_path, remesh):
    cpath, snap, out = _scene(tmp_path, remesh)
    pkg = os.path.join(ROOT, "3d-re-gen_amd")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([
        os.path.join(ROOT, "tests", "stubs"),          # sitecustomize: snapshot_download stub + CPU stand-ins
        os.path.join(pkg, "compat"),                   # trimesh stand-in (trimesh is not installed here)
        os.path.join(REF, "src"), os.path.join(REF, "src", "utils"),    # as reference run.py:72-86 builds it
        pkg,                                           # where the orchestrator puts <root>/Hunyuan3D-2
        ROOT])                                         # the oracle package, for the stand-ins only
    env.update(R3G_TEST_CPU_SHIM="1", R3G_TEST_SNAPSHOT=snap, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="",
               R3G_TEX_SIZE="192", R3G_TEX_RENDER="96")      # small texture: the stand-in rasteriser is numpy
    r = subprocess.run([sys.executable, SCRIPT, "--config", cpath], cwd=os.path.join(REF, "src"), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Running sequentially" in r.stdout and "Using 'full' shape generator" in r.stdout
    assert sorted(os.listdir(out)) == ["chair__(10, 20)"]          # skip list honoured, stale content cleared
    glb = out / "chair__(10, 20)" / "chair__(10, 20).glb"
    data = glb.read_bytes()
    assert data[:4] == b"glTF"
    sys.path.insert(0, pkg)
    from r3g.mesh import load_glb
    m = load_glb(data)
    assert m.n_faces > 0 and m.faces.max() < m.n_vertices
    if remesh:
        assert "Remeshing enabled" in r.stdout and m.n_faces <= 300
    assert "Saved chair__(10, 20)" in r.stdout
    # the script's texgen call (reference run.py:97) produced a base-colour texture that an independent validator accepts
    from gltf_validate import validate_glb
    got = validate_glb(data)
    assert got["image"] is not None and got["image"].shape[:2] == (192, 192)
    assert "TEXCOORD_0" in got["attributes"] and len(got["attributes"]["TEXCOORD_0"]) == len(got["positions"])
