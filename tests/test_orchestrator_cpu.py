"""SURVEY.md 8(b) level B1 through the reference's ORCHESTRATOR: `/root/reference/run.py -p 3 --config <yaml>` executed
UNMODIFIED (reference run.py:61-122 run_script, :167-173 GPU allow-list, :204-207, registry :258-266).

The orchestrator derives every path from the directory its own file is reached through (`script_dir`, run.py:225), so the
test lays a scratch checkout out of symlinks -- nothing is copied from, or written to, /root/reference:

    <root>/run.py                      -> /root/reference/run.py                       (the orchestrator, unmodified)
    <root>/venv_py310/bin/python       -> this interpreter                             (registry "venv", run.py:227)
    <root>/src/2d_to_3d_models/run.py  -> the stage script under test                  (registry "script", run.py:260)
    <root>/src/config.yaml                                                             (registry "args", run.py:262)
    <root>/Hunyuan3D-2                 -> this repo's 3d-re-gen_amd (the hy3dgen mirror; run.py:82 puts it on PYTHONPATH)

Two drop-in routes of INTEGRATION.md go through it: (2) THIS repo's stage script sitting where the registry expects the
stage, and (1) the reference's own stage script with the mirror standing in for the Hunyuan3D-2 submodule.  Stage 3 is
"Hunyuan_2d_to_3d" in the default (Use_VGGT) order, run.py:430-442.  The build container has no GPU: the device touch
points are the CPU stand-ins of tests/ref_shim.py (API / process contract only), exactly as in test_reference_script.py."""
import os
import subprocess
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "run.py")),
                                reason="the reference checkout only exists in the build container")


def _checkout(tmp_path, stage_script, conf_path):
    root = tmp_path / "checkout"
    (root / "venv_py310" / "bin").mkdir(parents=True)
    os.symlink(sys.executable, root / "venv_py310" / "bin" / "python")
    os.symlink(os.path.join(REF, "run.py"), root / "run.py")
    (root / "src" / "2d_to_3d_models").mkdir(parents=True)
    os.symlink(stage_script, root / "src" / "2d_to_3d_models" / "run.py")
    os.symlink(os.path.join(REF, "src", "utils"), root / "src" / "utils")          # utils.global_utils of the reference
    os.symlink(os.path.join(ROOT, "3d-re-gen_amd"), root / "Hunyuan3D-2")
    # test scaffolding on the stage's PYTHONPATH (its cwd): the snapshot_download stub and the CPU stand-ins
    os.symlink(os.path.join(ROOT, "tests", "stubs", "sitecustomize.py"), root / "src" / "sitecustomize.py")
    conf = yaml.safe_load(open(conf_path))
    conf.update(device_global="cuda:0", conda_env=None, Use_VGGT=True, use_hunyuan21=False, Use_MIDI=False, Use_DPA=False)
    (root / "src" / "config.yaml").write_text(yaml.safe_dump(conf))
    return root


@pytest.mark.parametrize("route", ["this repo's stage script", "the reference's stage script + the mirror"])
def test_orchestrator_runs_stage_three(tmp_path, route):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_reference_script import _scene
    cpath, snap, out = _scene(tmp_path, remesh=False)
    ours = route.startswith("this")
    if ours:
        conf = yaml.safe_load(open(cpath))
        conf["r3g_weights"] = snap
        open(cpath, "w").write(yaml.safe_dump(conf))
    script = (os.path.join(ROOT, "3d-re-gen_amd", "stage", "run.py") if ours
              else os.path.join(REF, "src", "2d_to_3d_models", "run.py"))
    root = _checkout(tmp_path, script, cpath)
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    env.update(R3G_TEST_CPU_SHIM="1", R3G_TEST_SNAPSHOT=snap, R3G_TEX_SIZE="192", R3G_TEX_RENDER="96",
               R3G_TEST_EXTRA_PATH=os.pathsep.join([os.path.join(ROOT, "3d-re-gen_amd", "compat"), ROOT]),
               HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, str(root / "run.py"), "-p", "3", "--config", str(root / "src" / "config.yaml")],
                       cwd=str(root), env=env, capture_output=True, text=True, timeout=900)
    tail = r.stdout[-4000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "Running 2D cropped images to 3D models" in r.stdout, tail            # registry entry 3 (run.py:258-266)
    assert "Finished 2D cropped images to 3D models" in r.stdout and "Failed to run" not in r.stdout, tail
    assert "Allowing subprocess to access all available GPUs" not in r.stdout    # use_all_available_cuda: false (:167-173)
    assert sorted(os.listdir(out)) == ["chair__(10, 20)"]                          # B2: skip list, stale content cleared
    data = (out / "chair__(10, 20)" / "chair__(10, 20).glb").read_bytes()
    from gltf_validate import validate_glb
    got = validate_glb(data)
    assert len(got["indices"]) > 0 and got["image"] is not None
    if ours:
        assert '{"stage": "Hunyuan_2d_to_3d"' in r.stdout


def test_orchestrator_stops_when_the_stage_fails(tmp_path):
    """B1: a non-zero exit of the stage aborts the remaining stages (run.py:197-200).  No images -> FileNotFoundError in
    the stage -> the orchestrator reports the failure."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_reference_script import _scene
    cpath, snap, out = _scene(tmp_path, remesh=False)
    conf = yaml.safe_load(open(cpath))
    conf["r3g_weights"] = snap
    conf["prepped_for_hunyuan"] = str(tmp_path / "empty")
    (tmp_path / "empty").mkdir()
    open(cpath, "w").write(yaml.safe_dump(conf))
    root = _checkout(tmp_path, os.path.join(ROOT, "3d-re-gen_amd", "stage", "run.py"), cpath)
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    env.update(R3G_TEST_CPU_SHIM="1", R3G_TEST_SNAPSHOT=snap, HIP_VISIBLE_DEVICES="",
               R3G_TEST_EXTRA_PATH=os.pathsep.join([os.path.join(ROOT, "3d-re-gen_amd", "compat"), ROOT]))
    r = subprocess.run([sys.executable, str(root / "run.py"), "-p", "3", "--config", str(root / "src" / "config.yaml")],
                       cwd=str(root), env=env, capture_output=True, text=True, timeout=600)
    assert "Failed to run 2D cropped images to 3D models" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "No images found" in r.stderr
