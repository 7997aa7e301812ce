"""INTEGRATION.md route 1 on the hardware: the reference's OWN stage script (`src/2d_to_3d_models/run.py`, unmodified) importing the
hy3dgen mirror and running on the MI355X through libr3g.so -- the GPU twin of tests/test_reference_script.py, WITHOUT its CPU
stand-ins.  The reference checkout does not exist on the GPU box: the test looks for the two reference files
it needs (the stage script and `src/utils/global_utils.py`, which the script imports) under $R3G_REFERENCE_ROOT, /root/reference or
oracle/_ref/reference_src (git-ignored scratch that `tools/r05_reference_on_gpu.sh` fills for one gpurun call and empties again) and
skips when they are nowhere.  Both of the script's routes: the sequential one (reference :194-213) and its multiprocessing pool
(:176-193, `jobs_per_gpu: 2`: two spawned workers, each loading the model on the GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml
from PIL import Image

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_root():
    for cand in (os.environ.get("R3G_REFERENCE_ROOT"), "/root/reference", os.path.join(ROOT, "oracle", "_ref", "reference_src")):
        if cand and os.path.exists(os.path.join(cand, "src", "2d_to_3d_models", "run.py")) and \
                os.path.exists(os.path.join(cand, "src", "utils", "global_utils.py")):
            return cand
    return None


REF = _reference_root()


@pytest.mark.skipif(REF is None, reason="the reference's stage script is not on this machine")
@pytest.mark.parametrize("route", ["sequential", "pool"])
def test_reference_stage_script_unmodified_on_the_gpu(tmp_path, route):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_reference_script import _scene
    cpath, snap, out = _scene(tmp_path, remesh=False)
    conf = yaml.safe_load(open(cpath))
    if route == "pool":
        rng = np.random.default_rng(9)
        img = np.zeros((96, 80, 4), np.uint8)
        img[25:75, 10:55, :3] = rng.integers(0, 255, (50, 45, 3))
        img[25:75, 10:55, 3] = 255
        Image.fromarray(img, "RGBA").save(os.path.join(conf["prepped_for_hunyuan"], "table__(3, 4).png"))
        conf["jobs_per_gpu"] = 2
        open(cpath, "w").write(yaml.safe_dump(conf))
    pkg = os.path.join(ROOT, "3d-re-gen_amd")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([
        os.path.join(ROOT, "tests", "stubs"),          # sitecustomize: the snapshot_download stub only (no CPU stand-ins)
        os.path.join(pkg, "compat"),                   # trimesh stand-in (trimesh is not installed here)
        os.path.join(REF, "src"), os.path.join(REF, "src", "utils"),    # as reference run.py:72-86 builds it
        pkg])                                          # where the orchestrator puts <root>/Hunyuan3D-2
    env.pop("R3G_TEST_CPU_SHIM", None)
    env.update(R3G_TEST_SNAPSHOT=snap, R3G_TEST_REPORT_MAPS="1", R3G_TEX_SIZE="256", R3G_TEX_RENDER="128")
    script = os.path.join(REF, "src", "2d_to_3d_models", "run.py")
    r = subprocess.run([sys.executable, script, "--config", cpath], cwd=os.path.join(REF, "src"), env=env,
                       capture_output=True, text=True, timeout=900)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "Using 'full' shape generator" in r.stdout
    stems = ["chair__(10, 20)"] + (["table__(3, 4)"] if route == "pool" else [])
    assert sorted(os.listdir(out)) == sorted(stems), tail      # skip list honoured, stale content cleared
    if route == "sequential":
        assert "Running sequentially" in r.stdout
        assert "native libraries mapped at exit: libr3g.so" in r.stdout, tail     # the HIP library, no stand-in
    else:
        assert "2 parallel worker(s)" in r.stdout and "All parallel tasks completed." in r.stdout
        assert r.stdout.count("finished '") == 2 and "ERROR in worker" not in r.stdout, tail
    sys.path.insert(0, pkg)
    from r3g.mesh import load_glb
    from gltf_validate import validate_glb
    for stem in stems:
        data = (out / stem / (stem + ".glb")).read_bytes()
        m = load_glb(data)
        assert m.n_faces > 0 and m.faces.max() < m.n_vertices and np.isfinite(m.vertices).all()
        got = validate_glb(data)
        assert got["image"] is not None and "TEXCOORD_0" in got["attributes"]
        assert "Saved %s" % stem in r.stdout
    print("\n---- the reference script's own output (route %s), last lines ----\n%s" % (route, "\n".join(r.stdout.strip().splitlines()[-22:])))   # shown under -s: the log in profiles/
