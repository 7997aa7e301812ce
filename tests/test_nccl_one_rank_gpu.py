"""The RCCL branch of r3g/dist.py and of the stage's run_distributed, executed: ONE rank under torch.distributed.run with backend
"nccl" on the leased MI355X (every distributed test so far was gloo).  broadcast_crops, the side store's port
broadcast, WorkQueue on the store, all_ok, exchange_json / share_json, gather_meshes with device tensors, barrier(device_ids)
-- then run_distributed on a tiny model writes its GLBs.  And bench.py under the launch line the driver uses for N = 8, at
N = 1: the process group is RCCL, the crops are broadcast, the meshes gathered.  Counterpart of the reference's pool,
/root/reference/src/2d_to_3d_models/run.py:176-193."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(args, timeout, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_one_rank_rccl_group_drives_every_collective_and_the_stage(tmp_path):
    from PIL import Image
    inp, out = tmp_path / "prepped", tmp_path / "out"
    inp.mkdir()
    out.mkdir()
    rng = np.random.default_rng(3)
    for k in range(3):
        a = np.zeros((96, 80, 4), np.uint8)
        a[20:70 - k, 15:60, :3] = rng.integers(0, 255, (50 - k, 45, 3))
        a[20:70 - k, 15:60, 3] = 255
        Image.fromarray(a, "RGBA").save(inp / ("thing__(%d, %d).png" % (k, k)))
    Image.fromarray(np.full((8, 8, 4), 255, np.uint8), "RGBA").save(inp / "floor__(0, 0).png")      # skip list
    cfg = {"mini": False, "num_inf_steps_hy": 3, "octree_resolution_hy": 24, "num_chunks_hy": 999, "seed": 1234567,
           "remesh": False, "input_folder_hy": str(tmp_path / "unused"), "output_folder_hy": str(out), "use_banana": True,
           "prepped_for_hunyuan": str(inp), "jobs_per_gpu": 1, "use_all_available_cuda": False, "r3g_objects_per_launch": 2}
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(cfg))
    p = _torchrun([os.path.join(ROOT, "tests", "workers", "nccl_one_rank.py"), str(out), str(tmp_path)], 900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("NCCL_ONE_RANK ")]
    assert len(line) == 1, p.stdout[-2000:]
    rep = json.loads(line[0][len("NCCL_ONE_RANK "):])
    assert rep["backend"] == "nccl" and rep["gather_meshes"] == "device tensors"
    assert [s[1] for s in rep["stage"]] == ["ok"] * 3
    assert rep["glbs"] == ["thing__(0, 0)", "thing__(1, 1)", "thing__(2, 2)"]
    sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
    from r3g.mesh import load_glb
    for stem in rep["glbs"]:
        m = load_glb(str(out / stem / (stem + ".glb")))
        assert m.n_faces > 0 and m.faces.max() < m.n_vertices


def test_bench_under_the_drivers_launch_line_at_one_rank_uses_rccl():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` at N = 1: bench.py sees the
    launcher's environment and goes through its multi-rank path on an RCCL group of one (broadcast of the crops into HBM, meshes
    gathered as device tensors, the strong block's queue) -- small model settings, a functional test"""
    p = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "mini", "--inference-steps", "2",
                   "--octree-resolution", "64", "--steps", "2", "--warmup", "1", "--objects-per-launch", "2", "--no-roofline",
                   "--no-cpu-baseline"], 1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["process_group"] == "nccl"
    assert out["strong"]["objects_total"] == 8 and out["strong"]["value"] > 0
