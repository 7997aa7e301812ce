"""The N = 8 path of bench.py -- the one the driver launches on an 8-GPU node -- executed end to end on ONE GPU: eight ranks
share the device (R3G_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0 and the process group on gloo; RCCL refuses two ranks
on one device), so that the broadcast of the crops, the per-rank objects, the warm-up gather, the batched point-to-point
return of the meshes (r3g/dist.py: gather_meshes), the strong block's dynamic queue and the teardown have all run somewhere
before a real node sees them.  The counterpart of the reference's pool, src/2d_to_3d_models/run.py:176-193.  Small model
settings (mini dims, 2 steps, 65^3 grid): this is a functional test, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [8])
def test_bench_with_eight_ranks_on_one_device(world):
    env = dict(os.environ)
    env.update(R3G_BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--model", "mini", "--inference-steps", "2", "--octree-resolution", "64",
           "--steps", "4", "--warmup", "1", "--objects-per-launch", "2", "--no-roofline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 alone prints the line
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["objects_total"] == world * 4 and out["value"] > 0
    # the strong block: 8 crops per rank through the shared queue, every mesh came back to rank 0
    assert out["strong"]["objects_total"] == 8 * world and out["strong"]["value"] > 0
    # round 6: one record per rank in both blocks -- which of load balance, mesh return and host contention a scaling number below
    # N x is made of -- and the fp16-guard counter
    for block, objects in ((out["per_rank"], world * 4), (out["strong"]["per_rank"], 8 * world)):
        assert [r["rank"] for r in block] == list(range(world))
        assert sum(r["objects"] for r in block) == objects
        for r in block:
            assert set(r) >= {"objects", "compute_s", "queue_wait_s", "gather_s", "end_wait_s"}
            assert r["compute_s"] >= 0 and r["gather_s"] >= 0 and r["queue_wait_s"] >= 0
    assert all(r["objects"] == 4 for r in out["per_rank"])
    assert out["dit_f16_fallbacks"] == 0 and out["dit_groups"] > 0
