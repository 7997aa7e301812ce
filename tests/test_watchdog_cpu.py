"""r3g.dist.Watchdog + init_process_group(timeout) (round 6): a rank that stops making progress -- hung, not dead; a dead one the
launcher notices itself -- ends the JOB with a non-zero exit code and a message that names it, well inside the launcher's limit.
What it replaces: the reference's pool loses the task silently (src/2d_to_3d_models/run.py:176-193).  gloo, world 2 and 3."""
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


DRIVER = """
import os, sys, time
sys.path.insert(0, %r)
import torch.distributed as dist
from r3g import dist as rdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rdist.init_process_group("gloo", rank=rank, world_size=world, timeout_s=60)
rdist.side_store()
wd = rdist.Watchdog(limit_s=3.0, interval_s=0.5)
mode = %r
t0 = time.time()
while time.time() - t0 < %f:
    if not (mode == "hang" and rank == world - 1 and time.time() - t0 > 1.0):
        wd.beat("step at %%.1f s" %% (time.time() - t0))
    time.sleep(0.2)
wd.done()
if mode == "early" and rank == 0:
    time.sleep(5.0)        # the others are done and wait: a finished rank is not a hung rank
rdist.barrier()
dist.destroy_process_group()
print("rank %%d finished" %% rank)
"""


def _run(tmp_path, world, mode, seconds):
    drv = tmp_path / "wd_driver.py"
    drv.write_text(textwrap.dedent(DRIVER % (os.path.join(ROOT, "3d-re-gen_amd"), mode, seconds)))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(drv)]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
    return r, time.time() - t0


def test_a_hung_rank_ends_the_job_with_its_name(tmp_path):
    r, secs = _run(tmp_path, 3, "hang", 40.0)
    assert r.returncode != 0
    assert "no progress from rank 2" in r.stderr, r.stderr[-2000:]
    assert "[r3g watchdog]" in r.stderr and "last phase 'step at" in r.stderr
    assert secs < 35.0, "the watchdog (limit 3 s) should have ended the 40-second job early: %.1f s" % secs


@pytest.mark.parametrize("mode", ["ok", "early"])
def test_healthy_and_early_finishing_ranks_are_left_alone(tmp_path, mode):
    r, _ = _run(tmp_path, 2, mode, 4.0)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[r3g watchdog]" not in r.stderr
    assert "rank 0 finished" in r.stdout and "rank 1 finished" in r.stdout
