"""Worker of tests/test_nccl_one_rank_gpu.py: ONE rank, launched by torch.distributed.run on the leased MI355X, backend "nccl"
(= RCCL).  Drives every collective wrapper of r3g/dist.py with DEVICE tensors -- the branch the gloo tests cannot reach -- and
then the stage's run_distributed (the counterpart of the reference's pool, src/2d_to_3d_models/run.py:176-193) on a tiny model.
Prints one JSON line; any exception is a non-zero exit."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, in_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from r3g import dist as rdist
    rep = {"backend": dist.get_backend(), "world": world}
    assert rdist._comm_device().type == "cuda"
    rdist.barrier()
    # broadcast of the packed crops: lands in HBM
    rng = np.random.default_rng(0)
    crops = [rng.integers(0, 255, (h, w, 4), dtype=np.uint8) for (h, w) in ((64, 48), (512, 512), (7, 9))]
    got = rdist.broadcast_crops(crops if rank == 0 else None, src=0)
    assert all(g.is_cuda and g.dtype == torch.uint8 for g in got)
    assert all(np.array_equal(g.cpu().numpy(), c) for g, c in zip(got, crops))
    rep["broadcast_crops"] = [list(g.shape) for g in got]
    # the side store (port travels in a device-tensor broadcast), a queue on it, JSON exchange
    store = rdist.side_store()
    assert store is not None
    q = rdist.WorkQueue(10, name="one_rank", use_store=True)
    assert q.store is not None
    claims = [q.claim_guided(4, world=1), q.claim_guided(4, world=1), q.claim_guided(4, world=1), q.claim_guided(4, world=1)]
    assert claims == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9], []] and q.remaining() == 0
    rep["claims"] = claims
    assert rdist.all_ok(True) == [True] * world and rdist.all_ok(False) == [False] * world
    assert rdist.exchange_json({"rank": rank}, dst=0) == [{"rank": r} for r in range(world)]
    assert rdist.share_json(["a", 1]) == ["a", 1]
    # mesh return with device tensors: metadata all_gather on the device, the arrays stay in HBM
    v = torch.arange(12, dtype=torch.float32, device="cuda").view(4, 3)
    f = torch.tensor([[0, 1, 2], [1, 2, 3]], dtype=torch.int32, device="cuda")
    tex = torch.full((4, 4, 3), 7, dtype=torch.uint8, device="cuda")
    uv = torch.rand(4, 2, device="cuda")
    g = rdist.gather_meshes([(5, v, f), (2, v * 2, f, uv, tex)], dst=0, to_host=False)
    assert sorted(g) == [2, 5] and g[5][0].is_cuda and torch.equal(g[5][0], v) and torch.equal(g[5][1], f)
    assert len(g[2]) == 4 and torch.equal(g[2][3], tex) and torch.equal(g[2][2], uv)
    gh = rdist.gather_meshes([(5, v, f)], dst=0, to_host=True)
    assert isinstance(gh[5][0], np.ndarray) and np.array_equal(gh[5][1], f.cpu().numpy())
    rep["gather_meshes"] = "device tensors"
    rdist.barrier()
    # the stage's distributed path on this one rank: broadcast -> queue -> shape model + cleaners + texture -> gather -> GLBs
    import yaml
    from stage import run as stage_run
    with open(os.path.join(in_dir, "config.yaml")) as fh:
        config = yaml.safe_load(fh)
    from nccl_support import tiny_factory
    results, textured, failed = stage_run.run_distributed(config, config["prepped_for_hunyuan"], out_dir, rank, world, tiny_factory)
    assert failed == [] and results is not None
    rep["stage"] = [[os.path.basename(r[1]), r[2]] for r in results]
    rep["glbs"] = sorted(os.listdir(out_dir))
    rdist.barrier()
    rdist.reset()
    dist.destroy_process_group()
    print("NCCL_ONE_RANK " + json.dumps(rep))


if __name__ == "__main__":
    main()
