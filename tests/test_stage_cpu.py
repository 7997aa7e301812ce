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
        from gltf_validate import validate_glb                     # independent glTF 2.0 checks (tests/gltf_validate.py)
        got = validate_glb(data)
        assert len(got["positions"]) > 0 and len(got["indices"]) > 0
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


def content_factory(config, device):
    """deterministic CPU stand-in whose mesh is a function of the crop's pixels (a different mesh for every object):
    a fan of triangles whose vertex coordinates are the crop's first bytes"""
    import numpy as np
    from r3g.mesh import Mesh

    class Shape:
        def __call__(self, image=None, num_inference_steps=None, octree_resolution=None, num_chunks=None,
                     generator=None, output_type=None):
            px = np.asarray(image, np.uint8).reshape(-1).astype(np.float32)
            n = 6 + int(px[0]) % 7
            v = (px[: 3 * n].reshape(n, 3) / 255.0 + np.arange(n, dtype=np.float32)[:, None] * 0.125).astype(np.float32)
            f = np.stack([np.zeros(n - 2, np.int64), np.arange(1, n - 1), np.arange(2, n)], 1)
            return [Mesh(v, f)]
    return Shape(), (lambda mesh, image=None: mesh), [lambda m: m]


def make_distinct_scene(tmp_path, n):
    cfg, inp, out = make_scene(tmp_path, [])
    rng = np.random.default_rng(7)
    names = []
    for i in range(n):
        name = "obj__(%d, %d).png" % (i, 3 * i)
        Image.fromarray(rng.integers(0, 256, (16 + i, 12 + 2 * i, 4), dtype=np.uint8), "RGBA").save(inp / name)
        names.append(name)
    return cfg, inp, out, names


def _run_ranks(tmp_path, cfg, world, port, factory_name="fake_factory", expect_rc=0):
    driver = tmp_path / ("drv%d.py" % world)
    driver.write_text(textwrap.dedent("""
        import importlib.util, os, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        spec = importlib.util.spec_from_file_location("stage_run", %r)
        stage = importlib.util.module_from_spec(spec); spec.loader.exec_module(stage)
        import test_stage_cpu as T
        def factory(config, device):
            s, t, c = getattr(T, %r)(config, device)
            open(os.path.join(%r, "loaded_rank%%s" %% os.environ.get("RANK", "0")), "a").write("x")
            return s, t, c
        sys.exit(stage.main(["--config", %r], factory=factory))
    """ % (os.path.join(ROOT, "3d-re-gen_amd"), os.path.join(ROOT, "tests"), STAGE, factory_name, str(tmp_path), cfg)))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    if world == 1:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, str(driver)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(driver)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if expect_rc is not None:
        assert (r.returncode == 0) == (expect_rc == 0), r.stdout + r.stderr
    return r


def test_object_parallel_two_ranks_gloo(tmp_path):
    names = ["obj__(%d, %d).png" % (i, i) for i in range(5)]
    cfg, inp, out = make_scene(tmp_path, names)
    r = _run_ranks(tmp_path, cfg, 2, 29611)
    assert sorted(os.listdir(out)) == sorted(n[:-4] for n in names)       # every object exactly once, stale dir gone
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["objects"] == 5 and rep["ok"] == 5 and rep["textured"] is True
    assert sorted(set(rep["rank_of_object"])) <= [0, 1] and len(rep["rank_of_object"]) == 5
    # the model is loaded once per RANK (the reference reloads it per image)
    assert open(tmp_path / "loaded_rank0").read() == "x" and open(tmp_path / "loaded_rank1").read() == "x"


def _glbs(out):
    return {d: (out / d / (d + ".glb")).read_bytes() for d in sorted(os.listdir(out))}


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_meshes_are_byte_identical_to_the_one_rank_run(tmp_path, world):
    """the same inputs at world size 1 and N must give byte-identical GLBs (object-parallel =>
    exact).  Crops of different sizes and content, scattered by broadcast, claimed dynamically, gathered to rank 0."""
    cfg, inp, out, names = make_distinct_scene(tmp_path, 7)
    _run_ranks(tmp_path, cfg, 1, 0, "content_factory")
    one = _glbs(out)
    assert sorted(one) == sorted(n[:-4] for n in names)
    assert len(set(one.values())) == len(names)                  # the stand-in really produces a different mesh per crop
    r = _run_ranks(tmp_path, cfg, world, 29620 + world, "content_factory")
    many = _glbs(out)
    assert many == one
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["objects"] == 7 and rep["ok"] == 7


def test_streamed_outputs_equal_the_gathered_ones(tmp_path):
    """r3g_stream_outputs: every rank writes its GLB as soon as the object is done; same bytes as the gather to rank 0"""
    import yaml
    cfg, inp, out, names = make_distinct_scene(tmp_path, 5)
    _run_ranks(tmp_path, cfg, 2, 29641, "content_factory")
    gathered = _glbs(out)
    conf = yaml.safe_load(open(cfg))
    conf["r3g_stream_outputs"] = True
    open(cfg, "w").write(yaml.safe_dump(conf))
    r = _run_ranks(tmp_path, cfg, 2, 29642, "content_factory")
    assert _glbs(out) == gathered and len(gathered) == 5
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["ok"] == 5 and sorted(set(rep["rank_of_object"])) == [0, 1]


def test_failed_object_does_not_fail_the_distributed_stage(tmp_path):
    cfg, inp, out, names = make_distinct_scene(tmp_path, 4)
    r = _run_ranks(tmp_path, cfg, 2, 29631, "flaky_factory")
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["objects"] == 4 and rep["ok"] == 3 and rep["failed"] == [names[2]]
    assert sorted(os.listdir(out)) == sorted(n[:-4] for i, n in enumerate(names) if i != 2)


def flaky_factory(config, device):
    s, t, c = content_factory(config, device)

    class Flaky:
        def __call__(self, image=None, **kw):
            if image.size == (12 + 2 * 2, 16 + 2):        # the third crop of make_distinct_scene
                raise RuntimeError("synthetic failure")
            return s(image=image, **kw)
    return Flaky(), t, c


def test_broadcast_crops_and_gather_roundtrip_single_process(tmp_path):
    """r3g.dist on a world of one (gloo): shapes, dtypes and bytes survive broadcast and gather"""
    import torch
    import torch.distributed as dist
    from r3g import dist as rdist
    dist.init_process_group("gloo", init_method="file://%s" % (tmp_path / "rdzv"), rank=0, world_size=1)
    try:
        rng = np.random.default_rng(0)
        crops = [rng.integers(0, 256, (5 + i, 9, 4), dtype=np.uint8) for i in range(3)]
        got = rdist.broadcast_crops(crops)
        assert [tuple(g.shape) for g in got] == [c.shape for c in crops]
        assert all(np.array_equal(g.numpy(), c) for g, c in zip(got, crops))
        q = rdist.WorkQueue(3)
        assert [q.claim() for _ in range(5)] == [0, 1, 2, None, None]
        v = np.arange(12, dtype=np.float32).reshape(4, 3)
        f = np.array([[0, 1, 2], [0, 2, 3]])
        out = rdist.gather_meshes([(4, torch.from_numpy(v), torch.from_numpy(f)), (1, v[:3], f[:1])])
        assert sorted(out) == [1, 4] and np.array_equal(out[4][0], v) and out[4][1].dtype == np.int32
        dev = rdist.gather_meshes([(4, torch.from_numpy(v), torch.from_numpy(f))], to_host=False)     # stays a tensor
        assert isinstance(dev[4][0], torch.Tensor) and torch.equal(dev[4][0], torch.from_numpy(v)) and dev[4][1].dtype == torch.int32
        assert rdist.broadcast_crops([]) == []
    finally:
        dist.destroy_process_group()


def test_world_of_eight_ranks_gloo(tmp_path):
    """the driver's 8-GPU run, on the CPU: 8 ranks, 19 objects claimed two at a time from the shared counter, crops by
    one broadcast, meshes back point to point -- byte-identical to the one-rank run.  This is the code the RCCL run
    executes; only the backend string and the device of the tensors differ."""
    cfg, inp, out, names = make_distinct_scene(tmp_path, 19)
    _run_ranks(tmp_path, cfg, 1, 0, "content_factory")
    one = _glbs(out)
    r = _run_ranks(tmp_path, cfg, 8, 29660, "content_factory")
    assert _glbs(out) == one and len(one) == 19
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["objects"] == 19 and rep["ok"] == 19
    assert len(set(rep["rank_of_object"])) >= 4          # the queue really spread the work


def batch_factory(config, device):
    """stand-in that takes a LIST of images like the real pipeline (accepts_image_list) and records the group sizes"""
    s, t, c = content_factory(config, device)

    class Batched:
        accepts_image_list = True
        groups = []

        def __call__(self, image=None, generator=None, **kw):
            if isinstance(image, (list, tuple)):
                assert isinstance(generator, list) and len(generator) == len(image)
                assert len({g.initial_seed() for g in generator}) == 1 and len({id(g) for g in generator}) == len(image)
                Batched.groups.append(len(image))
                return [s(image=im, generator=g, **kw)[0] for im, g in zip(image, generator)]
            Batched.groups.append(1)
            return s(image=image, generator=generator, **kw)
    return Batched(), t, c


def test_objects_per_launch_groups_give_the_same_files(tmp_path):
    """r3g_objects_per_launch: crops go through the shape pipeline in groups (one generator per object, each seeded with
    cfg.seed as the reference seeds every call); the files do not depend on the group size"""
    stage = load_stage()
    cfg, inp, out, names = make_distinct_scene(tmp_path, 5)
    assert stage.main(["--config", cfg], factory=content_factory) == 0
    ref = _glbs(out)
    for per, want in ((1, [1, 1, 1, 1, 1]), (2, [2, 2, 1]), (4, [4, 1]), (None, [4, 1])):
        conf = yaml.safe_load(open(cfg))
        conf.pop("r3g_objects_per_launch", None)
        if per is not None:
            conf["r3g_objects_per_launch"] = per
        open(cfg, "w").write(yaml.safe_dump(conf))
        seen = {}

        def factory(config, device):
            sgen, t, c = batch_factory(config, device)
            type(sgen).groups.clear()
            seen["s"] = sgen
            return sgen, t, c
        assert stage.main(["--config", cfg], factory=factory) == 0
        assert type(seen["s"]).groups == want
        assert _glbs(out) == ref


def dying_factory(config, device):
    """rank 1 cannot load its model (the ADVICE scenario: an out-of-memory on one GPU)"""
    if os.environ.get("RANK") == "1":
        raise MemoryError("synthetic: model load failed on this rank")
    return content_factory(config, device)


def test_a_rank_that_dies_at_model_load_does_not_hang_the_others(tmp_path):
    cfg, inp, out, names = make_distinct_scene(tmp_path, 6)
    r = _run_ranks(tmp_path, cfg, 3, 29670, "dying_factory", expect_rc=4)
    assert r.returncode != 0                                   # a rank-level failure is visible in the exit code
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["objects"] == 6 and rep["ok"] == 6            # the surviving ranks took every object
    assert sorted(os.listdir(out)) == sorted(n[:-4] for n in names)
    assert "rank(s) [1] failed" in r.stderr


def test_work_queues_of_the_same_name_start_fresh(tmp_path):
    """ADVICE: a second WorkQueue with the same name must not continue the first one's count (two ranks, gloo)"""
    drv = tmp_path / "q.py"
    drv.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from r3g import dist as rdist
        dist.init_process_group("gloo")
        got = []
        for n in (5, 4):
            q = rdist.WorkQueue(n, name="same")
            mine = []
            while True:
                c = q.claim_many(2)
                if not c:
                    break
                mine += c
            got.append(mine)
        allgot = rdist.exchange_json(got)
        if dist.get_rank() == 0:
            first = sorted(i for g in allgot for i in g[0]); second = sorted(i for g in allgot for i in g[1])
            assert first == [0, 1, 2, 3, 4] and second == [0, 1, 2, 3], (first, second)
            print("QUEUES_OK")
        assert rdist.all_ok(True) == [True, True]
        assert rdist.share_json({"a": [1, 2]} if dist.get_rank() == 0 else None) == {"a": [1, 2]}
        rdist.barrier()
        dist.destroy_process_group()
    """ % os.path.join(ROOT, "3d-re-gen_amd")))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29680", str(drv)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "QUEUES_OK" in r.stdout, r.stdout + r.stderr


def prefetching_factory(config, device):
    """content_factory with the real pipeline's list call, prefetch / close_prefetch (recorded per rank in a log file)"""
    s, t, c = batch_factory(config, device)
    log = os.path.join(os.path.dirname(config["prepped_for_hunyuan"]), "prefetch_rank%s.log" % os.environ.get("RANK", "0"))

    class Ahead(type(s)):
        def prefetch(self, images):
            self._ahead = [id(im) for im in images]
            open(log, "a").write("prefetch %d\n" % len(images))

        def close_prefetch(self):
            open(log, "a").write("close\n")

        def __call__(self, image=None, **kw):
            ims = image if isinstance(image, (list, tuple)) else [image]
            hit = getattr(self, "_ahead", None) == [id(im) for im in ims]
            if hit:                                   # (as the pipeline: a call on other objects leaves the prefetch alone)
                self._ahead = None
            open(log, "a").write("call %d %s\n" % (len(ims), "prepared-ahead" if hit else "cold"))
            return super().__call__(image=image, **kw)
    return Ahead(), t, c


def test_ranks_claim_ahead_and_prefetch_while_work_is_plentiful(tmp_path):
    """two ranks, 21 objects, groups of 4: while at least 4 x world objects are unclaimed a rank claims one group ahead and
    hands exactly those image objects to the pipeline's prefetch; near the end it claims only when free; same files as the
    one-rank run; the prefetch pool is closed on every rank"""
    cfg, inp, out, names = make_distinct_scene(tmp_path, 21)
    _run_ranks(tmp_path, cfg, 1, 0, "content_factory")
    one = _glbs(out)
    r = _run_ranks(tmp_path, cfg, 2, 29681, "prefetching_factory")
    assert _glbs(out) == one and len(one) == 21
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["ok"] == 21
    calls = picked_up = 0
    for rank in (0, 1):
        lines = open(tmp_path / ("prefetch_rank%d.log" % rank)).read().split("\n")
        lines = [l for l in lines if l]
        assert lines[-1] == "close" and lines.count("close") == 1
        assert lines[0].startswith("prefetch") or lines[0].startswith("call")
        ahead = [l for l in lines if l.endswith("prepared-ahead")]
        # every prefetched group is a full one (claims ahead only happen while >= 4 x world objects are left) and was picked up
        assert len(ahead) == lines.count("prefetch 4") == sum(l.startswith("prefetch") for l in lines)
        picked_up += len(ahead)
        calls += sum(int(l.split()[1]) for l in lines if l.startswith("call"))
    assert calls == 21 and picked_up >= 1       # (which rank claims ahead how often depends on who gets to the counter first)


def slow_batch_factory(config, device):
    """batch_factory whose calls take half a second (a rank that is on an object cannot come back for the next one at once)"""
    import time
    s, t, c = batch_factory(config, device)

    class Slow(type(s)):
        def __call__(self, image=None, **kw):
            time.sleep(0.5)
            return super().__call__(image=image, **kw)
    return Slow(), t, c


def test_eight_crops_on_eight_ranks_are_one_object_each(tmp_path):
    """BASELINE.json configs[1] (1 scene / 8 crops) on an 8-GPU node: guided claims hand out single objects, so the work
    spreads over the ranks instead of two ranks taking four objects each"""
    cfg, inp, out, names = make_distinct_scene(tmp_path, 8)
    r = _run_ranks(tmp_path, cfg, 8, 29690, "slow_batch_factory")
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"stage"')][-1])
    assert rep["objects"] == 8 and rep["ok"] == 8
    per_rank = {k: rep["rank_of_object"].count(k) for k in set(rep["rank_of_object"])}
    # the ranks leave the queue's constructor together (a barrier) and an object takes 0.5 s: nobody gets a second one while
    # others have none -- unless a rank is more than half a second late on a loaded machine, hence the slack
    assert len(per_rank) >= 6 and max(per_rank.values()) <= 2, per_rank
