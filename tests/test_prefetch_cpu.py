"""Host preparation of the NEXT launch group on a thread (hy3dgen/shapegen/pipelines.py: prefetch / _prepared /
close_prefetch; used by stage/run.py and bench.py so that a crop's ~40 ms of host work runs under the previous group's GPU
loop).  The host side needs no GPU: the pipeline object is built around a stub model."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))


class _StubModel:
    num_latents, in_channels = 8, 4


def _pipeline():
    import hy3dgen.shapegen.pipelines as pl

    class P(pl.Hunyuan3DDiTFlowMatchingPipeline):
        def _make_model(self, cfg, state_dict, grid_chunk):
            return _StubModel()
    cfg = pl.builtin_config("full")
    cfg["cond"]["image_size"] = 70
    return P(cfg, {}, "cuda:0")


def _crop(seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    a = np.zeros((96, 80, 4), np.uint8)
    a[20:70, 10:60, :3] = rng.integers(0, 255, (50, 50, 3))
    a[20:70, 10:60, 3] = 255
    return Image.fromarray(a, "RGBA")


def test_prefetched_inputs_equal_the_direct_ones_and_run_on_the_worker_thread():
    p = _pipeline()
    imgs = [_crop(0), _crop(1), _crop(2)]
    direct = [p._host_prepare(im) for im in imgs]
    assert all(d.shape == (3, 70, 70) and d.dtype == torch.float32 for d in direct)
    seen = []
    orig = p._host_prepare
    p._host_prepare = lambda im: (seen.append(threading.current_thread().name), orig(im))[1]
    before = torch.get_num_threads()
    try:
        p.prefetch(imgs)
        # round 5: the pool leaves torch's process-wide intra-op thread count alone (the host path runs no torch operator)
        assert torch.get_num_threads() == before
        got = p._prepared(imgs)                              # the same image OBJECTS: picked up from the worker
        assert all(torch.equal(a, b) for a, b in zip(got, direct))
        assert len(seen) == 3 and all(n.startswith("r3g-host-prep") for n in seen)
        assert p._prefetched == {}                           # consumed
        # a call on other objects (here: equal pixels, different objects) prepares them itself, on this thread
        seen.clear()
        p.prefetch(imgs)
        other = [im.copy() for im in imgs[:2]]
        got2 = p._prepared(other)
        assert all(torch.equal(a, b) for a, b in zip(got2, direct[:2]))
        assert seen.count(threading.current_thread().name) == 2
        # (round 6: kept per image) a second prefetch of an image that is already pending changes nothing; a single image is accepted
        p.prefetch(imgs[2])
        assert len(p._prefetched) == 3
        assert torch.equal(p._prepared([imgs[2]])[0], direct[2])
        assert all(torch.equal(a, b) for a, b in zip(p._prepared(imgs), direct)) and p._prefetched == {}
        assert torch.get_num_threads() == before
    finally:
        p.close_prefetch()
    assert torch.get_num_threads() == before and p._prefetch_pool is None
    p.close_prefetch()                                       # idempotent
    assert p.timings["host_prepare_s"] > 0.0


def test_no_crop_is_prepared_twice_when_the_next_group_is_prefetched_before_the_current_one_runs():
    """the order bench.py and stage/run.py use: prefetch(group g + 1), THEN the call on group g.  Round 4 kept one pending slot,
    which that order overwrites before the pick-up: every crop but the last group's was prepared twice."""
    p = _pipeline()
    groups = [[_crop(10 * g + i) for i in range(2)] for g in range(5)]
    calls = []
    orig = p._host_prepare
    p._host_prepare = lambda im: (calls.append((id(im), threading.current_thread().name)), orig(im))[1]
    main = threading.current_thread().name
    try:
        p.prefetch(groups[0])                                # (the first group of a run is prepared during the group before it)
        for g in range(5):
            if g + 1 < 5:
                p.prefetch(groups[g + 1])
            got = p._prepared(groups[g])
            assert len(got) == 2
    finally:
        p.close_prefetch()
    ids = [c[0] for c in calls]
    assert len(ids) == 10 and len(set(ids)) == 10            # every crop exactly once ...
    assert all(name != main for _, name in calls)            # ... and never on the calling thread
    assert p.timings["prefetch_hits"] == 10 and p.timings["host_prepare_n"] == 10
    # groups nobody comes for do not pile up
    p2 = _pipeline()
    try:
        keep = [[_crop(100 + g)] for g in range(p2._PREFETCH_MAX_GROUPS * 8 + 5)]
        for grp in keep:
            p2.prefetch(grp)
        assert len(p2._prefetched) == p2._PREFETCH_MAX_GROUPS * 8
        assert torch.equal(p2._prepared(keep[-1])[0], p2._host_prepare(keep[-1][0]))
    finally:
        p2.close_prefetch()


def test_groups_cut_differently_from_the_prefetches_still_prepare_every_crop_once():
    """bench.py's warm-up of 5 crops in groups of 4: the prefetch covers crops 4..7, the next call takes crop 4 alone and the
    timed run then starts with crops 5..8 ('20 of 34' -- five crops were prepared twice)"""
    p = _pipeline()
    crops = [_crop(200 + i) for i in range(9)]
    calls = []
    orig = p._host_prepare
    p._host_prepare = lambda im: (calls.append(id(im)), orig(im))[1]
    try:
        p.prefetch(crops[0:4])
        p.prefetch(crops[4:8])
        p._prepared(crops[0:4])
        p.prefetch(crops[8:9])
        p._prepared(crops[4:5])
        p._prepared(crops[5:9])
    finally:
        p.close_prefetch()
    assert len(calls) == 9 and len(set(calls)) == 9 and p.timings["prefetch_hits"] == 9


def test_a_crop_that_cannot_be_prepared_raises_at_pick_up():
    from PIL import Image
    p = _pipeline()
    bad = Image.fromarray(np.zeros((16, 16, 4), np.uint8), "RGBA")      # fully transparent: nothing to recentre on
    with pytest.raises(Exception) as direct:
        p._host_prepare(bad)
    try:
        grp = [_crop(3), bad]
        p.prefetch(grp)
        with pytest.raises(type(direct.value)):
            p._prepared(grp)
    finally:
        p.close_prefetch()


def test_guided_claim_sizes():
    """r3g.dist.guided_claim_size / WorkQueue.claim_guided: full launch groups while the list is long, smaller claims when it
    runs short -- BASELINE.json configs[1] (8 crops) on 8 GPUs is one object per rank, not two ranks with four each"""
    from r3g import dist as rdist
    g = rdist.guided_claim_size
    assert g(8, 4, 8) == 1 and g(64, 4, 8) == 4 and g(20, 4, 8) == 3 and g(9, 4, 8) == 2 and g(1, 4, 8) == 1
    assert g(0, 4, 8) == 1 and g(5, 4, 1) == 4 and g(3, 4, 1) == 3          # (claim_many clips an empty list to [])
    # a queue without a process group: one "rank", groups of 4 then the rest
    q = rdist.WorkQueue(10)
    assert q.remaining() == 10
    assert q.claim_guided(4) == [0, 1, 2, 3] and q.claim_guided(4) == [4, 5, 6, 7] and q.claim_guided(4) == [8, 9]
    assert q.remaining() == 0 and q.claim_guided(4) == []
    # eight ranks taking turns on 8 objects: everybody gets one; on 40 objects the first claims are full groups
    q = rdist.WorkQueue(8)
    assert [q.claim_guided(4, world=8) for _ in range(8)] == [[i] for i in range(8)]
    q = rdist.WorkQueue(40)
    sizes = []
    while True:
        c = q.claim_guided(4, world=8)
        if not c:
            break
        sizes.append(len(c))
    assert sizes[:3] == [4, 4, 4] and sizes[-1] == 1 and sum(sizes) == 40 and sorted(sizes, reverse=True) == sizes
