"""FaceReducer on the GPU (quadric edge collapse, csrc/qem_core.h + mesh_kernels.hip) through the C ABI: bit-identical to
the host run of the same per-element code (tests/emu/qem_emu.cpp -- the algorithm is a pure function of its input), and
the geometric contract at the size of a real extraction (257^3 grid)."""
import time

import numpy as np
import pytest
import torch

import emu_qem
import mesh_metrics as mm

pytestmark = pytest.mark.gpu


def _mc(vol, level=0.0):
    from r3g import mc
    return mc.marching_cubes(torch.from_numpy(np.ascontiguousarray(vol, np.float32)).cuda(), level)


def _blob(n, seed):
    from test_qem_cpu import blob
    return blob(n, seed)


@pytest.mark.parametrize("n,budget", [(48, 1500), (96, 5000), (96, 40000)])
def test_gpu_equals_the_host_run_bit_for_bit(n, budget):
    from r3g import meshops
    v, f = _mc(_blob(n, n))
    gv, gf = meshops.reduce_faces(v, f, budget)
    ev, ef, _ = emu_qem.reduce_faces(v.cpu().numpy(), f.cpu().numpy(), budget)
    assert np.array_equal(gf.cpu().numpy(), ef)
    assert np.array_equal(gv.cpu().numpy().view(np.uint32), ev.view(np.uint32))
    for _ in range(3):                                     # and stable from run to run
        a, b = meshops.reduce_faces(v, f, budget)
        assert torch.equal(a, gv) and torch.equal(b, gf)


def test_contract_at_the_reference_grid_size():
    """257^3 blob field (the reference's octree_resolution_hy: 256) -> 40 000 faces (upstream FaceReducer's default)"""
    from r3g import meshops
    from parity_support import report
    v, f = _mc(_blob(257, 3))
    assert f.shape[0] > 100000
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gv, gf = meshops.reduce_faces(v, f, 40000)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    hv, hf = v.cpu().numpy(), f.cpu().numpy()
    ov, of = gv.cpu().numpy(), gf.cpu().numpy()
    assert 39600 <= len(of) <= 40000                       # within 1 % of the budget
    _, c0 = mm.edge_face_counts(hf)
    _, c1 = mm.edge_face_counts(of)
    assert (c0 == 2).all() and (c1 == 2).all()             # closed manifold in, closed manifold out
    assert mm.euler(len(ov), of) == mm.euler(len(hv), hf)
    vol0, vol1 = mm.signed_volume(hv, hf), mm.signed_volume(ov, of)
    assert vol0 * vol1 > 0 and abs(vol1 - vol0) <= 2e-3 * abs(vol0)
    diag = np.linalg.norm(hv.max(0) - hv.min(0))
    hmax, hmean = mm.hausdorff(hv, hf, ov, of, per_face=1)
    report("FaceReducer 257^3 blob: Hausdorff / bbox diagonal", hmax / diag, 4e-3)
    report("FaceReducer 257^3 blob: mean distance / bbox diagonal", hmean / diag, 5e-4)
    report("FaceReducer 257^3 blob: milliseconds (%d -> %d faces)" % (len(hf), len(of)), ms, 1e3)
    assert hmax <= 4e-3 * diag and hmean <= 5e-4 * diag     # < one voxel of the 257^3 grid at the worst point


def test_noise_mesh_and_stage_budget():
    """a rough field with many components and handles (what the synthetic-weight pipeline produces): budget, manifoldness
    and genus per component are kept; floaters first, as the stage does"""
    from r3g import meshops
    g = torch.Generator().manual_seed(5)
    lo = torch.randn(1, 1, 20, 20, 20, generator=g)
    vol = torch.nn.functional.interpolate(lo, size=(129, 129, 129), mode="trilinear", align_corners=True)[0, 0]
    v, f = _mc(vol.numpy())
    v, f = meshops.remove_degenerate(*meshops.remove_floaters(v, f, 0.005))
    gv, gf = meshops.reduce_faces(v, f, 40000)
    assert gf.shape[0] <= 40000
    e0, c0 = mm.edge_face_counts(f.cpu().numpy())
    e1, c1 = mm.edge_face_counts(gf.cpu().numpy())
    assert set(np.unique(c1)) <= set(np.unique(c0))        # no new non-manifold or boundary edges appear
    assert mm.euler(gv.shape[0], gf.cpu().numpy()) == mm.euler(v.shape[0], f.cpu().numpy())
