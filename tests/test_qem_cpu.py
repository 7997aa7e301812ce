"""Geometric contract of the quadric edge-collapse FaceReducer (reference src/2d_to_3d_models/run.py:93-94: upstream
MeshLab quadric edge collapse to 40 000 faces with boundary / normal / topology preservation), checked on the CPU through
tests/emu/qem_emu.cpp -- the product's own per-element bodies and round loop (csrc/qem_core.h, qem_driver.h) in host loops.
Bit parity with MeshLab's sequential queue is not a goal; the contract is: face budget met to within 1 %, closed meshes
stay closed 2-manifolds of the same genus, boundaries stay boundaries, no inverted faces, small two-sided distance."""
import numpy as np
import pytest

import emu_qem
import mesh_metrics as mm
from oracle import mc as omc


def blob(n, seed=0):
    rng = np.random.default_rng(seed)
    I, J, K = np.meshgrid(*(np.linspace(-1, 1, n),) * 3, indexing="ij")
    f = np.zeros((n, n, n))
    for _ in range(5):
        c = rng.uniform(-0.45, 0.45, 3)
        r = rng.uniform(0.25, 0.45)
        f = np.maximum(f, 1.0 - ((I - c[0]) ** 2 + (J - c[1]) ** 2 + (K - c[2]) ** 2) / r ** 2)
    return (f - 0.5).astype(np.float32)


@pytest.fixture(scope="module")
def blob_mesh():
    v, f = omc.marching_cubes(blob(72), 0.0)
    return v, f


@pytest.mark.parametrize("ratio", [0.5, 0.1, 0.05])
def test_closed_surface_budget_topology_and_distance(blob_mesh, ratio):
    v, f = blob_mesh
    _, cnt0 = mm.edge_face_counts(f)
    assert (cnt0 == 2).all()                                   # the input is a closed manifold
    target = int(len(f) * ratio)
    v2, f2, rounds = emu_qem.reduce_faces(v, f, target)
    assert 0.99 * target <= len(f2) <= target, (len(f2), target)
    assert f2.min() == 0 and f2.max() == len(v2) - 1 and len(np.unique(f2)) == len(v2)   # compact, all referenced
    _, cnt = mm.edge_face_counts(f2)
    assert (cnt == 2).all()                                    # still closed, every edge shared by exactly two faces
    assert mm.euler(len(v2), f2) == mm.euler(len(v), f)        # same genus / number of components
    assert (f2[:, 0] != f2[:, 1]).all() and (f2[:, 1] != f2[:, 2]).all() and (f2[:, 0] != f2[:, 2]).all()
    # orientation is kept: the signed volume keeps its sign and (almost) its value
    vol0, vol1 = mm.signed_volume(v, f), mm.signed_volume(v2, f2)
    assert vol0 * vol1 > 0 and abs(vol1 - vol0) <= {0.5: 0.01, 0.1: 0.03, 0.05: 0.08}[ratio] * abs(vol0)
    diag = np.linalg.norm(v.max(0) - v.min(0))
    hmax, hmean = mm.hausdorff(v, f, v2, f2)
    bound = {0.5: 0.004, 0.1: 0.01, 0.05: 0.03}[ratio]
    assert hmax <= bound * diag, (hmax / diag, rounds)
    assert hmean <= 0.25 * bound * diag


def test_decimation_is_deterministic_and_idempotent_below_budget(blob_mesh):
    v, f = blob_mesh
    a = emu_qem.reduce_faces(v, f, 3000)
    b = emu_qem.reduce_faces(v, f, 3000)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])
    c = emu_qem.reduce_faces(a[0], a[1], 3000)                 # already within the budget: untouched
    assert np.array_equal(c[0], a[0]) and np.array_equal(c[1], a[1]) and c[2] == 0


def test_open_surface_keeps_its_boundary():
    """a height field over a square: the outline must stay the square (boundary vertices only slide along the boundary)"""
    n = 60
    x, y = np.meshgrid(np.linspace(0, 1, n), np.linspace(0, 1, n), indexing="ij")
    z = 0.1 * np.sin(6 * x) * np.cos(5 * y)
    v = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    f = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int32)
    v2, f2, _ = emu_qem.reduce_faces(v, f, 600)
    assert 594 <= len(f2) <= 600
    e, cnt = mm.edge_face_counts(f2)
    assert set(np.unique(cnt)) <= {1, 2}
    bv = np.unique(e[cnt == 1])
    on_outline = (np.abs(v2[bv, 0]) < 1e-4) | (np.abs(v2[bv, 0] - 1) < 1e-4) | (np.abs(v2[bv, 1]) < 1e-4) | \
                 (np.abs(v2[bv, 1] - 1) < 1e-4)
    assert on_outline.all()
    for corner in ([0, 0], [0, 1], [1, 0], [1, 1]):             # the four corners survive
        assert (np.abs(v2[:, :2] - corner).max(1) < 1e-4).any()
    assert mm.euler(len(v2), f2) == 1                           # still a disc
    assert (mm.face_normals(v2, f2)[:, 2] > 0).all()            # no face flipped over
    hmax, _ = mm.hausdorff(v, f, v2, f2)
    assert hmax <= 0.02


def test_marching_cubes_zero_area_faces_do_not_block_the_decimation():
    """skimage keeps zero-area faces (coincident vertices with distinct ids): golden volume A has them"""
    from mc_volumes import golden_volume
    vol, level = golden_volume("A")
    v, f = omc.marching_cubes(vol, level)
    v2, f2, _ = emu_qem.reduce_faces(v, f, 1500)
    assert 1485 <= len(f2) <= 1500
    hmax, _ = mm.hausdorff(v, f, v2, f2)
    assert hmax <= 0.01 * np.linalg.norm(v.max(0) - v.min(0))


def test_two_components_stay_two_components():
    a = blob(40, 1)
    vol = np.concatenate([a, a[::-1]], axis=0)
    v, f = omc.marching_cubes(vol, 0.0)
    v2, f2, _ = emu_qem.reduce_faces(v, f, len(f) // 10)
    assert mm.euler(len(v2), f2) == mm.euler(len(v), f)
    _, cnt = mm.edge_face_counts(f2)
    assert (cnt == 2).all()


def test_impossible_budget_relaxes_shape_rules_but_never_topology(blob_mesh):
    """five spheres in 100 faces: the shape rules (slivers, turning angle) give way step by step, the link condition
    never does -- the result is still a closed manifold of the same genus, as small as the topology allows"""
    v, f = blob_mesh
    v2, f2, _ = emu_qem.reduce_faces(v, f, 100)
    _, cnt = mm.edge_face_counts(f2)
    assert (cnt == 2).all() and mm.euler(len(v2), f2) == mm.euler(len(v), f)
    assert len(f2) <= 100 or len(f2) < 0.03 * len(f)
