"""GPU parity of the HIP marching-cubes path (through the C ABI) against the oracle and the golden
fixtures recorded from the real scikit-image kernel.  Bar: faces AND vertices bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

from mc_volumes import cube_zoo, golden_volume, small_volumes

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan]))


@pytest.fixture(scope="module")
def gmc():
    import torch
    from r3g import mc as gpu_mc

    def run(vol, level, classic=False):
        g = torch.from_numpy(np.ascontiguousarray(vol, np.float32)).cuda()
        v, f = gpu_mc.marching_cubes(g, level, use_classic=classic)
        return v.cpu().numpy(), f.cpu().numpy()
    return run


@pytest.mark.parametrize("name", ["A", "B", "C", "D"])
def test_golden_vectors(name, gmc, golden_dir):
    with open(os.path.join(golden_dir, "mc_sha.json")) as f:
        g = json.load(f)[name]
    vol, level = golden_volume(name)
    v, f = gmc(vol, level)
    assert (len(v), len(f)) == (g["V"], g["F"])
    assert f.dtype == np.int32 and v.dtype == np.float32
    assert sha(f) == g["faces_sha"]
    assert sha(v) == g["verts_sha"]


def test_classic_tables(gmc, golden_dir):
    with open(os.path.join(golden_dir, "mc_sha.json")) as f:
        g = json.load(f)["B_classic"]
    vol, level = golden_volume("B")
    v, f = gmc(vol, level, classic=True)
    assert (len(v), len(f), sha(f), sha(v)) == (g["V"], g["F"], g["faces_sha"], g["verts_sha"])


def test_small_fixtures_and_errors(gmc, golden_dir):
    d = np.load(os.path.join(golden_dir, "mc_small.npz"))
    vols = small_volumes()
    for k in [k for k in vols if not k.startswith("level_")]:
        level = float(vols["level_" + k])
        if "e_" + k in d.files:
            exc = {"ValueError": ValueError, "RuntimeError": RuntimeError}[str(d["e_" + k])]
            with pytest.raises(exc):
                gmc(vols[k], level)
            continue
        v, f = gmc(vols[k], level)
        assert np.array_equal(f, d["f_" + k]), k
        assert bits_equal(v, d["v_" + k]), k


def test_cube_zoo(gmc, golden_dir):
    d = np.load(os.path.join(golden_dir, "mc_cubes.npz"))
    cubes = cube_zoo()
    fo = np.concatenate([[0], np.cumsum(d["nf"].astype(np.int64) * 3)])
    vo = np.concatenate([[0], np.cumsum(d["nv"].astype(np.int64) * 3)])
    for i, c in enumerate(cubes):
        try:
            v, f = gmc(c, 0.0)
        except (RuntimeError, ValueError):
            v, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
        assert len(f) == d["nf"][i] and len(v) == d["nv"][i], i
        assert np.array_equal(f.reshape(-1), d["f"][fo[i]:fo[i + 1]]), i
        assert bits_equal(v.reshape(-1), d["v"][vo[i]:vo[i + 1]]), i


def test_random_volumes_vs_oracle(gmc):
    from oracle import mc as omc
    rng = np.random.default_rng(1234)
    for it in range(24):
        shape = tuple(int(x) for x in rng.integers(2, 70, 3))
        vol = rng.standard_normal(shape).astype(np.float32)
        if it % 3 == 1:
            vol = np.round(vol * 2).astype(np.float32)   # exact ties
        if it % 3 == 2:
            for _ in range(2):
                for ax in range(3):
                    vol = (np.roll(vol, 1, ax) + 2 * vol + np.roll(vol, -1, ax)) / 4
        ov, of = omc.marching_cubes(vol, 0.0)
        v, f = gmc(vol, 0.0)
        assert np.array_equal(f, of), shape
        assert bits_equal(v, ov), shape


@pytest.mark.parametrize("shape", [(2, 2, 257), (5, 19, 257), (4, 9, 513), (3, 40, 769)])
def test_row_kernel_shapes_vs_oracle(gmc, shape):
    """Row length % 256 == 0 takes the row-marching classify kernel (4 cells per lane, float-domain sign test,
    LDS lane packing): dense noise (every cell active), exact ties with the level, levels that are / are not
    float32 values, infinities -- all against the oracle, bit for bit."""
    from oracle import mc as omc
    rng = np.random.default_rng(hash(shape) % 1000)
    noise = rng.standard_normal(shape).astype(np.float32)
    ties = np.round(noise * 2).astype(np.float32) / 2
    smooth = noise.copy()
    for _ in range(3):
        for ax in range(3):
            smooth = (np.roll(smooth, 1, ax) + 2 * smooth + np.roll(smooth, -1, ax)) / 4
    smooth = smooth.astype(np.float32)
    spiky = smooth.copy()
    spiky[0, 0, 5] = np.inf
    spiky[-1, -1, -7] = -np.inf
    cases = [(noise, 0.0), (noise, 0.1), (ties, 0.5), (ties, 0.25), (smooth, 0.0), (smooth, float(np.float32(0.01))),
             (smooth, 0.01), (smooth, 1e-300), (spiky, 0.0)]
    for vol, level in cases:
        ov, of = omc.marching_cubes(vol, level)
        v, f = gmc(vol, level)
        assert np.array_equal(f, of), (shape, level)
        assert bits_equal(v, ov), (shape, level)


def test_row_kernel_range_errors(gmc):
    vol = np.zeros((3, 3, 257), np.float32)
    with pytest.raises(RuntimeError):          # level inside [min, max] but no crossing: "No surface found"
        gmc(vol, 0.0)
    with pytest.raises(ValueError):
        gmc(vol, 1.0)
    vol[1, 1, 100] = np.nan                     # NaN disables the range check, as numpy's comparisons do
    with pytest.raises(RuntimeError):
        gmc(vol, 1.0)


def test_full_size_grid_vs_oracle(gmc):
    """257^3 (the reference's octree_resolution_hy=256, src/config.yaml:168): smooth random field with
    genuinely ambiguous cells, compared in full with the oracle."""
    from oracle import mc as omc
    rng = np.random.default_rng(5)
    lo = rng.standard_normal((33, 33, 33)).astype(np.float32)
    import torch
    t = torch.from_numpy(lo)[None, None]
    vol = torch.nn.functional.interpolate(t, size=(257, 257, 257), mode="trilinear", align_corners=True)[0, 0]
    vol = vol.numpy().astype(np.float32)
    ov, of = omc.marching_cubes(vol, 0.0)
    v, f = gmc(vol, 0.0)
    assert len(of) > 100000
    assert np.array_equal(f, of)
    assert bits_equal(v, ov)


def test_full_size_closed_surface_property(gmc):
    """Size-independent property at the full grid: a sphere's mesh is a closed 2-manifold
    (every undirected edge in exactly two faces, Euler characteristic 2), vertices on the sphere."""
    vol, level = golden_volume("D")
    v, f = gmc(vol, level)
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).astype(np.int64), axis=1)
    key, cnt = np.unique(e[:, 0] * (len(v) + 1) + e[:, 1], return_counts=True)
    assert np.all(cnt == 2)
    assert len(v) - len(key) + len(f) == 2
    r = np.linalg.norm(v.astype(np.float64) - 128.0, axis=1)
    assert np.all(np.abs(r - np.sqrt(10000 - 0.5)) < 0.6)
    assert np.array_equal(np.unique(f), np.arange(len(v)))  # every vertex referenced, ids dense


def test_extract_mesh_matches_upstream_conventions():
    import torch
    from oracle import mc as omc
    from r3g import mc as gpu_mc
    vol, level = golden_volume("C")
    wv, wf = omc.hy3d_mesh(vol, level, bound=1.01)
    v, f = gpu_mc.extract_mesh(torch.from_numpy(vol).cuda(), mc_level=level, bounds=1.01)
    assert np.array_equal(f.cpu().numpy(), wf)
    assert bits_equal(v.cpu().numpy(), wv)


def test_repeat_calls_and_size_changes_are_deterministic(gmc):
    rng = np.random.default_rng(9)
    a = rng.standard_normal((40, 41, 42)).astype(np.float32)
    b = rng.standard_normal((9, 9, 9)).astype(np.float32)
    va, fa = gmc(a, 0.0)
    vb, fb = gmc(b, 0.0)
    va2, fa2 = gmc(a, 0.0)
    assert np.array_equal(fa, fa2) and bits_equal(va, va2)
    assert len(fb) > 0 and len(vb) > 0
