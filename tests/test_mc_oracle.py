"""Pins oracle/mc_lewiner.c to the reference's marching-cubes dependency (scikit-image Lewiner).

Golden fixtures were produced by tools/make_mc_golden.py from the real compiled skimage kernel;
tests marked `skimage` additionally re-run that kernel live (build container only).
Bar: faces AND vertices bit-exact (the restatement reproduces the kernel's double arithmetic).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from mc_volumes import cube_zoo, golden_volume, small_volumes
from oracle import mc, mc_skimage


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a) & np.isnan(b)  # NaN payloads are not part of the contract
    return bool(np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan]))


@pytest.fixture(scope="module")
def sha_table(golden_dir):
    with open(os.path.join(golden_dir, "mc_sha.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["A", "B", "C", "D"])
def test_survey_golden_vectors(name, sha_table):
    vol, level = golden_volume(name)
    g = sha_table[name]
    assert sha(vol) == g["in_sha"]
    v, f = mc.marching_cubes(vol, level)
    assert (len(v), len(f)) == (g["V"], g["F"])
    assert f.dtype == np.int32 and v.dtype == np.float32
    assert sha(f) == g["faces_sha"]
    assert sha(v) == g["verts_sha"]
    assert f[0].tolist() == g["f0"] and [float(x) for x in v[0]] == g["v0"]


def test_survey_values_match_recorded_table(sha_table):
    # the numbers SURVEY.md section 4.3 recorded independently during the survey session
    rec = {"A": (7470, 14936, "ca4e8afec52452e3", "306d1155dddd8231"),
           "B": (55405, 117647, "709d1c7d9fc59455", "236e2774fb7e0342"),
           "C": (7010, 13596, "570c0764748120fa", "52c799ff5afb7113"),
           "D": (188382, 376760, "d33ba4922708e5da", "6650695f8289f3da")}
    for k, (V, F, fs, vs) in rec.items():
        g = sha_table[k]
        assert (g["V"], g["F"], g["faces_sha"], g["verts_sha"]) == (V, F, fs, vs)


def test_classic_tables_same_kernel(sha_table):
    vol, level = golden_volume("B")
    v, f = mc.marching_cubes(vol, level, use_classic=True)
    g = sha_table["B_classic"]
    assert (len(v), len(f), sha(f), sha(v)) == (g["V"], g["F"], g["faces_sha"], g["verts_sha"])
    for name in "AC":  # convex smooth shapes never reach an ambiguous case
        vol, level = golden_volume(name)
        v1, f1 = mc.marching_cubes(vol, level, use_classic=True)
        v2, f2 = mc.marching_cubes(vol, level)
        assert np.array_equal(f1, f2) and bits_equal(v1, v2)


def _check_against(vols, res):
    names = [k for k in vols if not k.startswith("level_")]
    assert names
    for k in names:
        level = float(vols["level_" + k])
        if "e_" + k in res:
            with pytest.raises({"ValueError": ValueError, "RuntimeError": RuntimeError}[str(res["e_" + k])]):
                mc.marching_cubes(vols[k], level)
            continue
        v, f = mc.marching_cubes(vols[k], level)
        assert np.array_equal(f, res["f_" + k]), k
        assert bits_equal(v, res["v_" + k]), k


def test_small_volume_fixtures(golden_dir):
    d = np.load(os.path.join(golden_dir, "mc_small.npz"))
    vols = small_volumes()
    for k, a in vols.items():  # fixture inputs are reproducible from the seeded generator
        assert np.array_equal(a, d[k], equal_nan=True), k
    _check_against(vols, d)
    # behaviours SURVEY.md 4.3 lists
    assert str(d["e_lone_equal_below"]) == "RuntimeError"
    assert str(d["e_outside_level"]) == "ValueError"
    assert d["v_single_voxel"].shape == (6, 3) and d["f_single_voxel"].shape == (8, 3)
    assert d["v_lone_equal_above"].shape == (6, 3) and d["f_lone_equal_above"].shape == (8, 3)


def test_cube_zoo_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "mc_cubes.npz"))
    cubes = cube_zoo()
    assert np.array_equal(cubes, d["cubes"])
    fo = np.concatenate([[0], np.cumsum(d["nf"].astype(np.int64) * 3)])
    vo = np.concatenate([[0], np.cumsum(d["nv"].astype(np.int64) * 3)])
    seen_nf = set()
    for i, c in enumerate(cubes):
        try:
            v, f = mc.marching_cubes(c, 0.0)
        except (RuntimeError, ValueError):
            v, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
        assert len(f) == d["nf"][i] and len(v) == d["nv"][i], i
        assert np.array_equal(f.reshape(-1), d["f"][fo[i]:fo[i + 1]]), i
        assert bits_equal(v.reshape(-1), d["v"][vo[i]:vo[i + 1]]), i
        seen_nf.add(len(f))
    assert {1, 2, 3, 4, 5, 6, 8, 9, 10, 12} <= seen_nf  # every tiling size is exercised


def test_hy3d_mesh_conventions():
    # upstream rescale uses R+1 (not R) and export_to_trimesh reverses the winding
    vol, level = golden_volume("A")
    v, f = mc.marching_cubes(vol, level)
    wv, wf = mc.hy3d_mesh(vol, level, bound=1.01)
    assert np.array_equal(wf, f[:, ::-1])
    exp = (v.astype(np.float64) / 65 * 2.02 - 1.01).astype(np.float32)
    assert np.array_equal(wv, exp)
    # positive-inside field -> outward orientation -> positive signed volume
    a, b, c = (wv[wf[:, i]].astype(np.float64) for i in range(3))
    assert np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6 > 0


def test_invalid_inputs():
    with pytest.raises(ValueError):
        mc.marching_cubes(np.zeros((2, 2, 1), np.float32), 0)
    with pytest.raises(ValueError):
        mc.marching_cubes(np.zeros((20, 20), np.float32), 0)
    with pytest.raises(ValueError):
        mc.marching_cubes(np.zeros((4, 4, 4), np.float32), 1.0)


def test_float64_and_fortran_inputs_identical():
    rng = np.random.default_rng(3)
    vol = rng.standard_normal((7, 8, 9)).astype(np.float32)
    v0, f0 = mc.marching_cubes(vol, 0.0)
    v1, f1 = mc.marching_cubes(np.asfortranarray(vol.astype(np.float64)), 0.0)
    assert np.array_equal(f0, f1) and bits_equal(v0, v1)


@pytest.mark.skimage
def test_live_skimage_random_volumes():
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    for _ in range(6):
        shape = tuple(int(x) for x in rng.integers(2, 24, 3))
        vol = rng.standard_normal(shape).astype(np.float32)
        if rng.random() < 0.5:
            vol = np.round(vol * 2).astype(np.float32)
        try:
            sv, sf = mc_skimage.marching_cubes(vol, 0.0)
        except (RuntimeError, ValueError) as e:
            with pytest.raises(type(e)):
                mc.marching_cubes(vol, 0.0)
            continue
        v, f = mc.marching_cubes(vol, 0.0)
        assert np.array_equal(f, sf) and bits_equal(v, sv)
