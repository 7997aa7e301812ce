"""Deterministic marching-cubes test volumes (shared by the fixture generator and the tests).

A-D are the integer-defined vectors of SURVEY.md section 4.3: int64 arithmetic, then one cast
to float32, so any language reproduces identical bits.
"""
import numpy as np


def _ijk(n):
    I, J, K = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    return I.astype(np.int64), J.astype(np.int64), K.astype(np.int64)


def golden_volume(name):
    """-> (float32 volume, level)"""
    if name == "A":
        I, J, K = _ijk(65)
        return (400 - ((I - 32) ** 2 + (J - 32) ** 2 + (K - 32) ** 2)).astype(np.float32), 0.5
    if name == "B":
        I, J, K = _ijk(33)
        h = ((I * 73856093) ^ (J * 19349663) ^ (K * 83492791)) % 2001 - 1000
        return (h.astype(np.float32) / np.float32(1000)).astype(np.float32), 0.0
    if name == "C":
        n = 48
        I, J, K = _ijk(n)
        return (4 * n * n - (4 * (I - n // 2) ** 2 + 9 * (J - n // 3) ** 2 + 25 * (K - n // 2) ** 2)
                ).astype(np.float32), 0.5
    if name == "D":
        I, J, K = _ijk(257)
        return (10000 - ((I - 128) ** 2 + (J - 128) ** 2 + (K - 128) ** 2)).astype(np.float32), 0.5
    raise KeyError(name)


def _smooth(rng, shape, passes):
    v = rng.standard_normal(shape)
    for _ in range(passes):  # separable [1 2 1]/4 box smoothing, no scipy dependency
        for ax in range(3):
            v = (np.roll(v, 1, ax) + 2 * v + np.roll(v, -1, ax)) / 4
    return v


def small_volumes(seed=20260925):
    """dict name -> float32 volume, plus 'level_<name>' entries.  Ragged shapes, NaNs, exact zeros,
    level outside the data range and no-surface cases included."""
    rng = np.random.default_rng(seed)
    out = {}

    def add(name, vol, level):
        out[name] = np.ascontiguousarray(vol, np.float32)
        out["level_" + name] = np.float64(level)

    add("noise_9", rng.standard_normal((9, 9, 9)), 0.0)
    add("noise_ragged", rng.standard_normal((5, 11, 17)), 0.1)
    add("noise_thin", rng.standard_normal((2, 2, 31)), 0.0)
    add("smooth_12", _smooth(rng, (12, 12, 12), 2) * 10, 0.0)
    add("smooth_ragged", _smooth(rng, (7, 18, 13), 3) * 50, 0.01)
    add("ints", rng.integers(-2, 3, (10, 9, 8)), 0.0)            # many exact-zero face tests
    add("ints_level", rng.integers(0, 5, (8, 8, 8)), 2.0)        # corners == level
    nanv = rng.standard_normal((8, 9, 10))
    nanv[rng.random(nanv.shape) < 0.05] = np.nan
    add("nan", nanv, 0.0)
    add("single_voxel", np.pad(np.ones((1, 1, 1)), 2), 0.5)      # V=6 F=8 octahedron
    two = np.zeros((6, 6, 9)); two[1, 1, 6] = 1; two[4, 4, 1] = 1
    add("two_blobs", two, 0.5)                                    # numbering follows axis-0 first
    add("plane", np.tile(np.array([0, 0, 1, 1], np.float32)[:, None, None], (1, 4, 4)), 0.5)
    eq = np.zeros((3, 3, 3)); eq[1, 1, 1] = 0.0; eq -= 1; eq[1, 1, 1] = 0.0
    add("lone_equal_below", eq, 0.0)                              # -> RuntimeError (no surface)
    eq2 = np.ones((3, 3, 3)); eq2[1, 1, 1] = 0.0
    add("lone_equal_above", eq2, 0.0)                             # zero-size octahedron kept
    add("outside_level", rng.standard_normal((4, 4, 4)), 100.0)  # -> ValueError
    add("minimal", np.array([[[0, 1], [1, 0]], [[1, 0], [0, 1]]], np.float32), 0.5)  # case 13
    return out


def cube_zoo(per_index=16, n_int=2500, n13=1500, seed=99):
    """[N,2,2,2] float32 single cells: every one of the 256 sign patterns with random magnitudes,
    small integers (exact ties in the ambiguity tests), and a dense sample of the two case-13
    patterns (alternating corners) with log-uniform magnitudes so that all 13.x sub-tilings occur."""
    rng = np.random.default_rng(seed)
    # corner k of Lewiner's numbering -> (z,y,x) of the 2x2x2 array
    pos = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]

    def pattern(idx, mag):
        c = np.empty((2, 2, 2))
        for k, p in enumerate(pos):
            c[p] = mag[k] if (idx >> k) & 1 else -mag[k]
        return c

    cubes = []
    for idx in range(256):
        for _ in range(per_index):
            cubes.append(pattern(idx, np.abs(rng.standard_normal(8)) + 1e-3))
    for _ in range(n13):
        cubes.append(pattern(90 if rng.random() < 0.5 else 165, np.exp(rng.uniform(-3, 3, 8))))
    a = np.stack(cubes)
    b = rng.integers(-3, 4, (n_int, 2, 2, 2))
    return np.concatenate([a, b]).astype(np.float32)
