"""cv2.INTER_AREA restated (cv2 is not in the image): the product's gathered-taps form against the oracle's running-integral form and
against the definition of the pixel-area relation, the enlarging branch's properties, and what PIL's BOX filter (rounds 1-2) was
off by."""
import os
import sys
import time

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))

from hy3dgen.shapegen import preprocessors as P  # noqa: E402
from oracle import hy3d_torch as H  # noqa: E402


def _definition(arr, w, h):
    """the area relation by brute force in float64: every destination pixel = the mean of the source over its footprint"""
    Hs, Ws, C = arr.shape
    sy, sx = Hs / h, Ws / w
    out = np.zeros((h, w, C))
    for i in range(h):
        for j in range(w):
            acc, area = np.zeros(C), 0.0
            for y in range(int(np.floor(i * sy)), min(int(np.ceil((i + 1) * sy)), Hs)):
                wy = min((i + 1) * sy, y + 1) - max(i * sy, y)
                for x in range(int(np.floor(j * sx)), min(int(np.ceil((j + 1) * sx)), Ws)):
                    wx = min((j + 1) * sx, x + 1) - max(j * sx, x)
                    acc += wy * wx * arr[y, x]
                    area += wy * wx
            out[i, j] = acc / area
    return out


@pytest.mark.parametrize("shape,size", [((23, 31, 4), (17, 11)), ((40, 40, 3), (20, 10)), ((19, 19, 1), (19, 19))])
def test_shrinking_is_the_pixel_area_relation(shape, size):
    rng = np.random.default_rng(sum(shape))
    arr = rng.integers(0, 256, shape, dtype=np.uint8)
    w, h = size
    want = _definition(arr, w, h)
    got = P.resize_area_u8(arr, w, h)
    assert got.shape == (h, w, shape[2]) and got.dtype == np.uint8
    assert np.abs(got.astype(np.float64) - want).max() <= 0.5 + 1e-3          # the rounded exact value
    assert np.abs(H.resize_area(arr, w, h).astype(int) - got.astype(int)).max() <= 1      # float32 vs float64 at a .5 tie
    if w == shape[1] and h == shape[0]:
        assert np.array_equal(got, arr)


def test_exact_halving_rounds_halves_up_like_opencvs_fast_path():
    arr = np.zeros((4, 4, 1), np.uint8)
    arr[0, 0] = 2                                   # block sums 2, 0, 0, 0 -> means 0.5, 0, 0, 0
    arr[2:, 2:] = [[[1], [2]], [[3], [4]]]          # 10 / 4 = 2.5
    got = P.resize_area_u8(arr, 2, 2)[..., 0]
    assert got.tolist() == [[1, 0], [0, 3]]         # half-even would give 0 and 2
    assert np.array_equal(H.resize_area(arr, 2, 2)[..., 0], got)


def test_integer_ratio_is_the_block_mean():
    rng = np.random.default_rng(5)
    arr = rng.integers(0, 256, (64, 48, 4), dtype=np.uint8)
    got = P.resize_area_u8(arr, 16, 16)
    want = np.rint(arr.reshape(16, 4, 16, 3, 4).astype(np.float64).mean(axis=(1, 3)))
    assert np.abs(got - want).max() <= 1


@pytest.mark.parametrize("shape,size", [((20, 30, 4), (45, 30)), ((7, 9, 3), (20, 16)), ((16, 16, 1), (17, 17))])
def test_enlarging_branch_product_equals_oracle_and_behaves(shape, size):
    rng = np.random.default_rng(sum(size))
    arr = rng.integers(0, 256, shape, dtype=np.uint8)
    w, h = size
    got = P.resize_area_u8(arr, w, h)
    assert np.array_equal(got, H.resize_area(arr, w, h))                        # integer arithmetic on both sides: identical
    assert got.shape == (h, w, shape[2])
    flat = np.full(shape, 137, np.uint8)
    assert np.array_equal(P.resize_area_u8(flat, w, h), np.full((h, w, shape[2]), 137, np.uint8))     # constants survive
    assert got.min() >= arr.min() and got.max() <= arr.max()                    # two non-negative taps: no overshoot
    twice = P.resize_area_u8(np.arange(16, dtype=np.uint8).reshape(1, 16, 1).repeat(4, 0) * 10, 32, 8)
    assert np.all(np.diff(twice[0, :, 0].astype(int)) >= 0)                     # a ramp stays monotone


def test_channels_are_filtered_independently_unlike_pil():
    """cv2 resizes the four channels separately; PIL premultiplies RGBA by alpha first: colours under alpha 0 bleed in cv2"""
    arr = np.zeros((8, 8, 4), np.uint8)
    arr[:, :4] = (200, 10, 10, 255)
    arr[:, 4:] = (10, 200, 10, 0)                                               # transparent, but it has a colour
    got = P.resize_area_u8(arr, 4, 4)                                           # column pairs (2, 3) | (4, 5) do not mix: use 8 -> 1
    one = P.resize_area_u8(arr, 1, 1)[0, 0]
    assert tuple(one) == (105, 105, 10, 128) or tuple(one) == (105, 105, 10, 127)
    pil = np.asarray(Image.fromarray(arr, "RGBA").resize((1, 1), Image.BOX))[0, 0]
    assert pil[0] > 190 and pil[1] < 20                                         # PIL: only the opaque half contributes colour
    assert got.shape == (4, 4, 4)


def test_what_the_box_filter_was_off_by_and_the_cost():
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (437, 391, 3), dtype=np.uint8)
    area = P.resize_area_u8(noise, 268, 300)
    box = np.stack([np.asarray(Image.fromarray(noise[..., c]).resize((268, 300), Image.BOX)) for c in range(3)], -1)
    d = np.abs(area.astype(int) - box.astype(int))
    assert d.mean() > 10                                                        # 0 / 1 weights against fractional coverage
    crop = np.zeros((470, 450, 4), np.uint8)
    crop[..., :3] = rng.integers(0, 256, (470, 450, 3))
    crop[..., 3] = 255
    t0 = time.perf_counter()
    for _ in range(5):
        P.resize_area_u8(crop, 416, 435)
    ms = 1000 * (time.perf_counter() - t0) / 5
    assert ms < 500, ms      # a guard against an accidental dense-matrix version (640 ms), not a benchmark: ~10-30 ms here


def test_area_relation_properties_over_random_shapes():
    """whatever the ratio: constants survive, values stay inside the input's range, and the image mean is preserved to rounding
    when shrinking (every source pixel's total weight is out/in on each axis)"""
    rng = np.random.default_rng(11)
    for _ in range(25):
        H0, W0 = int(rng.integers(2, 60)), int(rng.integers(2, 60))
        h, w = int(rng.integers(1, H0 + 1)), int(rng.integers(1, W0 + 1))
        arr = rng.integers(0, 256, (H0, W0, 2), dtype=np.uint8)
        out = P.resize_area_u8(arr, w, h)
        assert out.shape == (h, w, 2) and out.min() >= arr.min() and out.max() <= arr.max()
        assert abs(out.astype(np.float64).mean() - arr.astype(np.float64).mean()) <= 0.5 + 127.5 * (1.0 / max(h * w, 1)) ** 0.5
        flat = np.full((H0, W0, 2), int(rng.integers(0, 256)), np.uint8)
        assert np.array_equal(P.resize_area_u8(flat, w, h), np.full((h, w, 2), flat[0, 0, 0], np.uint8))
        grow_h, grow_w = H0 + int(rng.integers(1, 20)), W0 + int(rng.integers(1, 20))
        big = P.resize_area_u8(arr, grow_w, grow_h)
        assert big.shape == (grow_h, grow_w, 2) and np.array_equal(big, H.resize_area(arr, grow_w, grow_h))


def test_antialiased_bilinear_resize_against_the_real_torch_operator():
    """conditioner_transform's resize (round 5: numpy / scipy.sparse, so that the host-preparation thread runs no torch operator)
    against F.interpolate(mode="bilinear", antialias=True) itself -- enlarging (the path's 512 -> 518), shrinking, ragged
    shapes: one float32 ulp"""
    import torch
    import torch.nn.functional as F
    from hy3dgen.shapegen.preprocessors import conditioner_transform, resize_bilinear_aa, IMAGENET_MEAN, IMAGENET_STD
    rng = np.random.default_rng(5)
    for (H, W, nh, nw) in [(64, 64, 70, 70), (512, 512, 518, 518), (512, 512, 224, 224), (300, 200, 518, 345), (100, 130, 37, 51),
                           (77, 77, 77, 77)]:
        x = rng.random((3, H, W), dtype=np.float32)
        a = resize_bilinear_aa(x, nh, nw)
        b = F.interpolate(torch.from_numpy(x)[None], size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)[0].numpy()
        assert a.shape == b.shape and a.dtype == np.float32
        assert np.abs(a - b).max() <= 2.4e-7, (H, W, nh, nw, np.abs(a - b).max())
    # the whole transform on a non-square image: resize of the short side, centre crop, normalisation
    img = torch.from_numpy(rng.random((2, 3, 90, 120), dtype=np.float32) * 2 - 1)
    got = conditioner_transform(img, 70)
    xr = F.interpolate((img + 1) / 2, size=(70, int(120 * 70 / 90)), mode="bilinear", antialias=True, align_corners=False)
    left = (xr.shape[-1] - 70) // 2
    ref = (xr[:, :, :, left:left + 70] - torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    assert got.shape == (2, 3, 70, 70) and got.dtype == torch.float32 and got.is_contiguous()
    assert torch.allclose(got, ref, atol=2e-6)
