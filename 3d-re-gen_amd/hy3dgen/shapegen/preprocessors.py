"""ImageProcessorV2 (upstream hy3dgen/shapegen/preprocessors.py): recentre the object on its alpha bounding
box, composite on white, resize to `size`, -> image in [-1,1] and mask.  Host-side, tiny.
cv2 is not available in this image.  The one resize that is not an identity on the reference's 512 x 512 crops -- the object's
bounding box to 85 % of the canvas with cv2.INTER_AREA -- is restated here (resize_area_u8): the pixel-area relation OpenCV
documents for shrinking (every source pixel weighted by the fraction of it that a destination pixel covers, all four channels
independently) and, for enlarging, what OpenCV's INTER_AREA does instead -- two-tap interpolation with its "area" coefficients in
11-bit fixed point ([RECALLED] from OpenCV imgproc/resize.cpp; not pinned: no cv2 here).  PIL's BOX filter, used in rounds 1-2,
gives a source pixel weight 0 or 1 and filters RGBA with premultiplied alpha: up to 20 grey levels mean difference from the area
relation on high-frequency content (tests/test_preprocess_cpu.py).  INTER_CUBIC / INTER_NEAREST of the final resize stay PIL
BICUBIC / NEAREST: identities at the reference's crop size."""
import numpy as np
import torch
from PIL import Image


def _resize(arr, w, h, resample):
    if arr.ndim == 3 and arr.shape[2] == 1:
        return np.asarray(Image.fromarray(arr[..., 0]).resize((w, h), resample))[..., None]
    return np.asarray(Image.fromarray(arr).resize((w, h), resample))


def _area_operator(n_in, n_out):
    """sparse [n_out][n_in] float32: the fraction of destination pixel i's footprint [i s, (i + 1) s) that source pixel j covers,
    s = n_in / n_out >= 1 (OpenCV computeResizeAreaTab: partial cells at both ends, whole cells between, normalised by the
    footprint's width)"""
    from scipy.sparse import csr_matrix
    s = n_in / n_out
    lo = np.arange(n_out, dtype=np.float64) * s
    hi = np.minimum(lo + s, n_in)
    taps = int(np.ceil(s)) + 1
    j = np.floor(lo).astype(np.int64)[:, None] + np.arange(taps)[None, :]
    cover = np.clip(np.minimum(hi[:, None], j + 1.0) - np.maximum(lo[:, None], j), 0.0, 1.0) / (hi - lo)[:, None]
    keep = (cover > 0) & (j < n_in)
    rows = np.repeat(np.arange(n_out), taps).reshape(n_out, taps)
    return csr_matrix((cover[keep].astype(np.float32), (rows[keep], j[keep])), shape=(n_out, n_in))


def _linear_area_taps(n_in, n_out):
    """OpenCV's INTER_AREA when a dimension grows: two taps, source indices (s0, s1) and 11-bit coefficients (a0, a1) with
    fx = frac((d + 1) - (s0 + 1) n_out / n_in) (0 when negative), a1 = round(2048 fx), a0 = round(2048 (1 - fx)); at the last
    source pixel the pixel itself"""
    scale, inv = n_in / n_out, n_out / n_in
    d = np.arange(n_out)
    s0 = np.floor(d * scale).astype(np.int64)
    fx = ((d + 1) - (s0 + 1) * inv).astype(np.float32)
    fx = np.where(fx <= 0, np.float32(0), fx - np.floor(fx)).astype(np.float32)
    fx = np.where(s0 >= n_in - 1, np.float32(0), fx)
    s0 = np.minimum(s0, n_in - 1)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int32)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int32)
    return s0, np.minimum(s0 + 1, n_in - 1), a0, a1


def _apply(op_y, op_x, a):
    """rows then columns of [H][W][C] through two sparse operators (y: [h][H], x: [w][W])"""
    H, W, C = a.shape
    t = op_y @ a.reshape(H, W * C)                                             # [h][W C]
    h = t.shape[0]
    t = op_x @ np.ascontiguousarray(t.reshape(h, W, C).transpose(1, 0, 2)).reshape(W, h * C)      # [w][h C]
    return t.reshape(-1, h, C).transpose(1, 0, 2)


def resize_area_u8(arr, w, h):
    """cv2.resize(arr, (w, h), interpolation=cv2.INTER_AREA) for uint8 [H][W][C], restated (module docstring)"""
    from scipy.sparse import csr_matrix
    arr = np.ascontiguousarray(arr)
    H, W, C = arr.shape
    if H == 2 * h and W == 2 * w:               # OpenCV's fast path for exactly 2 x 2: integer mean, halves rounded UP
        a = arr.astype(np.int32)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    if w <= W and h <= H:                       # shrinking (or equal): the area relation, float32 accumulation, round half to even
        out = _apply(_area_operator(H, h), _area_operator(W, w), arr.astype(np.float32))
        return np.rint(out).clip(0, 255).astype(np.uint8)
    # OpenCV's 8-bit linear path: the horizontal pass at scale 2^11 (one sparse product) ...
    x0, x1, ax0, ax1 = _linear_area_taps(W, w)
    d = np.arange(w)
    op = csr_matrix((np.concatenate([ax0, ax1]), (np.concatenate([d, d]), np.concatenate([x0, x1]))), shape=(w, W), dtype=np.int32)
    rows = (op @ np.ascontiguousarray(arr.astype(np.int32).transpose(1, 0, 2)).reshape(W, H * C)).reshape(w, H, C)
    rows = np.ascontiguousarray(rows.transpose(1, 0, 2)) >> 4                  # [H][w][C]
    # ... then VResizeLinear for 8-bit: the two vertical taps are shifted separately before they are added
    y0, y1, ay0, ay1 = _linear_area_taps(H, h)
    out = ((rows[y0] * ay0[:, None, None]) >> 16) + ((rows[y1] * ay1[:, None, None]) >> 16)
    return ((out + 2) >> 2).clip(0, 255).astype(np.uint8)


def array_to_tensor(np_array):
    """uint8 [H][W][C] -> float32 [1][C][H][W] in [-1, 1].  numpy arithmetic (the same IEEE operations, in the same order, as
    upstream's `torch.tensor(a).float() / 255 * 2 - 1`: bit-identical), so that the host-preparation thread runs no torch
    operator -- torch's intra-op thread pool is process-wide (pipelines.py: prefetch)"""
    a = np.asarray(np_array).astype(np.float32) / np.float32(255) * np.float32(2) - np.float32(1)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)[None]))


class ImageProcessorV2:
    def __init__(self, size=512, border_ratio=None):
        self.size = size
        self.border_ratio = border_ratio

    @staticmethod
    def recenter(image, border_ratio=0.2):
        if image.shape[-1] == 4:
            mask = image[..., 3]
        else:
            mask = np.ones_like(image[..., 0:1]) * 255
            image = np.concatenate([image, mask], axis=-1)
            mask = mask[..., 0]
        H, W, C = image.shape
        size = max(H, W)
        result = np.zeros((size, size, C), dtype=np.uint8)
        coords = np.nonzero(mask)
        x_min, x_max = coords[0].min(), coords[0].max()
        y_min, y_max = coords[1].min(), coords[1].max()
        h, w = x_max - x_min, y_max - y_min
        if h == 0 or w == 0:
            raise ValueError("input image is empty")
        desired_size = int(size * (1 - border_ratio))
        scale = desired_size / max(h, w)
        h2, w2 = int(h * scale), int(w * scale)
        x2_min, y2_min = (size - h2) // 2, (size - w2) // 2
        result[x2_min:x2_min + h2, y2_min:y2_min + w2] = resize_area_u8(image[x_min:x_max, y_min:y_max], w2, h2)
        bg = np.ones((size, size, 3), dtype=np.uint8) * 255
        m = result[..., 3:].astype(np.float32) / 255
        rgb = result[..., :3] * m + bg * (1 - m)
        return rgb.clip(0, 255).astype(np.uint8), (m * 255).clip(0, 255).astype(np.uint8)

    def load_image(self, image, border_ratio=0.15, to_tensor=True):
        if isinstance(image, str):
            image = Image.open(image)
        image = np.asarray(image.convert("RGBA"))
        image, mask = self.recenter(image, border_ratio=border_ratio)
        image = _resize(image, self.size, self.size, Image.BICUBIC)
        mask = _resize(mask, self.size, self.size, Image.NEAREST)
        if to_tensor:
            image, mask = array_to_tensor(image), array_to_tensor(mask)
        return image, mask

    def __call__(self, image, border_ratio=0.15, to_tensor=True, **kwargs):
        if self.border_ratio is not None:
            border_ratio = self.border_ratio
        image, mask = self.load_image(image, border_ratio=border_ratio, to_tensor=to_tensor)
        return {"image": image, "mask": mask}


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _aa_bilinear_operator(n_in, n_out):
    """One axis of torch's F.interpolate(mode="bilinear", antialias=True, align_corners=False) (aten
    _upsample_bilinear2d_aa: the triangle filter, stretched by the scale when the axis shrinks, weights normalised per output
    sample; float32 throughout, as aten computes them for float32 input) as a sparse [n_out][n_in] float32 operator.
    Checked against the real operator, shrinking and enlarging: tests/test_preprocess_cpu.py."""
    from scipy.sparse import csr_matrix
    f32 = np.float32
    scale = f32(n_in) / f32(n_out)
    support = scale if scale >= 1 else f32(1)
    inv = f32(1) / scale if scale >= 1 else f32(1)
    i = np.arange(n_out, dtype=np.float32)
    center = scale * (i + f32(0.5))
    xmin = np.maximum((center - support + f32(0.5)).astype(np.int64), 0)
    xsize = np.minimum((center + support + f32(0.5)).astype(np.int64), n_in) - xmin
    taps = int(xsize.max())
    j = np.arange(taps, dtype=np.int64)[None, :]
    x = ((j + xmin[:, None]).astype(np.float32) - center[:, None] + f32(0.5)) * inv
    w = np.maximum(f32(1) - np.abs(x), f32(0)).astype(np.float32)
    w[j >= xsize[:, None]] = 0
    tot = w.sum(axis=1, dtype=np.float32)
    w = np.where(tot[:, None] != 0, w / np.where(tot == 0, f32(1), tot)[:, None], w).astype(np.float32)
    col = j + xmin[:, None]
    keep = j < xsize[:, None]
    rows = np.repeat(np.arange(n_out), taps).reshape(n_out, taps)
    return csr_matrix((w[keep], (rows[keep], col[keep])), shape=(n_out, n_in), dtype=np.float32)


def resize_bilinear_aa(x, nh, nw):
    """float32 [C][H][W] -> [C][nh][nw]: the width pass first, then the height pass (aten's order)"""
    C, H, W = x.shape
    t = (_aa_bilinear_operator(W, nw) @ np.ascontiguousarray(x.reshape(C * H, W).T)).T.reshape(C, H, nw)     # [C][H][nw]
    t = _aa_bilinear_operator(H, nh) @ np.ascontiguousarray(t.transpose(1, 0, 2)).reshape(H, C * nw)          # [nh][C nw]
    return np.ascontiguousarray(t.reshape(nh, C, nw).transpose(1, 0, 2)).astype(np.float32, copy=False)


def conditioner_transform(image, image_size, value_range=(-1, 1)):
    """conditioner.ImageEncoder.forward up to the model call: map to [0,1], torchvision
    Resize(image_size, BILINEAR, antialias=True) + CenterCrop(image_size) + Normalize.  [B,3,H,W] -> same.
    Host work in numpy / scipy.sparse (round 5): the thread that prepares the next launch group must not start torch's
    process-wide intra-op thread team beside the HIP runtime (pipelines.py: prefetch), so no torch operator computes here."""
    low, high = value_range
    a = image.detach().cpu().numpy() if isinstance(image, torch.Tensor) else np.asarray(image)
    x = (a.astype(np.float32) - np.float32(low)) / np.float32(high - low)
    _, _, h, w = x.shape
    s = image_size / min(h, w)
    nh, nw = (image_size, int(w * s)) if h <= w else (int(h * s), image_size)   # torchvision Resize truncates
    top, left = (nh - image_size) // 2, (nw - image_size) // 2
    mean = np.asarray(IMAGENET_MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(IMAGENET_STD, np.float32).reshape(3, 1, 1)
    out = np.stack([(resize_bilinear_aa(xb, nh, nw)[:, top:top + image_size, left:left + image_size] - mean) / std for xb in x])
    return torch.from_numpy(np.ascontiguousarray(out.astype(np.float32, copy=False)))
