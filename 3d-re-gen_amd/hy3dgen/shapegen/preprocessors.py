"""ImageProcessorV2 (upstream hy3dgen/shapegen/preprocessors.py): recentre the object on its alpha bounding
box, composite on white, resize to `size`, -> image in [-1,1] and mask.  Host-side, tiny.
cv2 is not available in this image: INTER_AREA / INTER_CUBIC / INTER_NEAREST are replaced by PIL
BOX / BICUBIC / NEAREST (documented deviation; pixel-level parity with cv2 is not attainable here)."""
import numpy as np
import torch
from PIL import Image


def _resize(arr, w, h, resample):
    if arr.ndim == 3 and arr.shape[2] == 1:
        return np.asarray(Image.fromarray(arr[..., 0]).resize((w, h), resample))[..., None]
    return np.asarray(Image.fromarray(arr).resize((w, h), resample))


def array_to_tensor(np_array):
    t = torch.tensor(np.ascontiguousarray(np_array)).float() / 255 * 2 - 1
    return t.permute(2, 0, 1)[None].contiguous()


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
        result[x2_min:x2_min + h2, y2_min:y2_min + w2] = _resize(image[x_min:x_max, y_min:y_max], w2, h2, Image.BOX)
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


def conditioner_transform(image, image_size, value_range=(-1, 1)):
    """conditioner.ImageEncoder.forward up to the model call: map to [0,1], torchvision
    Resize(image_size, BILINEAR, antialias=True) + CenterCrop(image_size) + Normalize.  [B,3,H,W] -> same."""
    low, high = value_range
    x = (image.float() - low) / (high - low)
    _, _, h, w = x.shape
    s = image_size / min(h, w)
    nh, nw = (image_size, int(w * s)) if h <= w else (int(h * s), image_size)   # torchvision Resize truncates
    x = torch.nn.functional.interpolate(x, size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
    top, left = (nh - image_size) // 2, (nw - image_size) // 2
    x = x[:, :, top:top + image_size, left:left + image_size]
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return ((x - mean) / std).contiguous()
