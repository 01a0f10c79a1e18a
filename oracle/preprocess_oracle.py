"""CPU restatement of the preprocessing in front of the hot path.  TEST INFRASTRUCTURE ONLY.

The arithmetic lives in un-vendored third-party code the reference calls:
  * `ResizeLongestSide.apply_image` (model/segment_anything/utils/transforms.py:27-34) -> torchvision
    `resize(to_pil_image(img), size)` -> Pillow `Image.resize(size[::-1], BILINEAR)`;
  * `CLIPImageProcessor.preprocess` (utils/refer_seg.py:578-580; transformers==4.31.0, requirements.txt:29) ->
    Pillow BICUBIC shortest-edge resize, `image * (1 / 255)` in float64 cast to float32, `(image - mean) / std` in
    float32; then `F.interpolate(bilinear, align_corners=False)` (:581-587).
Pillow's published algorithm (src/libImaging/Resample.c; Pillow 12.2.0 is what this image has) is restated below in
numpy integers.  Pinned by `tests/golden/preprocess_pil.npz`: outputs of Pillow / the HF image processor / torch
themselves, made by `tests/golden/make_golden_preprocess.py` in the build container.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


def precompute_coeffs(in_size: int, out_size: int, filt: str):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc, scalar loops as in the C source."""
    f, support0 = FILTERS[filt]
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ww, ss = 0.0, 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_resize_u8(img: np.ndarray, out_hw, filt: str = "bilinear") -> np.ndarray:
    """`np.array(Image.fromarray(img).resize((w, h), filt))` for uint8 [H, W, C]: horizontal pass, then vertical, each
    out = clip8((2^21 + sum pixel * coeff) >> 22) (ImagingResampleHorizontal_8bpc / Vertical_8bpc)."""
    H, W, C = img.shape
    oh, ow = out_hw
    out = img
    if ow != W:
        b, k = precompute_coeffs(W, ow, filt)
        tmp = np.zeros((H, ow, C), dtype=np.uint8)
        for xx in range(ow):
            x0, n = b[xx]
            ss = (1 << (PRECISION_BITS - 1)) + (out[:, x0:x0 + n, :].astype(np.int64) * k[xx, :n][None, :, None]).sum(1)
            tmp[:, xx, :] = np.clip(ss >> PRECISION_BITS, 0, 255)
        out = tmp
    if oh != H:
        b, k = precompute_coeffs(H, oh, filt)
        tmp = np.zeros((oh, out.shape[1], C), dtype=np.uint8)
        for yy in range(oh):
            y0, n = b[yy]
            ss = (1 << (PRECISION_BITS - 1)) + (out[y0:y0 + n].astype(np.int64) * k[yy, :n][:, None, None]).sum(0)
            tmp[yy] = np.clip(ss >> PRECISION_BITS, 0, 255)
        out = tmp
    return out


def get_preprocess_shape(oldh, oldw, long_side_length):
    """transforms.py:102-113."""
    scale = long_side_length * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def resize_longest_side(img: np.ndarray, target: int = 1024) -> np.ndarray:
    """transforms.py:27-34."""
    return pil_resize_u8(img, get_preprocess_shape(img.shape[0], img.shape[1], target), "bilinear")


def clip_preprocess(img: np.ndarray, size: int = 224, resize_wo_crop: bool = True,
                    mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)) -> torch.Tensor:
    """utils/refer_seg.py:578-587 -> f32 [3, size, size]."""
    H, W = img.shape[:2]
    short, long = (W, H) if W <= H else (H, W)
    new_short, new_long = size, int(size * long / short)
    oh, ow = (new_long, new_short) if W <= H else (new_short, new_long)
    r = pil_resize_u8(img, (oh, ow), "bicubic")
    if not resize_wo_crop:
        top, left = (oh - size) // 2, (ow - size) // 2
        r = r[top: top + size, left: left + size]
    x = torch.from_numpy(np.ascontiguousarray(r)).permute(2, 0, 1)
    x = (x.double() * (1 / 255)).float()
    x = (x - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    if resize_wo_crop:
        x = F.interpolate(x[None], size=(size, size), mode="bilinear", align_corners=False)[0]
    return x
