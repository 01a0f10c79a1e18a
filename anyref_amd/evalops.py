"""The steps either side of the hot path, on the device.

`intersection_and_union` mirrors `utils/utils.py:79-91 intersectionAndUnionGPU` *including* the
`(torch.sigmoid(pred) > 0.5).int()` the eval scripts apply first (eval_referseg.py:189-208), fused
into one pass over the full-resolution logits; `sam_preprocess` mirrors
`utils/refer_seg.py:560-570` (normalise + zero-pad) on the uint8 image `ResizeLongestSide`
produced.  Both call libanyref_hip.so and fail loudly without it (no CPU fallback).
"""
import ctypes as C
from typing import Sequence, Tuple

import torch

from . import _lib

SAM_PIXEL_MEAN = (123.675, 116.28, 103.53)   # utils/refer_seg.py: pixel_mean / pixel_std
SAM_PIXEL_STD = (58.395, 57.12, 57.375)


def _ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def intersection_and_union(pred_logits: torch.Tensor, target: torch.Tensor, K: int = 2, ignore_index: int = 255,
                           per_mask: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (area_intersection, area_union, area_target) as float tensors [K], what the reference returns for
    `output = (sigmoid(pred_logits) > 0.5).int()` (it flattens everything it is given into one histogram).
    per_mask=True keeps the leading dimension of a [n, h, w] input: tensors [n, K]."""
    if K != 2 or ignore_index != 255:
        raise ValueError("the fused kernel covers the eval scripts' call: K=2, ignore_index=255")
    if not pred_logits.is_cuda:
        raise RuntimeError("intersection_and_union needs device logits (there is no CPU fallback)")
    if pred_logits.dim() not in (1, 2, 3) or tuple(pred_logits.shape) != tuple(target.shape):
        raise ValueError(f"pred {tuple(pred_logits.shape)} and target {tuple(target.shape)} must have the same 1-3-d shape")
    lib = _lib.load()
    n = pred_logits.shape[0] if (per_mask and pred_logits.dim() == 3) else 1
    x = pred_logits.to(torch.float32).contiguous().reshape(n, -1)
    t = target.to(x.device).reshape(n, -1).to(torch.uint8).contiguous()
    counts = torch.empty(n, 6, dtype=torch.int64, device=x.device)
    rc = lib.anyref_op_iou_counts(_stream(x.device), _ptr(x), _ptr(t), n, x.shape[1], _ptr(counts))
    if rc != 0:
        raise RuntimeError("iou_counts: " + lib.anyref_op_last_error().decode())
    c = counts.to(torch.float32)
    inter, out, tgt = c[:, 0:2], c[:, 2:4], c[:, 4:6]
    union = out + tgt - inter
    if per_mask and pred_logits.dim() == 3:
        return inter, union, tgt
    return inter[0], union[0], tgt[0]


def sam_preprocess(image_hwc_u8: torch.Tensor, sam_image_size: int = 1024, pixel_mean: Sequence[float] = SAM_PIXEL_MEAN,
                   pixel_std: Sequence[float] = SAM_PIXEL_STD) -> torch.Tensor:
    """uint8 [h, w, 3] (already resized so that max(h, w) <= sam_image_size) -> f32 [3, S, S]."""
    if not image_hwc_u8.is_cuda or image_hwc_u8.dtype != torch.uint8 or image_hwc_u8.dim() != 3 or image_hwc_u8.shape[2] != 3:
        raise ValueError("sam_preprocess wants a uint8 [h, w, 3] device tensor")
    lib = _lib.load()
    img = image_hwc_u8.contiguous()
    h, w = int(img.shape[0]), int(img.shape[1])
    out = torch.empty(3, sam_image_size, sam_image_size, dtype=torch.float32, device=img.device)
    mean = (C.c_float * 3)(*pixel_mean)
    std = (C.c_float * 3)(*pixel_std)
    rc = lib.anyref_op_sam_preprocess(_stream(img.device), _ptr(img), h, w, sam_image_size, mean, std, _ptr(out))
    if rc != 0:
        raise RuntimeError("sam_preprocess: " + lib.anyref_op_last_error().decode())
    return out
