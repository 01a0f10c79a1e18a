"""The steps either side of the hot path, on the device.

`intersection_and_union` mirrors `utils/utils.py:79-91 intersectionAndUnionGPU` *including* the
`(torch.sigmoid(pred) > 0.5).int()` the eval scripts apply first (eval_referseg.py:189-208), fused
into one pass over the full-resolution logits; `sam_preprocess` mirrors
`utils/refer_seg.py:560-570` (normalise + zero-pad) on the uint8 image `ResizeLongestSide`
produced; `mask_iou` / `eval_fmeasure` mirror `utils/pyutils.py:163-236` (the AVS eval loop,
eval_avs_object.py:168-178) on one fused counting pass.  All call libanyref_hip.so and fail loudly
without it (no CPU fallback).
"""
import functools
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


# --- AVS metrics (utils/pyutils.py:163-236) ---------------------------------------------------------------
def _f32_keys(x: torch.Tensor) -> torch.Tensor:
    """f32 -> int64 key with the same order (so that bisection over *representable* floats is integer bisection)"""
    b = x.view(torch.int32).to(torch.int64)
    return torch.where(b < 0, -(b & 0x7FFFFFFF) - 1, b)   # -0.0 -> -1, +0.0 -> 0


def _least_logit(th: torch.Tensor, strict: bool) -> torch.Tensor:
    """smallest f32 x with sigmoid(x) >= th (strict: > th), elementwise, against torch's own f32 sigmoid (the
    function the reference thresholds); -inf where every x passes, +inf where none does."""
    n = th.numel()

    def sig(x):
        # ATen runs the SIMD sigmoid on whole vectors and a scalar one (which can differ in the last bit) on the
        # few elements left over at the END of a tensor; a mask has ~1e5..1e6 pixels, so the SIMD result is the
        # one to reproduce: pad so that no probe sits in that remainder
        pad = torch.zeros((n + 127) // 64 * 64)
        pad[:n] = x
        return torch.sigmoid(pad)[:n]

    ok = (lambda x: sig(x) > th) if strict else (lambda x: sig(x) >= th)
    lo = _f32_keys(torch.full((n,), -float("inf")))      # invariant: lo fails (or is -inf), hi passes
    hi = _f32_keys(torch.full((n,), float("inf")))
    neg_inf_ok = ok(torch.full((n,), -float("inf")))
    pos_inf_ok = ok(torch.full((n,), float("inf")))
    for _ in range(34):
        mid = (lo + hi) // 2
        good = ok(_keys_to_f32(mid))
        hi = torch.where(good, mid, hi)
        lo = torch.where(good, lo, mid)
    out = _keys_to_f32(hi)
    out = torch.where(neg_inf_ok, torch.full_like(out, -float("inf")), out)
    return torch.where(pos_inf_ok, out, torch.full_like(out, float("inf")))


def _keys_to_f32(k: torch.Tensor) -> torch.Tensor:
    b = torch.where(k < 0, (-(k + 1)) - 0x80000000, k)   # as a signed 32-bit pattern
    return b.to(torch.int32).view(torch.float32)


@functools.lru_cache(maxsize=8)
def _avs_cuts(pr_num: int):
    th = torch.linspace(0, 1 - 1e-10, pr_num)            # utils/pyutils.py:226 (f32: the end point rounds to 1.0)
    cuts = _least_logit(th, strict=False)
    cut_pred = float(_least_logit(torch.tensor([0.5]), strict=True)[0])
    return cuts, cut_pred


def _avs_counts(pred_logits: torch.Tensor, target: torch.Tensor, pr_num: int):
    if not pred_logits.is_cuda:
        raise RuntimeError("the AVS metrics need device logits (there is no CPU fallback)")
    if pred_logits.dim() != 3 or tuple(pred_logits.shape) != tuple(target.shape):
        raise ValueError(f"pred {tuple(pred_logits.shape)} and target {tuple(target.shape)} must both be [N, H, W]")
    if not 1 <= pr_num <= 255:
        raise ValueError("1 <= pr_num <= 255")
    lib = _lib.load()
    n = pred_logits.shape[0]
    x = pred_logits.to(torch.float32).contiguous().reshape(n, -1)
    t = target.to(x.device)
    if not bool(((t == 0) | (t == 1)).all()):
        raise ValueError("the fused AVS metrics cover binary ground truth (values 0 / 1), as the AVS loaders produce")
    t = t.reshape(n, -1).to(torch.uint8).contiguous()
    cuts, cut_pred = _avs_cuts(pr_num)
    cuts_d = cuts.to(x.device)
    conf = torch.empty(n, 4, dtype=torch.int64, device=x.device)
    hist = torch.empty(n, pr_num + 1, 2, dtype=torch.int64, device=x.device)
    rc = lib.anyref_op_avs_counts(_stream(x.device), _ptr(x), _ptr(t), n, x.shape[1], _ptr(cuts_d), pr_num,
                                  C.c_float(cut_pred), _ptr(conf), _ptr(hist))
    if rc != 0:
        raise RuntimeError("avs_counts: " + lib.anyref_op_last_error().decode())
    return conf.cpu(), hist.cpu()


def mask_iou(pred: torch.Tensor, target: torch.Tensor, eps: float = 1e-7, size_average: bool = True) -> torch.Tensor:
    """utils/pyutils.py:163-190 (same signature; like the reference, size_average is accepted and the mean is
    returned either way).  pred: logits [N, H, W] on the device; target: binary [N, H, W]."""
    conf, _ = _avs_counts(pred, target, 1)
    n, npix = pred.shape[0], pred.shape[-1] * pred.shape[-2]
    n00, n01, n10, n11 = conf[:, 0], conf[:, 1], conf[:, 2], conf[:, 3]
    empty_gt = (n01 + n11) == 0
    inter = torch.where(empty_gt, n00, n11)
    union = torch.where(empty_gt, torch.full_like(n11, npix), n11 + n10 + n01)
    return torch.sum(inter / (union + eps)) / n


def eval_fmeasure(pred: torch.Tensor, gt: torch.Tensor, measure_path=None, pr_num: int = 255) -> float:
    """utils/pyutils.py:193-236 Eval_Fmeasure (+ _eval_pr): the 255 threshold sweeps become suffix sums of one
    histogram; the F_beta arithmetic after the counts is the reference's, in f32.  measure_path is accepted and
    unused (the reference only creates an empty FMeasure.txt there)."""
    _, hist = _avs_counts(pred, gt, pr_num)
    beta2 = 0.3
    total, used = 0.0, 0
    score = torch.zeros(pr_num)
    # above[m, i, g] = pixels of label g passing threshold i  (bins b > i)
    above = hist.flip(1).cumsum(1).flip(1)[:, 1:, :]
    for m in range(hist.shape[0]):
        gsum = hist[m, :, 1].sum().float()
        if gsum == 0:
            continue
        tp = above[m, :, 1].float()
        passed = (above[m, :, 0] + above[m, :, 1]).float()
        prec, recall = tp / (passed + 1e-20), tp / (gsum + 1e-20)
        f = (1 + beta2) * prec * recall / (beta2 * prec + recall)
        f[f != f] = 0
        total = total + f
        used += 1
        score = total / used
    return score.max().item()


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
