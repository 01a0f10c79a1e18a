"""SURVEY.md §8 f-1 (and the audio front-end of f-4, at the end): the host preprocessing right in front of the hot path, on the GPU.

  * `resize_longest_side`  = `ResizeLongestSide.apply_image` (segment_anything/utils/transforms.py:27-34,102-113):
                             torchvision `resize(to_pil_image(img), (newh, neww))` = Pillow BILINEAR resample
  * `sam_image`            = that + `sam_preprocess` normalise / pad (utils/refer_seg.py:560-570,588-593)
  * `clip_image`           = `CLIPImageProcessor.preprocess` (shortest edge -> 224 BICUBIC, optional centre crop,
                             x/255, normalise) + `F.interpolate(..., (224, 224), bilinear)` (utils/refer_seg.py:578-587)

Pillow resamples 8-bit images in fixed point (Resample.c): this module builds the per-output-index windows and
int32 coefficients on the host exactly as `precompute_coeffs` / `normalize_coeffs_8bpc` do (IEEE double, same
operation order), the HIP kernels do the two integer passes -- uint8 results are bit-identical to Pillow's.
Images stay on the device from the decoded uint8 HWC tensor to the model inputs.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .evalops import SAM_PIXEL_MEAN, SAM_PIXEL_STD, sam_preprocess

PRECISION_BITS = 32 - 8 - 2          # Resample.c
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # openai/clip-vit-large-patch14 preprocessor_config.json
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _bilinear(x):
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


def _bicubic(x, a=-0.5):
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1,
                    np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


_FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


@lru_cache(maxsize=64)
def pil_coeffs(in_size: int, out_size: int, filt: str) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` for the box (0, in_size):
    -> bounds int32 [out, 2] (first input index, tap count), coeffs int32 [out, ksize]."""
    f, support0 = _FILTERS[filt]
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size        # box edges are C floats
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)        # (int) truncates toward zero
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):                                                 # sequential sum, as the C loop
        w = np.where(x < xmax, f((x + xmin - center + 0.5) * ss), 0.0)
        kk[:, x] = w
        ww = ww + w
    nz = ww != 0.0
    kk[nz] = kk[nz] / ww[nz, None]
    half = np.where(kk < 0, -0.5, 0.5)
    ki = (half + kk * float(1 << PRECISION_BITS)).astype(np.int64).astype(np.int32)
    return np.stack([xmin, xmax], 1).astype(np.int32), ki


def _tables(in_size, out_size, filt, device):
    b, k = pil_coeffs(in_size, out_size, filt)
    return torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1]


def pil_resize_u8(image_hwc_u8: torch.Tensor, out_hw: Sequence[int], filt: str = "bilinear") -> torch.Tensor:
    """`PIL.Image.resize((w, h), filt)` of a uint8 [H, W, C] device image -> uint8 [h, w, C] device image."""
    if not image_hwc_u8.is_cuda or image_hwc_u8.dtype != torch.uint8 or image_hwc_u8.dim() != 3:
        raise ValueError("pil_resize_u8 wants a uint8 [H, W, C] device tensor (there is no CPU fallback)")
    lib = _lib.load()
    img = image_hwc_u8.contiguous()
    H, W, Cc = (int(v) for v in img.shape)
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dev = img.device
    out = torch.empty(oh, ow, Cc, dtype=torch.uint8, device=dev)
    xb = xk = yb = yk = tmp = None
    kx = ky = 0
    if ow != W:
        xb, xk, kx = _tables(W, ow, filt, dev)
    if oh != H:
        yb, yk, ky = _tables(H, oh, filt, dev)
    if ow != W and oh != H:
        tmp = torch.empty(H, ow, Cc, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.anyref_op_pil_resample_u8(st, _ptr(img), H, W, Cc, _ptr(tmp), _ptr(out), ow, oh, _ptr(xb), _ptr(xk), kx,
                                       _ptr(yb), _ptr(yk), ky)
    if rc != 0:
        raise RuntimeError("pil_resample_u8: " + lib.anyref_op_last_error().decode())
    return out


def get_preprocess_shape(oldh: int, oldw: int, long_side_length: int) -> Tuple[int, int]:
    """transforms.py:102-113."""
    scale = long_side_length * 1.0 / max(oldh, oldw)
    newh, neww = oldh * scale, oldw * scale
    return int(newh + 0.5), int(neww + 0.5)


def resize_longest_side(image_hwc_u8: torch.Tensor, target_length: int = 1024) -> torch.Tensor:
    """`ResizeLongestSide(target_length).apply_image` (transforms.py:27-34)."""
    return pil_resize_u8(image_hwc_u8, get_preprocess_shape(int(image_hwc_u8.shape[0]), int(image_hwc_u8.shape[1]),
                                                            target_length), "bilinear")


def sam_image(image_hwc_u8: torch.Tensor, sam_image_size: int = 1024, pixel_mean=SAM_PIXEL_MEAN, pixel_std=SAM_PIXEL_STD):
    """utils/refer_seg.py:588-593: -> (sam_image f32 [3, S, S], sam_resized_size (h, w))."""
    r = resize_longest_side(image_hwc_u8, sam_image_size)
    return sam_preprocess(r, sam_image_size, pixel_mean, pixel_std), (int(r.shape[0]), int(r.shape[1]))


def clip_image(image_hwc_u8: torch.Tensor, size: int = 224, resize_wo_crop: bool = True, mean=CLIP_MEAN, std=CLIP_STD):
    """utils/refer_seg.py:578-587 -> f32 [3, size, size].  `resize_wo_crop` = the datasets' `clip_resize_wo_crop`
    (True: no centre crop, the [224, w'] image is squeezed to 224 x 224 by bilinear interpolation)."""
    H, W = int(image_hwc_u8.shape[0]), int(image_hwc_u8.shape[1])
    # HF get_resize_output_image_size(shortest_edge = size, default_to_square = False)
    short, long = (W, H) if W <= H else (H, W)
    new_short, new_long = size, int(size * long / short)
    oh, ow = (new_long, new_short) if W <= H else (new_short, new_long)
    r = pil_resize_u8(image_hwc_u8, (oh, ow), "bicubic")
    y0 = x0 = 0
    h, w = oh, ow
    if not resize_wo_crop:                               # HF center_crop to (size, size); sizes here are >= size
        y0, x0, h, w = (oh - size) // 2, (ow - size) // 2, size, size
    lib = _lib.load()
    out = torch.empty(3, size, size, dtype=torch.float32, device=r.device)
    st = C.c_void_p(torch.cuda.current_stream(r.device).cuda_stream)
    rc = lib.anyref_op_clip_finish(st, _ptr(r), oh, ow, y0, x0, h, w, size, (C.c_float * 3)(*mean), (C.c_float * 3)(*std),
                                   _ptr(out))
    if rc != 0:
        raise RuntimeError("clip_finish: " + lib.anyref_op_last_error().decode())
    return out


# ---------------------------------------------------------------------------------------------------------------
# SURVEY.md §8 f-4, the audio front-end in front of the ImageBind trunk (model/ImageBind/data.py:28-64,114-161):
# waveform -> 3 clips of 2 s -> Kaldi log-mel filterbank (torchaudio.compliance.kaldi.fbank, the reference's options)
# -> [mel, 204] padded / cut -> Normalize(-4.268, 9.138), one HIP launch pair per clip.  File decoding and resampling
# to 16 kHz (torchaudio.load / functional.resample, data.py:133-137) stay with the caller.  No CPU fallback.
# ---------------------------------------------------------------------------------------------------------------
AUDIO_MEAN, AUDIO_STD = -4.268, 9.138          # data.py:121-122


def kaldi_mel_banks(num_bins: int, padded_window: int, sample_freq: float, low_freq: float = 20.0,
                    high_freq: float = 0.0) -> np.ndarray:
    """torchaudio.compliance.kaldi.get_mel_banks (no VTLN warp) + the zero Nyquist column fbank() pads on:
    float32 [num_bins, padded_window / 2 + 1] (host; built in double, same operation order)."""
    nfft = padded_window // 2
    if high_freq <= 0.0:
        high_freq += 0.5 * sample_freq
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    lo, hi = mel(low_freq), mel(high_freq)
    delta = (hi - lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
    m = mel(sample_freq / padded_window * np.arange(nfft, dtype=np.float64))[None, :]
    banks = np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))
    return np.concatenate([banks, np.zeros((num_bins, 1))], axis=1).astype(np.float32)


@lru_cache(maxsize=8)
def _fbank_tables(num_bins, padded, sample_rate, device):
    i = np.arange(padded, dtype=np.float64)
    tw = np.concatenate([np.cos(2.0 * np.pi * i / padded), np.sin(2.0 * np.pi * i / padded)])
    return (torch.from_numpy(kaldi_mel_banks(num_bins, padded, float(sample_rate))).to(device),
            torch.from_numpy(tw).to(device))


def waveform2melspec(waveform: torch.Tensor, sample_rate: int = 16000, num_mel_bins: int = 128, target_length: int = 204,
                     mean: float = AUDIO_MEAN, std: float = AUDIO_STD) -> torch.Tensor:
    """`waveform2melspec` + `Normalize(mean, std)` (data.py:28-64,152-153) of one clip: float32 [C, T] device waveform at
    `sample_rate` -> float32 [1, num_mel_bins, target_length] on the device."""
    if not waveform.is_cuda or waveform.dtype != torch.float32 or waveform.dim() != 2:
        raise ValueError("waveform2melspec wants a float32 [C, T] device tensor (there is no CPU fallback)")
    if sample_rate != 16000:
        raise ValueError("the HIP filterbank is built for 16 kHz input (25 ms = 400 samples -> 512-point transform), "
                         "the rate data.py:118 resamples to")
    lib = _lib.load()
    w = waveform.contiguous()
    Cn, T = (int(v) for v in w.shape)
    dev = w.device
    banks, tw = _fbank_tables(num_mel_bins, 512, sample_rate, dev)
    out = torch.empty(1, num_mel_bins, target_length, dtype=torch.float32, device=dev)
    scratch = torch.empty(1, dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.anyref_op_kaldi_fbank(st, _ptr(w), Cn, T, 400, 160, 512, C.c_float(0.97), _ptr(banks), num_mel_bins, _ptr(tw),
                                   _ptr(scratch), target_length, C.c_float(mean), C.c_float(std), _ptr(out))
    if rc != 0:
        raise RuntimeError("kaldi_fbank: " + lib.anyref_op_last_error().decode())
    return out


def constant_clip_timepoints(duration: float, clip_duration: float = 2.0, clips_per_video: int = 3):
    """pytorchvideo `ConstantClipsPerVideoSampler(clip_duration, clips_per_video)` as `get_clip_timepoints`
    (data.py:66-75) drives it: evenly spread (start, end) seconds, exact rational arithmetic."""
    from fractions import Fraction
    dur, clip = Fraction(duration), Fraction(clip_duration)
    step = max(dur - clip, 0) / max(clips_per_video - 1, 1)
    return [(float(step * i), float(step * i + clip)) for i in range(clips_per_video)]


def load_and_transform_audio_data(waveforms, device=None, num_mel_bins: int = 128, target_length: int = 204,
                                  sample_rate: int = 16000, clip_duration: float = 2, clips_per_video: int = 3,
                                  mean: float = AUDIO_MEAN, std: float = AUDIO_STD) -> torch.Tensor:
    """`load_and_transform_audio_data` (data.py:114-161) with decoded waveforms in place of paths: a list of float32
    [C, T] tensors already at `sample_rate` -> float32 [len, clips_per_video, 1, num_mel_bins, target_length] on the device
    (the `audios` item the model takes: [1, 3, 1, 128, 204])."""
    if waveforms is None:
        return None
    outs = []
    for w in waveforms:
        w = w.to(device if device is not None else w.device, dtype=torch.float32)
        clips = []
        for t0, t1 in constant_clip_timepoints(w.shape[1] / sample_rate, clip_duration, clips_per_video):
            clip = w[:, int(t0 * sample_rate): int(t1 * sample_rate)]
            clips.append(waveform2melspec(clip, sample_rate, num_mel_bins, target_length, mean, std))
        outs.append(torch.stack(clips, dim=0))
    return torch.stack(outs, dim=0)
