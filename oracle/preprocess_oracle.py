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

The audio front-end at the end of the file (model/ImageBind/data.py:28-64,114-161) is PARITY UNPINNED:
torchaudio 0.13.0 and pytorchvideo are not installed here and the reference holds no fixture for it (see the section header).
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


# ---------------------------------------------------------------------------------------------------------------
# Audio front-end: waveform -> normalised log-mel clips, model/ImageBind/data.py:28-64,114-161.
#
# PARITY UNPINNED: the arithmetic lives in two third-party packages the build container does not have --
# torchaudio==0.13.0 (requirements.txt:26; `torchaudio.compliance.kaldi.fbank`, a restatement of Kaldi's
# feature-fbank) and pytorchvideo @ 28fe037 (requirements.txt:17; `ConstantClipsPerVideoSampler`) -- and the reference
# holds no fixture for it.  What follows restates their published algorithms for exactly the options the reference's
# call site passes (data.py:31-41: htk_compat=True, use_energy=False, window_type="hanning", dither=0.0,
# frame_length=25, frame_shift=10; everything else at torchaudio's defaults: preemphasis 0.97, remove_dc_offset,
# round_to_power_of_two, snip_edges, low_freq 20, high_freq 0 -> Nyquist, use_power, use_log_fbank, no VTLN warp).
# Accumulations are float64 here (torch: float32 rfft / mm); tests hold the GPU path to 1e-4 on the normalised log-mel.
# ---------------------------------------------------------------------------------------------------------------
KALDI_EPS = float(np.finfo(np.float32).eps)  # torchaudio _get_epsilon: torch.finfo(torch.float32).eps


def kaldi_mel_banks(num_bins: int, padded_window: int, sample_freq: float, low_freq: float = 20.0, high_freq: float = 0.0):
    """torchaudio.compliance.kaldi.get_mel_banks (vtln_warp = 1): [num_bins, padded_window / 2 + 1] float32 with the
    zero Nyquist column fbank() pads on (compliance/kaldi.py: `mel_energies = pad(mel_energies, (0, 1))`)."""
    nfft = padded_window // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1) * delta, mel_low + (b + 2) * delta
    m = mel(sample_freq / padded_window * np.arange(nfft, dtype=np.float64))[None, :]
    up, down = (m - left) / (center - left), (right - m) / (right - center)
    banks = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([banks, np.zeros((num_bins, 1))], axis=1).astype(np.float32)


def kaldi_fbank(waveform: np.ndarray, sample_rate: int = 16000, num_mel_bins: int = 128, frame_length_ms: float = 25.0,
                frame_shift_ms: float = 10.0, preemphasis: float = 0.97) -> np.ndarray:
    """torchaudio.compliance.kaldi.fbank for the reference's options: waveform [C, T] (channel 0 is used) ->
    [num_frames, num_mel_bins] float32 log-mel energies."""
    x = np.asarray(waveform, dtype=np.float32)[0].astype(np.float64)
    shift = int(sample_rate * frame_shift_ms * 0.001)
    win = int(sample_rate * frame_length_ms * 0.001)
    padded = 1 << (win - 1).bit_length()                      # round_to_power_of_two
    if x.shape[0] < win:
        return np.zeros((0, num_mel_bins), dtype=np.float32)
    m = 1 + (x.shape[0] - win) // shift                       # snip_edges
    idx = np.arange(m)[:, None] * shift + np.arange(win)[None, :]
    fr = x[idx]
    fr = fr - fr.mean(axis=1, keepdims=True)                  # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)    # replicate-padded shift by one
    fr = fr - preemphasis * prev
    n = np.arange(win, dtype=np.float64)
    fr = fr * (0.5 - 0.5 * np.cos(2.0 * np.pi * n / (win - 1)))   # torch.hann_window(win, periodic=False)
    fr = np.concatenate([fr, np.zeros((m, padded - win))], axis=1)
    power = np.abs(np.fft.rfft(fr, axis=1)) ** 2
    mel = power @ kaldi_mel_banks(num_mel_bins, padded, float(sample_rate)).astype(np.float64).T
    return np.log(np.maximum(mel, KALDI_EPS)).astype(np.float32)


def waveform2melspec(waveform: np.ndarray, sample_rate: int, num_mel_bins: int, target_length: int) -> np.ndarray:
    """data.py:28-64: subtract the clip mean (all channels), fbank, transpose, zero-pad / cut to target_length ->
    [1, num_mel_bins, target_length]."""
    w = np.asarray(waveform, dtype=np.float32)
    w = w - w.mean(dtype=np.float64).astype(np.float32)
    fb = kaldi_fbank(w, sample_rate, num_mel_bins).T
    p = target_length - fb.shape[1]
    if p > 0:
        fb = np.pad(fb, ((0, 0), (0, p)))
    elif p < 0:
        fb = fb[:, :target_length]
    return fb[None]


def constant_clip_timepoints(duration: float, clip_duration: float = 2.0, clips_per_video: int = 3):
    """pytorchvideo ConstantClipsPerVideoSampler driven by data.py:66-75: `clips_per_video` windows of `clip_duration`
    seconds spread evenly over [0, duration] (exact rational arithmetic, as the sampler's Fractions)."""
    from fractions import Fraction
    dur, clip = Fraction(duration), Fraction(clip_duration)
    max_start = max(dur - clip, 0)
    step = max_start / max(clips_per_video - 1, 1)
    return [(float(step * i), float(step * i + clip)) for i in range(clips_per_video)]


def load_and_transform_audio(waveform: np.ndarray, sample_rate: int = 16000, num_mel_bins: int = 128,
                             target_length: int = 204, clip_duration: float = 2.0, clips_per_video: int = 3,
                             mean: float = -4.268, std: float = 9.138) -> np.ndarray:
    """data.py:114-161 from an in-memory waveform [C, T] already at `sample_rate` (file decoding and resampling are the
    caller's): -> [clips_per_video, 1, num_mel_bins, target_length] float32, Normalize(mean, std) applied."""
    out = []
    for t0, t1 in constant_clip_timepoints(waveform.shape[1] / sample_rate, clip_duration, clips_per_video):
        clip = waveform[:, int(t0 * sample_rate): int(t1 * sample_rate)]
        out.append((waveform2melspec(clip, sample_rate, num_mel_bins, target_length) - np.float32(mean)) / np.float32(std))
    return np.stack(out).astype(np.float32)
