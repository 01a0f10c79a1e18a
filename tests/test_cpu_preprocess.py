"""f-1 on the CPU: the oracle's Pillow restatement against outputs of Pillow / the HF image processor / torch
(tests/golden/preprocess_pil.npz), and the product's vectorised coefficient tables against the oracle's scalar ones."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_preprocess as gp  # noqa: E402
from oracle import preprocess_oracle as PO  # noqa: E402

FX = np.load(os.path.join(HERE, "golden", "preprocess_pil.npz"))


@pytest.mark.parametrize("name", list(gp.CASES))
def test_oracle_preprocess_against_pillow_and_hf(name):
    img = gp.preprocess_inputs(name)
    r = PO.resize_longest_side(img, 1024)
    assert list(r.shape) == FX[name + ".sam_shape"].tolist()
    assert np.array_equal(r[::8, ::8], FX[name + ".sam_u8"]) and int(r.astype(np.int64).sum()) == int(FX[name + ".sam_sum"])
    for wo in (True, False):
        x = PO.clip_preprocess(img, 224, resize_wo_crop=wo)
        ref = FX[f"{name}.clip_{'wo' if wo else 'crop'}"]
        assert x.shape == (3, 224, 224)
        d = np.abs(x.numpy()[:, ::5, ::5] - ref).max()
        assert d <= (1e-6 if wo else 0.0), (wo, d)          # the crop path has no float interpolation: bit-exact


def test_oracle_matches_installed_pillow_directly():
    """belt and braces on the GPU box too (same image): PIL itself, sizes that up- and down-scale"""
    from PIL import Image
    rng = np.random.default_rng(5)
    for (H, W), (oh, ow) in (((61, 47), (200, 31)), ((300, 500), (224, 373)), ((5, 1), (9, 7))):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        for filt, pf in (("bilinear", Image.BILINEAR), ("bicubic", Image.BICUBIC)):
            assert np.array_equal(PO.pil_resize_u8(img, (oh, ow), filt), np.array(Image.fromarray(img).resize((ow, oh), pf)))


@pytest.mark.parametrize("filt", ["bilinear", "bicubic"])
@pytest.mark.parametrize("sizes", [(640, 1024), (2000, 1024), (750, 576), (91, 1024), (224, 224), (500, 373), (1, 7)])
def test_product_coefficient_tables_equal_the_scalar_restatement(filt, sizes):
    from anyref_amd.preprocess import pil_coeffs, get_preprocess_shape
    b, k = pil_coeffs(sizes[0], sizes[1], filt)
    b2, k2 = PO.precompute_coeffs(sizes[0], sizes[1], filt)
    assert np.array_equal(b, b2) and np.array_equal(k, k2)
    assert get_preprocess_shape(480, 640, 1024) == PO.get_preprocess_shape(480, 640, 1024) == (768, 1024)


def test_audio_front_end_oracle_properties():
    """The numpy restatement of the audio front-end (model/ImageBind/data.py:28-64,114-161; torchaudio 0.13.0 kaldi.fbank +
    pytorchvideo's clip sampler, neither installed here: PARITY UNPINNED) against what the published algorithm implies:
    frame count under snip_edges, the padded frames after Normalize, DC / constant-offset invariance, a pure tone landing in
    the mel filter that covers it, the empty low filters at 128 bins / 512-point transform giving log(eps), and the sampler."""
    from oracle import preprocess_oracle as PO
    sr = 16000
    rng = np.random.default_rng(7)
    w = (rng.standard_normal((1, 2 * sr)) * 0.1).astype(np.float32)
    fb = PO.kaldi_fbank(w, sr)
    assert fb.shape == (1 + (2 * sr - 400) // 160, 128) == (198, 128)
    m = PO.waveform2melspec(w, sr, 128, 204)
    assert m.shape == (1, 128, 204) and np.all(m[0, :, 198:] == 0.0)
    assert np.abs(m[0, :, :198] - fb.T).max() < 2e-3                      # (the clip mean is removed first: f32 rounding of the input)
    out = PO.load_and_transform_audio(w, sr)
    assert out.shape == (3, 1, 128, 204)
    assert np.allclose(out[0, 0, :, 198:], (0.0 + 4.268) / 9.138)          # zero-padded frames after Normalize
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])  # a 2 s waveform: three identical clips
    # DC removal per frame: a constant offset changes nothing beyond f32 rounding of the input
    fb_off = PO.kaldi_fbank(w + np.float32(0.25), sr)
    assert np.abs(fb_off - fb).max() < 2e-3
    # a constant signal has no energy after DC removal: every filter at log(eps)
    const = PO.kaldi_fbank(np.full((1, sr), 0.5, dtype=np.float32), sr)
    assert np.allclose(const, np.log(np.float32(PO.KALDI_EPS)))
    # mel filters: triangles that tile [20 Hz, Nyquist]; the lowest are narrower than one FFT bin (31.25 Hz) and some are empty
    banks = PO.kaldi_mel_banks(128, 512, float(sr))
    assert banks.shape == (128, 257) and np.all(banks >= 0) and np.all(banks <= 1.0) and np.all(banks[:, 256] == 0)
    empty = np.where(banks.sum(1) == 0)[0]
    assert 0 < len(empty) < 8 and empty.max() < 16
    assert np.allclose(fb[:, empty], np.log(np.float32(PO.KALDI_EPS)))
    # 1 kHz tone: the strongest filter is the one whose triangle contains 1 kHz
    t = np.arange(2 * sr) / sr
    tone = PO.kaldi_fbank((0.5 * np.sin(2 * np.pi * 1000.0 * t))[None].astype(np.float32), sr)
    peak = int(np.bincount(tone.argmax(1)).argmax())
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    delta = (mel(8000.0) - mel(20.0)) / 129
    assert abs(peak - ((mel(1000.0) - mel(20.0)) / delta - 1)) <= 1.0
    # the sampler: clips_per_video windows spread evenly, the last one ending at the end
    assert PO.constant_clip_timepoints(5.0) == [(0.0, 2.0), (1.5, 3.5), (3.0, 5.0)]
    assert PO.constant_clip_timepoints(2.0) == [(0.0, 2.0)] * 3
    assert PO.constant_clip_timepoints(1.0) == [(0.0, 2.0)] * 3             # shorter than a clip: the slice just ends early
