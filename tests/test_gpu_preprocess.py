"""SURVEY.md §8 f-1 on the GPU: Pillow's 8-bit resampling (ResizeLongestSide, CLIP's bicubic shortest-edge resize)
bit for bit, and the CLIP normalise + bilinear squeeze within float tolerance, against the oracle restatement and
against fixtures produced by Pillow / the HF image processor / torch themselves."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_preprocess as gp  # noqa: E402
from oracle import preprocess_oracle as PO  # noqa: E402

FX = np.load(os.path.join(HERE, "golden", "preprocess_pil.npz"))


@pytest.mark.parametrize("name", list(gp.CASES))
def test_sam_and_clip_inputs_against_pillow_hf_fixtures(name):
    from anyref_amd.preprocess import resize_longest_side, sam_image, clip_image
    img = gp.preprocess_inputs(name)
    dev = torch.from_numpy(img).cuda()
    r = resize_longest_side(dev, 1024).cpu().numpy()
    assert list(r.shape) == FX[name + ".sam_shape"].tolist()
    assert np.array_equal(r[::8, ::8], FX[name + ".sam_u8"]) and int(r.astype(np.int64).sum()) == int(FX[name + ".sam_sum"])
    assert np.array_equal(r, PO.resize_longest_side(img, 1024))                      # every pixel, vs the oracle
    x, size = sam_image(dev)                                                          # + normalise / pad, bit-exact
    from oracle import anyref_oracle as O
    assert size == tuple(r.shape[:2]) and torch.equal(x.cpu(), O.sam_preprocess(torch.from_numpy(r), 1024))
    for wo in (True, False):
        c = clip_image(dev, 224, resize_wo_crop=wo).cpu()
        ref = FX[f"{name}.clip_{'wo' if wo else 'crop'}"]
        d = float(np.abs(c.numpy()[:, ::5, ::5] - ref).max())
        want = PO.clip_preprocess(img, 224, resize_wo_crop=wo)
        d2 = float((c - want).abs().max())
        # crop path: integer resize + the processor's exact f64->f32 rescale / f32 normalise: bit-exact.
        # squeeze path: + one float bilinear interpolation (FMA contraction differs between CPU and GPU): 2 ulp at |x| < 3
        assert d <= (5e-7 if wo else 0.0) and d2 <= (5e-7 if wo else 0.0), (name, wo, d, d2)


@pytest.mark.parametrize("filt", ["bilinear", "bicubic"])
@pytest.mark.parametrize("shape,out", [((61, 47, 3), (200, 31)), ((5, 1, 3), (9, 7)), ((300, 500, 1), (224, 373)),
                                       ((2047, 1365, 3), (1024, 683)), ((64, 64, 4), (64, 17)), ((64, 64, 3), (17, 64))])
def test_pil_resample_edge_shapes_bit_exact(filt, shape, out):
    """up- and down-scaling, one pass skipped (equal width / height), 1 and 4 channels, a 1-pixel-wide source"""
    from anyref_amd.preprocess import pil_resize_u8
    rng = np.random.default_rng(shape[0] * 7 + out[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    got = pil_resize_u8(torch.from_numpy(img).cuda(), out, filt).cpu().numpy()
    assert np.array_equal(got, PO.pil_resize_u8(img, out, filt))


def test_preprocess_refusals():
    from anyref_amd.preprocess import pil_resize_u8
    with pytest.raises(ValueError):
        pil_resize_u8(torch.zeros(4, 4, 3, dtype=torch.uint8), (2, 2))        # host tensor: no CPU fallback
    with pytest.raises(ValueError):
        pil_resize_u8(torch.zeros(4, 4, 3).cuda(), (2, 2))                     # not uint8


@pytest.mark.parametrize("case", ["noise 5 s stereo", "tone + noise 2.0 s", "short 1.3 s", "long 9.7 s, DC offset"])
def test_audio_front_end_vs_oracle(case):
    """`load_and_transform_audio_data` / `waveform2melspec` (model/ImageBind/data.py:28-64,114-161) on the device against the
    numpy restatement of torchaudio.compliance.kaldi.fbank + the clip sampler (oracle/preprocess_oracle.py; PARITY
    UNPINNED: torchaudio 0.13.0 and pytorchvideo are not in the build container, the reference holds no fixture).
    Tolerance 1e-4 on the normalised log-mel (values in [-1.5, 1.5]; f64 sums on both sides, f32 log): clip mean
    removal over all channels, channel 0 analysed, 198 frames of a 2 s clip padded to 204 with Normalize(0), a clip
    shorter than 2 s (fewer frames), a long waveform (three spread clips), empty mel bins (log eps)."""
    from anyref_amd.preprocess import load_and_transform_audio_data, waveform2melspec
    from oracle import preprocess_oracle as PO
    import zlib
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    sr = 16000
    if case.startswith("noise"):
        w = (rng.standard_normal((2, 5 * sr)) * 0.1).astype(np.float32)
    elif case.startswith("tone"):
        t = np.arange(2 * sr) / sr
        w = (0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.01 * rng.standard_normal(2 * sr))[None].astype(np.float32)
    elif case.startswith("short"):
        w = (rng.standard_normal((1, int(1.3 * sr))) * 0.05).astype(np.float32)
    else:
        w = (rng.standard_normal((1, int(9.7 * sr))) * 0.2 + 0.35).astype(np.float32)
    ref = PO.load_and_transform_audio(w, sr)
    got = load_and_transform_audio_data([torch.from_numpy(w).cuda()])
    assert got.shape == (1, 3, 1, 128, 204) and got.dtype == torch.float32
    err = np.abs(got[0].cpu().numpy() - ref).max()
    print(f"    {case}: max abs err {err:.2e} on normalised log-mel in [{ref.min():.2f}, {ref.max():.2f}]")
    assert err <= 1e-4, (case, err)
    one = waveform2melspec(torch.from_numpy(w[:, : 2 * sr]).cuda())
    ref1 = (PO.waveform2melspec(w[:, : 2 * sr], sr, 128, 204) - np.float32(-4.268)) / np.float32(9.138)
    assert np.abs(one.cpu().numpy() - ref1).max() <= 1e-4
