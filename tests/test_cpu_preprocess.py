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
