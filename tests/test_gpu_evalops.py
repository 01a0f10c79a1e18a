"""SURVEY.md §8 f-1 / f-2 on the GPU against the oracle: integer counts and exact-f32 pixels, bit for bit."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import anyref_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _case(n, h, w, seed, ignore_frac=0.1, empty_target=False):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(n, h, w, generator=g) * 3
    logits[logits.abs() < 1e-3] = 0.0           # sigmoid(x) > 0.5 and x > 0 agree away from 0 < x < 1.2e-7
    gt = (torch.rand(n, h, w, generator=g) > 0.6).to(torch.uint8)
    if empty_target:
        gt.zero_()
    gt[torch.rand(n, h, w, generator=g) < ignore_frac] = 255
    return logits, gt


@pytest.mark.parametrize("n,h,w", [(1, 480, 640), (1, 333, 517), (3, 64, 64), (1, 1, 7), (2, 1024, 1024)])
def test_iou_counts_match_reference_formula(n, h, w):
    from anyref_amd.evalops import intersection_and_union
    logits, gt = _case(n, h, w, seed=n * 1000 + h)
    got = intersection_and_union(logits.cuda(), gt.cuda(), per_mask=True)
    for i in range(n):
        want = O.eval_mask_counts(logits[i:i + 1], gt[i:i + 1])
        for a, b in zip(got, want):
            assert torch.equal(a[i].cpu(), b), (a, b)
    # the reference call flattens whatever it is given into one histogram
    for a, b in zip(intersection_and_union(logits.cuda(), gt.cuda()), O.eval_mask_counts(logits, gt)):
        assert torch.equal(a.cpu(), b), (a, b)


def test_iou_counts_edge_cases():
    from anyref_amd.evalops import intersection_and_union
    # all ignored, no-object target, exact zeros (sigmoid(0) = 0.5 is NOT > 0.5)
    logits, gt = _case(1, 40, 50, seed=5, ignore_frac=1.1)
    for a, b in zip(intersection_and_union(logits.cuda(), gt.cuda()), O.eval_mask_counts(logits, gt)):
        assert torch.equal(a.cpu(), b)
    logits, gt = _case(1, 40, 50, seed=6, empty_target=True, ignore_frac=0.0)
    inter, union, tgt = intersection_and_union(logits.cuda(), gt.cuda())
    for a, b in zip((inter, union, tgt), O.eval_mask_counts(logits, gt)):
        assert torch.equal(a.cpu(), b)
    assert tgt[1].item() == 0
    z = torch.zeros(1, 8, 8)
    ones = torch.ones(1, 8, 8, dtype=torch.uint8)
    for a, b in zip(intersection_and_union(z.cuda(), ones.cuda()), O.eval_mask_counts(z, ones)):
        assert torch.equal(a.cpu(), b)
    with pytest.raises(ValueError):
        intersection_and_union(z.cuda(), ones.cuda(), K=3)
    with pytest.raises(ValueError):
        intersection_and_union(z.cuda(), ones[:, :4].cuda())
    with pytest.raises(RuntimeError):
        intersection_and_union(z, ones)            # host logits: no CPU fallback


@pytest.mark.parametrize("h,w,S", [(1024, 683, 1024), (768, 1024, 1024), (224, 224, 224), (17, 5, 32)])
def test_sam_preprocess_bit_exact(h, w, S):
    from anyref_amd.evalops import sam_preprocess
    g = torch.Generator().manual_seed(h * 7 + w)
    img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
    got = sam_preprocess(img.cuda(), S).cpu()
    want = O.sam_preprocess(img, S)
    assert got.shape == want.shape and torch.equal(got, want)
    with pytest.raises(RuntimeError):
        sam_preprocess(torch.zeros(S + 1, 4, 3, dtype=torch.uint8).cuda(), S)
