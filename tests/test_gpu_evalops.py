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


# --- AVS metrics (utils/pyutils.py:163-236; eval_avs_object.py:168-178) ---------------------------------------
def _avs_case(n, h, w, seed, scale=3.0, empty=()):
    from anyref_amd import evalops as E
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(n, h, w, generator=g) * scale
    gt = (torch.rand(n, h, w, generator=g) > 0.6).to(torch.int64)
    for m in empty:
        gt[m].zero_()
    # plant the decision boundaries themselves: each cut and the float just below it
    cuts, cut_pred = E._avs_cuts(255)
    edge = torch.cat([cuts[1:], torch.nextafter(cuts[1:], torch.full_like(cuts[1:], -float("inf"))),
                      torch.tensor([cut_pred, 0.0, -0.0, 1e-8, 30.0, -30.0, float("inf"), -float("inf")])])
    flat = logits.view(n, -1)
    if flat.shape[1] >= 2 * edge.numel():   # (torch's CPU sigmoid treats the last few elements of a tensor with scalar
        for m in range(n):                  #  code that may round differently: keep the probes away from the end)
            flat[m, :edge.numel()] = edge
    return logits, gt


@pytest.mark.parametrize("n,h,w", [(1, 224, 224), (5, 224, 224), (2, 333, 517), (1, 1, 7), (1, 1024, 1024)])
def test_avs_mask_iou_and_fmeasure_match_reference_formulas(n, h, w):
    from anyref_amd.evalops import mask_iou, eval_fmeasure
    logits, gt = _avs_case(n, h, w, seed=n * 77 + w, empty=(n - 1,) if n > 1 else ())
    got = mask_iou(logits.cuda(), gt.cuda())
    want = O.avs_mask_iou(logits, gt)
    assert got.dtype == want.dtype and float(got) == float(want), (got, want)      # same counts, same f32 ops
    f_got = eval_fmeasure(logits.cuda(), gt.float().cuda(), None)
    f_want = O.avs_fmeasure(logits, gt.float())
    assert f_got == f_want, (f_got, f_want)


def test_avs_counts_are_the_reference_threshold_sweep():
    """every one of the 255 (tp, #passed) pairs of _eval_pr, not only the final maximum"""
    from anyref_amd import evalops as E
    logits, gt = _avs_case(2, 96, 128, seed=3, scale=6.0)
    conf, hist = E._avs_counts(logits.cuda(), gt.cuda(), 255)
    above = hist.flip(1).cumsum(1).flip(1)[:, 1:, :]
    th = torch.linspace(0, 1 - 1e-10, 255)
    prob = torch.sigmoid(logits)
    for m in range(2):
        for i in range(255):
            passed = prob[m] >= th[i]
            assert int(above[m, i].sum()) == int(passed.sum()), (m, i)
            assert int(above[m, i, 1]) == int((passed & (gt[m] == 1)).sum()), (m, i)
        p = torch.sigmoid(logits[m]) > 0.5
        assert conf[m].tolist() == [int((~p & (gt[m] == 0)).sum()), int((~p & (gt[m] == 1)).sum()),
                                    int((p & (gt[m] == 0)).sum()), int((p & (gt[m] == 1)).sum())]


def test_avs_edge_cases():
    from anyref_amd.evalops import mask_iou, eval_fmeasure
    z = torch.zeros(2, 8, 8)
    gt0 = torch.zeros(2, 8, 8, dtype=torch.int64)
    assert float(mask_iou(z.cuda(), gt0.cuda())) == float(O.avs_mask_iou(z, gt0))        # all-empty ground truth
    assert eval_fmeasure(z.cuda(), gt0.float().cuda()) == 0.0 == O.avs_fmeasure(z, gt0.float())
    nan = torch.full((1, 4, 4), float("nan"))
    g1 = torch.ones(1, 4, 4, dtype=torch.int64)
    assert float(mask_iou(nan.cuda(), g1.cuda())) == float(O.avs_mask_iou(nan, g1))      # NaN passes no threshold
    assert eval_fmeasure(nan.cuda(), g1.float().cuda()) == O.avs_fmeasure(nan, g1.float())
    with pytest.raises(ValueError):
        mask_iou(z.cuda(), (gt0 + 2).cuda())          # non-binary ground truth
    with pytest.raises(ValueError):
        mask_iou(z.cuda(), gt0[:, :4].cuda())
    with pytest.raises(RuntimeError):
        mask_iou(z, gt0)                              # host logits: no CPU fallback


@pytest.mark.parametrize("name", ["m_small", "m_ragged", "m_empty_gt"])
def test_metrics_against_reference_fixtures(name):
    """The HIP metric kernels against numbers the reference's own utils/utils.py / utils/pyutils.py produced
    (tests/golden/make_golden_metrics.py, run in the build container)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_metrics as gm
    from anyref_amd.evalops import intersection_and_union, mask_iou, eval_fmeasure
    fx = np.load(os.path.join(ROOT, "tests", "golden", "metrics_ref.npz"))
    logits, gt, lab = gm.metric_inputs(name)
    got = intersection_and_union(logits.cuda(), lab.cuda())
    assert np.array_equal(torch.stack([g.cpu() for g in got]).numpy(), fx[name + ".iu"])
    per = intersection_and_union(logits.cuda(), lab.cuda(), per_mask=True)
    for k in range(logits.shape[0]):
        assert np.array_equal(torch.stack([p[k].cpu() for p in per]).numpy(), fx[f"{name}.iu{k}"])
    assert float(mask_iou(logits.cuda(), gt.cuda())) == float(fx[name + ".miou"])
    assert eval_fmeasure(logits.cuda(), gt.float().cuda(), None) == float(fx[name + ".fscore"])
