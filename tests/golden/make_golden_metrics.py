"""Golden vectors for the post-path metrics, made by the reference's own
`utils/utils.py::intersectionAndUnionGPU` (:79-91) and `utils/pyutils.py::mask_iou / Eval_Fmeasure / _eval_pr`
(:163-236), imported from `/root/reference` as they are (both files need only numpy + torch).

    python tests/golden/make_golden_metrics.py        # build container only

`Eval_Fmeasure` calls `_eval_pr` with its default `cuda_flag=True`; there is no GPU here, so the fixture runs the
function's own body with `_eval_pr(..., cuda_flag=False)` bound in (same arithmetic on CPU tensors).
Inputs are regenerated from seeds by `metric_inputs`; only outputs are stored.
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
CASES = {"m_small": (3, 37, 53, 0), "m_ragged": (2, 101, 67, 1), "m_empty_gt": (3, 40, 40, 2)}


def metric_inputs(name):
    n, h, w, seed = CASES[name]
    g = torch.Generator().manual_seed(100 + seed)
    logits = torch.randn(n, h, w, generator=g) * 2.5
    gt = (torch.rand(n, h, w, generator=g) > 0.55).int()
    if name == "m_empty_gt":
        gt[1] = 0                                         # "totally black GT" branch of both AVS metrics
    lab = gt.clone()
    lab[torch.rand(n, h, w, generator=g) > 0.9] = 255     # ignore label of intersectionAndUnionGPU
    return logits, gt, lab


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    U = _load(os.path.join(REF, "utils", "utils.py"), "ref_utils")
    P = _load(os.path.join(REF, "utils", "pyutils.py"), "ref_pyutils")
    cpu_pr = P._eval_pr
    P._eval_pr = lambda a, b, n, cuda_flag=True: cpu_pr(a, b, n, cuda_flag=False)
    out = {}
    for name in CASES:
        logits, gt, lab = metric_inputs(name)
        pred = (torch.sigmoid(logits) > 0.5).int()        # eval_referseg.py:189-208
        i, u, t = U.intersectionAndUnionGPU(pred.clone().float(), lab.clone().float(), 2, ignore_index=255)
        out[name + ".iu"] = torch.stack([i, u, t]).numpy()
        for k in range(logits.shape[0]):                  # per mask, as the eval loop calls it
            i, u, t = U.intersectionAndUnionGPU(pred[k].clone().float(), lab[k].clone().float(), 2, ignore_index=255)
            out[f"{name}.iu{k}"] = torch.stack([i, u, t]).numpy()
        out[name + ".miou"] = np.float64(float(P.mask_iou(logits, gt)))
        with tempfile.TemporaryDirectory() as d:
            out[name + ".fscore"] = np.float64(P.Eval_Fmeasure(logits, gt.float(), d))
        pr, rc = cpu_pr(torch.sigmoid(logits[0]), gt[0].float(), 255, cuda_flag=False)
        out[name + ".prec0"], out[name + ".recall0"] = pr.numpy(), rc.numpy()
        print(name, out[name + ".iu"].tolist(), out[name + ".miou"], out[name + ".fscore"])
    np.savez_compressed(os.path.join(HERE, "metrics_ref.npz"), **out)


if __name__ == "__main__":
    main()
