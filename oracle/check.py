"""Parity bookkeeping shared by `tests/`, `__graft_entry__.smoke()` and `bench.py`'s parity leg.

TEST INFRASTRUCTURE ONLY (like the rest of `oracle/`): it holds the HIP backend's `generate()` against a CPU
oracle result for the same (image, instruction) pair and reports what north_star asks for -- identical greedy ids,
mask-logit max-abs error -- plus what makes those numbers interpretable: the error relative to the logits' range,
the position of the first diverging token, and how tied the oracle's argmax was there.

When a reduced-precision mode flips a greedy id, everything after that token is a different (equally valid)
continuation and cannot be compared.  The NUMERICAL error of the path is then isolated by teacher forcing: the
oracle's own ids go through the backend's `model_forward_new` (anyref.py:239-430 semantics: hidden state that
predicted each [SEG], same hand-off, same SAM path) and the masks are compared -- never skipped.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


def compare_generate(model, ref: Dict, clip, ids_row: torch.Tensor, sam, sizes, H, W, max_new_tokens: int,
                     lm_head: Optional[torch.Tensor] = None, n_img: int = 256, **gen_kw) -> Dict:
    """One image.  `ref` = `anyref_oracle.anyref_generate(...)` of the same inputs; `model` = the HIP backend.
    `gen_kw` (e.g. `audios=[mel]`) goes to `generate` and to the teacher-forced `model_forward_new` alike."""
    out_ids, masks, _ = model.generate(clip, ids_row[None], sam, sizes, H, W, max_new_tokens=max_new_tokens, **gen_kw)
    torch.cuda.synchronize()
    got = out_ids[0].cpu().tolist()
    want = ref["output_ids"][0].tolist()
    res: Dict = {"greedy_ids_identical": got == want}
    ref_mask = ref["pred_masks"][0] if ref.get("pred_masks") is not None else None
    if ref_mask is not None:
        res["logit_range"] = float(ref_mask.abs().max())
    L0 = len(ids_row)
    if got != want:
        k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
        res["first_divergence_new_token"] = k - L0
        if lm_head is not None and 0 < k < len(want):
            with torch.no_grad():
                lg = torch.nn.functional.linear(ref["hidden"][0][k - 1 + n_img - 1], lm_head)
                top2 = torch.topk(lg, 2).values
            res["oracle_top2_logit_gap_there"] = float(top2[0] - top2[1])
            res["oracle_logit_std"] = float(lg.std())
        if ref_mask is not None:
            full = ref["output_ids"][0]
            fw = model.model_forward_new(clip, sam, full[None], full[None].clone(), None, sizes, None, H, W,
                                         _return_extras=True, **gen_kw)
            torch.cuda.synchronize()
            masks = fw.get("pred_masks")
            res["teacher_forced"] = True
    if ref_mask is not None:
        if masks is None or masks[0].shape != ref_mask.shape:
            res["mask_logit_max_abs_err"] = float("inf")
        else:
            res["mask_logit_max_abs_err"] = float((masks[0].cpu() - ref_mask).abs().max())
        res["mask_logit_rel_err"] = res["mask_logit_max_abs_err"] / max(res["logit_range"], 1e-30)
    return res


def summarize(rows: List[Dict]) -> Dict:
    """ids-match rate and worst mask error over several prompts."""
    n = len(rows)
    out = {"prompts": n, "ids_match_rate": sum(r["greedy_ids_identical"] for r in rows) / max(n, 1)}
    div = [r["first_divergence_new_token"] for r in rows if "first_divergence_new_token" in r]
    if div:
        out["first_divergence_new_token"] = min(div)
        gaps = [r["oracle_top2_logit_gap_there"] for r in rows if "oracle_top2_logit_gap_there" in r]
        if gaps:
            out["oracle_top2_logit_gap_at_divergence"] = [round(g, 4) for g in gaps]
    errs = [r for r in rows if "mask_logit_max_abs_err" in r]
    if errs:
        w = max(errs, key=lambda r: r["mask_logit_rel_err"])
        out.update(mask_logit_max_abs_err=w["mask_logit_max_abs_err"], mask_logit_rel_err=w["mask_logit_rel_err"],
                   logit_range=w["logit_range"], masks_compared=len(errs),
                   teacher_forced=sum(1 for r in errs if r.get("teacher_forced")))
    return out
