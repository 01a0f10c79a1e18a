"""BASELINE.json configs[1] at FULL size inside `-m gpu`: LLaVA-7B-shaped LLaMA (32 x 4096 / 11008) + CLIP ViT-L/14
(23 layers run) + SAM-H (32 x 1280, 1024^2) + mask decoder, the bench inputs (S = 320 prompt, 10 new tokens), both
arithmetic modes through the C-ABI against the CPU fp32 oracle on the same (image, instruction) pairs.

Two seeded workloads:
  * "normal"  SURVEY.md §8d's throughput workload, every matrix N(0, 0.02^2) -- what bench.py times.  Its residual
              stream is tiny and its logits nearly flat (mask logits +-0.02, LM top-2 gap ~0.04 at sigma 1.3), so its
              1e-3 absolute bound cannot bite for bf16: kept for continuity with bench.py, judged RELATIVE to range.
  * "fan_in"  the parity workload (anyref_amd.synth, init="fan_in"): O(1) activations, peaked LM logits (sigma ~4),
              mask logits of several units -- where a numerical error shows.
north_star bar -- identical greedy ids and mask logits within 1e-3 -- is asserted for parity mode AND for parity16 (f32
activations as bf16 pairs against exactly stored bf16 weights: the tolerance-meeting mode bench.py times) on both.  For the
bf16 perf mode the test asserts measured-error x 2 bounds (relative to the logit range) and reports the ids-match
rate over 8 prompts; a flipped greedy id is followed by a teacher-forced comparison, never skipped.
"""
import gc
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from anyref_amd.config import config_7b, IMAGE_TOKEN_INDEX  # noqa: E402
from anyref_amd.synth import synth_state_dict  # noqa: E402
from oracle import anyref_oracle as O  # noqa: E402
from oracle.check import compare_generate, summarize  # noqa: E402

T_NEW = 10
N_PROMPTS = 8          # ids-match rate is taken over these; the first N_MASKS also compare mask logits
N_MASKS = 2
# perf-mode bounds = 2 x the error measured on MI355X (gpurun_out/r2_t3.log), relative to the range
# (round 3, SAM encoder in f16: fan_in measured 5.3e-4 .. 6.8e-4; the all-bf16 build of round 2 measured 2.4e-3)
PERF_REL_BOUND = {"normal": 0.026, "fan_in": 0.0014}     # measured 1.3e-2 (teacher-forced) / 6.8e-4


def _inputs(cfg, n, seed):
    g = torch.Generator().manual_seed(seed)
    clip = torch.randn(1, 3, 224, 224, generator=g)
    sam = torch.randn(1, 3, 1024, 1024, generator=g)
    hi = min(32000, cfg.llm.vocab - 8)
    ids = [torch.cat([torch.tensor([1, IMAGE_TOKEN_INDEX]), torch.randint(3, hi, (63,), generator=g)]) for _ in range(n)]
    return clip, sam, ids


def _stage_attribution(cfg, sd, sd_cpu, clip, sam, ref0, img_emb, img_feats, sizes, H, W):
    """Where does the bf16 mode's mask-logit error come from?  One stage at a time in bf16 (the perf handle), everything
    else fp32 (the parity handle / the oracle's own intermediates), through the stage entry points of the C-ABI, on the
    first prompt of the parity workload.  -> {stage: max-abs mask-logit error vs the oracle's masks}."""
    from anyref_amd.model import AnyRefForCausalLM
    full, hid_ref, mask_ref = ref0["output_ids"][0], ref0["hidden"][0], ref0["pred_masks"][0]
    rows = (torch.where(O._is_seg(cfg, full[1:]))[0] + cfg.clip.n_patches - 1).tolist()
    rows = [r for r in rows if r < hid_ref.shape[0]]
    par = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="parity", max_batch=1, max_seg=4)
    perf = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="perf", max_batch=1, max_seg=4)
    perf.config.eos_token_id = None

    def masks_err(emb_img, hidden_rows):
        with torch.no_grad():
            pred = O.text_hidden_fc(sd_cpu, hidden_rows.float().cpu())
        m = par.mask_decode(emb_img[0], pred, sizes[0], (H[0], W[0]))["masks"]
        return float((m.cpu() - mask_ref).abs().max())

    def embeds(feats):
        with torch.no_grad():
            return O.splice_embeddings(sd_cpu, cfg, full[:-1], feats)[None]

    out = {"range": float(mask_ref.abs().max())}
    out["none (f32 stages on the oracle's intermediates)"] = masks_err(img_emb, hid_ref[rows])
    out["SAM image encoder"] = masks_err(perf.sam_encode(sam).cpu(), hid_ref[rows])
    feats_p = perf.encode_images(clip).cpu()[0]
    out["CLIP tower + projector"] = masks_err(img_emb, par.llm_forward(embeds(feats_p))["hidden"][0][rows])
    out["LLaMA (prefill kernels, teacher-forced)"] = masks_err(img_emb, perf.llm_forward(embeds(img_feats[0]))["hidden"][0][rows])
    ids0 = full[: len(full) - T_NEW]
    (oids, _, _), ex = perf.generate(clip, ids0[None], sam, sizes, H, W, max_new_tokens=T_NEW, _return_extras=True)
    if oids[0].cpu().tolist() == full.tolist():
        out["CLIP + LLaMA (generate: prefill + decode kernels)"] = masks_err(img_emb, ex["hidden"][0][rows])
    del par, perf
    gc.collect()
    torch.cuda.empty_cache()
    return out


# C3's per-GPU shape at FULL depth (BASELINE.json configs[2]: 4 images per GPU): hidden-state bounds of the bf16 mode,
# relative to the hidden scale, 2 x the error measured on MI355X at 32 layers (prefill rows / decode rows)
C3_PERF_HIDDEN_REL = 0.01      # measured 4.2e-3 (prefill rows) / 2.9e-3 (decode rows)


def _c3_shape_check(cfg, sd, clip, sam, ids, refs, sizes, H, W):
    """BASELINE configs[2]'s per-GPU call -- FOUR (image, instruction) pairs in one `generate` -- at full depth (32 LLaMA
    layers, 23 CLIP layers, SAM-H): the batched kernels the B = 1 path never selects (prefill GEMMs at M = 1280 on 256^2
    tiles, the decode GEMV with 4 batch rows per pass over the weights (packed-dot form) / two passes of 2 f32 rows
    (parity16), 4 SAM-H encodes per call) against the CPU oracle.  The four pairs share one image, so the oracle's one
    SAM-H / CLIP forward and its four greedy decodes (already run for the B = 1 comparison) serve all rows: ids of 4 rows,
    hidden states of 4 rows, mask logits of the rows the oracle decoded masks for."""
    from anyref_amd.model import AnyRefForCausalLM
    B = 4
    clip4, sam4 = clip.expand(B, -1, -1, -1).contiguous(), sam.expand(B, -1, -1, -1).contiguous()
    ids4 = torch.stack(ids[:B])
    out = {}
    for mode in ("parity16", "perf"):
        m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=B, max_seg=4)
        m.config.eos_token_id = None
        (oids, masks, _), ex = m.generate(clip4, ids4, sam4, sizes * B, H * B, W * B, max_new_tokens=T_NEW, _return_extras=True)
        torch.cuda.synchronize()
        rep = dict(ids_identical=0, prefill_hidden_rel=0.0, decode_hidden_rel=0.0, masks_compared=0, mask_logit_max_abs_err=0.0,
                   teacher_forced=0)
        flipped = []
        for b in range(B):
            want = refs[b]["output_ids"][0]
            same = oids[b, : len(want)].cpu().tolist() == want.tolist()
            rep["ids_identical"] += int(same)
            hid = refs[b]["hidden"][0]
            Sp = len(ids[b]) + cfg.clip.n_patches - 1
            got = ex["hidden"][b, : hid.shape[0]].cpu()
            scale = hid.abs().max().item()
            rep["prefill_hidden_rel"] = max(rep["prefill_hidden_rel"], (got[:Sp] - hid[:Sp]).abs().max().item() / scale)
            if same:
                rep["decode_hidden_rel"] = max(rep["decode_hidden_rel"], (got[Sp:] - hid[Sp:]).abs().max().item() / scale)
            else:
                flipped.append(b)
            if refs[b]["pred_masks"] is not None and same:
                ref_mask = refs[b]["pred_masks"][0]
                assert masks[b].shape == ref_mask.shape, (masks[b].shape, ref_mask.shape)
                rep["masks_compared"] += 1
                rep["mask_logit_max_abs_err"] = max(rep["mask_logit_max_abs_err"], (masks[b].cpu() - ref_mask).abs().max().item())
                rep["logit_range"] = float(ref_mask.abs().max())
        if any(refs[b]["pred_masks"] is not None for b in flipped):
            # a flipped greedy id (bf16): the oracle's ids of ALL four rows teacher-forced through the batched forward
            full4 = torch.stack([refs[b]["output_ids"][0] for b in range(B)])
            fw = m.model_forward_new(clip4, sam4, full4, full4.clone(), None, sizes * B, None, H * B, W * B, _return_extras=True)
            torch.cuda.synchronize()
            for b in flipped:
                if refs[b]["pred_masks"] is None:
                    continue
                ref_mask = refs[b]["pred_masks"][0]
                rep["masks_compared"] += 1
                rep["teacher_forced"] += 1
                rep["mask_logit_max_abs_err"] = max(rep["mask_logit_max_abs_err"], (fw["pred_masks"][b].cpu() - ref_mask).abs().max().item())
                rep["logit_range"] = float(ref_mask.abs().max())
        out[mode] = rep
        del m
        gc.collect()
        torch.cuda.empty_cache()
    print("C3_SHAPE_FULL_DEPTH " + json.dumps(out), flush=True)
    p16, q = out["parity16"], out["perf"]
    n_masks = sum(1 for b in range(B) if refs[b]["pred_masks"] is not None)
    # the tolerance-meeting mode: north_star's bar on every row
    assert p16["ids_identical"] == B, p16
    assert p16["masks_compared"] == n_masks and p16["mask_logit_max_abs_err"] <= 1e-3, p16
    assert p16["prefill_hidden_rel"] < 2e-4 and p16["decode_hidden_rel"] < 2e-4, p16
    # bf16 mode: every mask row compared (teacher-forced after a flip), bounded relative to its range
    assert q["masks_compared"] == n_masks and q["mask_logit_max_abs_err"] <= PERF_REL_BOUND["fan_in"] * q["logit_range"], q
    assert q["prefill_hidden_rel"] < C3_PERF_HIDDEN_REL and q["decode_hidden_rel"] < C3_PERF_HIDDEN_REL, q


@pytest.mark.parametrize("init", ["fan_in", "normal"])
def test_c2_full_size_parity_and_perf(init):
    from anyref_amd.model import AnyRefForCausalLM
    cores = int(os.environ.get("ANYREF_CPU_THREADS", min(16, os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    cfg = config_7b()
    cfg.llm.max_seq = 512
    t0 = time.time()
    sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16, init=init)
    clip, sam, ids = _inputs(cfg, N_PROMPTS, seed=1)
    sizes, H, W = [(1024, 1024)], [1024], [1024]
    sd_cpu = {k: v.float().cpu() for k, v in sd.items()}
    print(f"[{init}] weights ready in {time.time() - t0:.0f}s", flush=True)

    # the oracle: the [SEG] id is the id its own greedy decode emits at step 3 of prompt 0
    with torch.no_grad():
        img_feats = O.encode_images(sd_cpu, cfg, clip)
        first = O.greedy_generate(sd_cpu, cfg, O.splice_embeddings(sd_cpu, cfg, ids[0], img_feats[0]), T_NEW, None)[0]
    cfg.seg_token_idx = int(first[2])
    refs = []
    t0 = time.time()
    with torch.no_grad():
        img_emb = O.sam_image_encoder(sd_cpu, cfg, sam)        # prompt-independent: one SAM-H forward for all prompts
        for i in range(N_PROMPTS):
            emb = O.splice_embeddings(sd_cpu, cfg, ids[i], img_feats[0])
            new_ids, hidden, _ = O.greedy_generate(sd_cpu, cfg, emb, T_NEW, None)
            full = torch.cat([ids[i], torch.tensor(new_ids)])
            r = dict(output_ids=[full], hidden=[hidden], pred_masks=None)
            if i < N_MASKS:
                r["pred_masks"] = O.generate_tail(sd_cpu, cfg, [full], [len(ids[i])], [hidden], None, sam, sizes, H, W,
                                                  image_embeddings=img_emb)["pred_masks"]
            refs.append(r)
    print(f"[{init}] CPU oracle: {N_PROMPTS} prompts ({N_MASKS} with masks) in {time.time() - t0:.0f}s on {cores} threads",
          flush=True)
    assert refs[0]["pred_masks"] is not None

    report = {}
    for mode in ("parity", "parity16", "perf"):
        m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=1, max_seg=4)
        m.config.eos_token_id = None
        if mode == "parity16":      # weights never widened: the handle stays at the bf16 mode's footprint
            assert m.device_bytes <= 17 * 2 ** 30, m.device_bytes
        rows = [compare_generate(m, refs[i], clip, ids[i], sam, sizes, H, W, T_NEW, sd_cpu["lm_head.weight"],
                                 cfg.clip.n_patches) for i in range(N_PROMPTS)]
        report[mode] = summarize(rows)
        del m
        gc.collect()
        torch.cuda.empty_cache()
    print(f"C2_FULL[{init}] " + json.dumps(report), flush=True)
    if init == "fan_in":
        att = _stage_attribution(cfg, sd, sd_cpu, clip, sam, refs[0], img_emb, img_feats, sizes, H, W)
        print("C2_ERROR_BUDGET " + json.dumps(att), flush=True)
        # each stage alone stays inside the end-to-end bound, and the f32 path on the oracle's intermediates inside 1e-3
        assert att["none (f32 stages on the oracle's intermediates)"] <= 1e-3, att
        assert all(v <= PERF_REL_BOUND[init] * att["range"] for k, v in att.items() if k != "range"), att
        # configs[2]'s per-GPU shape (4 pairs per call) at full depth, on the oracle results above
        _c3_shape_check(cfg, sd, clip, sam, ids, refs, sizes, H, W)
    p, q = report["parity"], report["perf"]
    # north_star, parity mode: identical greedy ids on every prompt, mask logits within 1e-3
    assert p["ids_match_rate"] == 1.0, p
    assert p["masks_compared"] >= 1 and p["mask_logit_max_abs_err"] <= 1e-3, p
    # ... and the same ABSOLUTE bar for the mode that meets it at 16-bit MFMA rate (f32 activations as bf16 pairs)
    p16 = report["parity16"]
    assert p16["ids_match_rate"] == 1.0, p16
    assert p16["masks_compared"] == p["masks_compared"] and p16["mask_logit_max_abs_err"] <= 1e-3, p16
    # perf (bf16) mode: every prompt compared (teacher-forced after a flipped id), error bounded relative to range
    assert q["masks_compared"] == p["masks_compared"], q
    assert q["mask_logit_rel_err"] <= PERF_REL_BOUND[init], q
