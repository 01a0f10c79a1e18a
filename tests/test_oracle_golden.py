"""Pin the CPU oracle to the golden vectors.

SAM half: vectors produced by the reference's own `segment_anything/modeling` code.
LLaMA/CLIP half: vectors from the HF transformers stand-in (the reference's `model/llava`
layer is absent, so that half is pinned to the stand-in only).
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_golden as mg  # noqa: E402  (config/input helpers only; never touches /root/reference on import)
from anyref_amd.synth import synth_state_dict, SAM_PREFIX, CLIP_PREFIX  # noqa: E402
from oracle import anyref_oracle as O  # noqa: E402

TOL = 2e-5


def _close(a, b, tol=TOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b).max()
    scale = max(1.0, np.abs(b).max())
    assert err <= tol * scale, f"max abs err {err} (scale {scale})"


@pytest.mark.parametrize("name", list(mg.golden_cfgs().keys()))
def test_sam_half_against_reference(name):
    cfg = mg.golden_cfgs()[name]
    fx = np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    seed = int(fx["seed"])
    w = synth_state_dict(cfg, seed=seed, scale=0.05)
    assert abs(mg.checksum(w, SAM_PREFIX) - fx["wsum"]) < 1e-6 * fx["wsum"], "seeded weights drifted"
    img, text = mg.golden_inputs(cfg, seed)
    assert abs(float(img.double().abs().sum() + text.double().abs().sum()) - fx["insum"]) < 1e-6 * fx["insum"]
    with torch.no_grad():
        emb = O.sam_image_encoder(w, cfg, img)
        _close(emb[:, ::2], fx["emb"])
        dpe = O.dense_pe(w, cfg)
        _close(dpe[:, ::4], fx["dense_pe"])
        sparse, dense = O.prompt_encoder_text(w, cfg, text)
        masks4, iou4 = O.mask_decoder_predict(w, cfg, emb[0:1], dpe, sparse, dense)
        _close(masks4[:, :, ::2, ::2], fx["masks4"])
        _close(iou4, fx["iou4"])
        tokens = torch.cat([torch.cat([w[SAM_PREFIX + "mask_decoder.iou_token.weight"],
                                       w[SAM_PREFIX + "mask_decoder.mask_tokens.weight"]], 0)[None].expand(3, -1, -1),
                            sparse], 1)
        hs, keys = O.two_way_transformer(w, cfg, emb[0:1].expand(3, -1, -1, -1) + dense,
                                         dpe.expand(3, -1, -1, -1), tokens)
        _close(hs, fx["hs"])
        _close(keys[:, ::16], fx["keys"])
        low = masks4[:, 0:1]
        _close(low[:, :, ::2, ::2], fx["low"])
        S = cfg.sam.img_size
        _close(O.postprocess_masks(cfg, low, (S, S), (S, S))[:, :, ::8, ::8], fx["post_a"])
        _close(O.postprocess_masks(cfg, low, (150, 224), (301, 437))[:, :, ::8, ::8], fx["post_b"])


def test_llm_clip_half_against_hf_standin():
    cfg = mg.llm_clip_cfg()
    fx = np.load(os.path.join(HERE, "golden", "llm_clip_hf.npz"))
    seed = int(fx["seed"])
    w = synth_state_dict(cfg, seed=seed, scale=0.08)
    assert abs(mg.checksum(w, "model.layers") + mg.checksum(w, CLIP_PREFIX) - fx["wsum"]) < 1e-6 * fx["wsum"]
    images, embeds = mg.llm_clip_inputs(cfg, seed)
    with torch.no_grad():
        _close(O.clip_patch_tokens(w, cfg, images)[:, ::2], fx["clip_feat"])
        hidden, attn = O.llama_layers(w, cfg, embeds[0], want_last_attn=True)
        _close(hidden[None], fx["hidden"])
        _close(torch.nn.functional.linear(hidden[-1], w["lm_head.weight"]), fx["logits_last"], 5e-5)
        _close(attn.mean(0), fx["attn_last_mean"])
        for use_cache in (True, False):
            ids, hid2, att2 = O.greedy_generate(w, cfg, embeds[0], 12, None, use_cache=use_cache, want_attn=True)
            assert ids == fx["gen_ids"][0].tolist()
            # cached and uncached loops must expose the same hidden states / attention rows
            assert hid2.shape[0] == 40 + 11
            _close(hid2[:40][None], fx["hidden"], 5e-5)
            _close(att2[:40, :40], fx["attn_last_mean"], 5e-5)


def test_generate_cached_equals_uncached_tiny():
    from anyref_amd.config import config_tiny
    cfg = config_tiny()
    w = synth_state_dict(cfg, seed=3, scale=0.05)
    g = torch.Generator().manual_seed(4)
    clip = torch.randn(1, 3, 224, 224, generator=g)
    sam = torch.randn(1, 3, 224, 224, generator=g)
    ids = torch.cat([torch.tensor([1, O.IMAGE_TOKEN_INDEX]), torch.randint(3, 990, (14,), generator=g)])
    with torch.no_grad():
        r0 = O.anyref_generate(w, cfg, clip, [ids], sam, [(224, 224)], [224], [224], max_new_tokens=4, eos=False)
        # rig the [SEG] id as SURVEY.md §8c-3: the id emitted at step 3
        cfg.seg_token_idx = int(r0["output_ids"][0][-2])
        a = O.anyref_generate(w, cfg, clip, [ids], sam, [(224, 224)], [224], [224], max_new_tokens=4, eos=False)
        b = O.anyref_generate(w, cfg, clip, [ids], sam, [(224, 224)], [224], [224], max_new_tokens=4, eos=False,
                              use_cache=False)
    assert a["pred_masks"] is not None and a["pred_masks"][0].shape[-2:] == (224, 224)
    assert torch.equal(a["output_ids"][0], b["output_ids"][0])
    _close(a["pred_masks"][0], b["pred_masks"][0].numpy(), 1e-4)


# ------------------------------------------------------------------------------------------------
# Glue: the reference's own `model/anyref.py` run on canned LLM outputs (tests/golden/make_golden_glue.py)
# ------------------------------------------------------------------------------------------------
import make_golden_glue as gg  # noqa: E402  (case tables + seeded inputs; touches /root/reference only in main())

GLUE = np.load(os.path.join(HERE, "golden", "glue_anyref.npz"))


def _glue_setup(c, seg_list=False):
    cfg = gg.glue_cfg()
    cfg.rephrase_weight = c["rephrase"]
    cfg.seg_token_idx = gg.SEG_LIST if seg_list else gg.SEG
    return cfg, synth_state_dict(cfg, seed=gg.SEED, scale=0.05)


@pytest.mark.parametrize("name", list(gg.GEN_CASES))
def test_generate_tail_against_reference(name):
    """anyref.py:718-822 on canned `sequences` / `hidden_states[-1]` / `attentions[-1]`."""
    x = gg.case_inputs(name)
    c = x["c"]
    cfg, w = _glue_setup(c, c.get("seg_list", False))
    bs = c["bs"]
    with torch.no_grad():
        r = O.generate_tail(w, cfg, [x["seq"][b] for b in range(bs)], [c["L"]] * bs, [x["hidden"][b] for b in range(bs)],
                            [x["attn"][b] for b in range(bs)], x["sam"], c["sizes"], [h for h, _ in c["hw"]],
                            [w_ for _, w_ in c["hw"]])
    if int(GLUE[name + ".masks_none"]):
        assert r["pred_masks"] is None and int(GLUE[name + ".arity"]) == 3     # (ids, None, (None,)*3), :730
        return
    assert int(GLUE[name + ".arity"]) == 2                                     # the success path returns 2, :822
    for b in range(bs):
        assert list(r["pred_masks"][b].shape) == GLUE[f"{name}.shape{b}"].tolist()
        if r["pred_masks"][b].shape[0]:
            _close(r["pred_masks"][b][:, ::3, ::3], GLUE[f"{name}.mask{b}"], 3e-5)


@pytest.mark.parametrize("name", list(gg.FWD_CASES))
def test_forward_tail_against_reference(name):
    """anyref.py:273-282,356-466 (hand-off at pos-1+255, rephrase, mask decode, BCE + Dice) on canned LLM outputs."""
    x = gg.case_inputs(name)
    c = x["c"]
    cfg, w = _glue_setup(c)
    bs = c["bs"]
    with torch.no_grad():
        r = O.forward_tail(w, cfg, [x["seq"][b] for b in range(bs)], [x["labels"][b] for b in range(bs)],
                           [x["hidden"][b] for b in range(bs)], [x["attn"][b] for b in range(bs)], x["lm_loss"], x["sam"],
                           c["sizes"], x["gt"], [h for h, _ in c["hw"]], [w_ for _, w_ in c["hw"]])
    keys = GLUE[name + ".keys"].tolist()
    assert sorted(k for k in r if k in ("loss", "lm_loss", "ce_loss", "dice_loss", "mask_loss")) == keys
    for k in keys:
        assert abs(float(r[k]) - float(GLUE[f"{name}.{k}"])) < 2e-5 * max(1.0, abs(float(GLUE[f"{name}.{k}"]))), k


def test_handdown_and_losses_against_reference():
    """What the glue hands to the llava layer for audio / reference images (anyref.py:303-339, :663-702) and the
    two mask losses (:19-68)."""
    cfg, w = _glue_setup(dict(rephrase=0.0))
    hd = gg.handdown_inputs()
    proj = lambda e: torch.nn.functional.linear(e, w["model.audio_projector.weight"], w["model.audio_projector.bias"])
    with torch.no_grad():
        _close(proj(hd["audio_emb"])[0], GLUE["hand.gen_list.audio0"], 2e-5)
        _close(proj(hd["audio_emb"]), GLUE["hand.gen_tensor.audio"], 2e-5)
        _close(proj(hd["audio_emb2"])[0], GLUE["hand.fwd_list.audio0"], 2e-5)
        lst = O.ref_features_generate(gg.encode_stub, [hd["ref_a"]], 1)
        assert lst[0].shape == (256, gg.H_LLM)                  # list items go down UNPOOLED in generate (:691-692)
        _close(lst[0], GLUE["hand.gen_list.ref0"], 2e-5)
        roi = O.ref_features_generate(gg.encode_stub, [hd["roi"]], 1)
        assert np.array_equal(roi[0].numpy(), GLUE["hand.gen_list.roi0"])
        ten = O.ref_features_generate(gg.encode_stub, hd["ref_a"][None], 1)
        assert ten.shape == (1, gg.IMG_REF_NUM, gg.H_LLM)       # tensor batches are pooled 256 -> 16 -> 4 (:695-700)
        _close(ten, GLUE["hand.gen_tensor.ref"], 2e-5)
        fwd = O.ref_features_forward(gg.encode_stub, [hd["ref_b"]])
        _close(fwd[0], GLUE["hand.fwd_list.ref0"], 2e-5)
        # the llava-layer reading this build takes for unpooled rows equals the forward path's pooling
        _close(O.splice_ref_rows(lst[0], gg.IMG_REF_NUM), GLUE["hand.gen_tensor.ref"][0], 2e-5)
    g = torch.Generator().manual_seed(gg.SEED + 9)
    lg = torch.randn(3, 40, 50, generator=g) * 3
    tg = (torch.rand(3, 40, 50, generator=g) > 0.6).float()
    assert abs(float(O.dice_loss(lg, tg, 3)) - float(GLUE["loss.dice"])) < 1e-6
    assert abs(float(O.sigmoid_ce_loss(lg, tg, 3)) - float(GLUE["loss.bce"])) < 1e-6
    from anyref_amd.model import dice_loss, sigmoid_ce_loss     # the mirror's copies of the two formulas
    assert abs(float(dice_loss(lg, tg, 3)) - float(GLUE["loss.dice"])) < 1e-6
    assert abs(float(sigmoid_ce_loss(lg, tg, 3)) - float(GLUE["loss.bce"])) < 1e-6


# ------------------------------------------------------------------------------------------------
# Metrics: the reference's utils/utils.py + utils/pyutils.py run here (tests/golden/make_golden_metrics.py)
# ------------------------------------------------------------------------------------------------
import make_golden_metrics as gm  # noqa: E402

METRICS = np.load(os.path.join(HERE, "golden", "metrics_ref.npz"))


@pytest.mark.parametrize("name", list(gm.CASES))
def test_metric_restatements_against_reference(name):
    logits, gt, lab = gm.metric_inputs(name)
    i, u, t = O.eval_mask_counts(logits, lab)                                      # utils/utils.py:79-91
    assert np.array_equal(torch.stack([i, u, t]).numpy(), METRICS[name + ".iu"])
    for k in range(logits.shape[0]):
        i, u, t = O.eval_mask_counts(logits[k], lab[k])
        assert np.array_equal(torch.stack([i, u, t]).numpy(), METRICS[f"{name}.iu{k}"])
    assert float(O.avs_mask_iou(logits, gt)) == float(METRICS[name + ".miou"])      # pyutils.py:163-190
    assert O.avs_fmeasure(logits, gt.float()) == float(METRICS[name + ".fscore"])   # pyutils.py:193-220
    pr, rc = O.avs_pr_curve(torch.sigmoid(logits[0]), gt[0].float(), 255)           # pyutils.py:223-236
    assert np.array_equal(pr.numpy(), METRICS[name + ".prec0"]) and np.array_equal(rc.numpy(), METRICS[name + ".recall0"])


def test_sam_h_width_encoder_against_reference():
    """The oracle's image encoder at SAM-H's real shapes (1024^2, width 1280, heads of 80, 14-windows padded 64 -> 70)
    against the reference's `ImageEncoderViT` run at those shapes (4 blocks)."""
    cfg = mg.sam_h_width_cfg()
    fx = np.load(os.path.join(HERE, "golden", "sam_h_width.npz"))
    seed = int(fx["seed"])
    w = synth_state_dict(cfg, seed=seed, scale=0.02)
    assert abs(mg.checksum(w, SAM_PREFIX + "image_encoder.") - fx["wsum"]) < 1e-6 * fx["wsum"]
    img = mg.sam_h_width_inputs(seed)
    assert abs(float(img.double().abs().sum()) - fx["insum"]) < 1e-6 * fx["insum"]
    with torch.no_grad():
        emb = O.sam_image_encoder(w, cfg, img)
    _close(emb[:, ::4, ::2, ::2], fx["emb"], 3e-5)


def test_rel_pos_interpolation_against_reference():
    """`get_rel_pos` with tables of another length (image_encoder.py:333-345): the oracle's resample against the
    tables the reference's own function gathered from, and the encoder output of the reference run with them."""
    cfg = mg.golden_cfgs()["sam_w14"]
    fx = np.load(os.path.join(HERE, "golden", "sam_relpos_interp.npz"))
    seed = int(fx["seed"])
    w = synth_state_dict(cfg, seed=seed, scale=0.05)
    assert abs(mg.checksum(w, SAM_PREFIX + "image_encoder.") - fx["wsum"]) < 1e-6 * fx["wsum"], "seeded weights drifted"
    tabs = mg.relpos_interp_tables(cfg, seed)
    for k, v in tabs.items():
        short = k.split("blocks.")[1]
        np.testing.assert_array_equal(v.numpy(), fx["tab_" + short])
        _close(O.resample_rel_pos(v, 27), fx["res_" + short], 1e-6)
    w.update(tabs)
    img, _ = mg.golden_inputs(cfg, seed)
    with torch.no_grad():
        _close(O.sam_image_encoder(w, cfg, img)[:, ::2], fx["emb"])
