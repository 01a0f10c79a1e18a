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
