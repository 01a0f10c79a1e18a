"""Generate the golden vectors that pin `oracle/anyref_oracle.py`.

Run ONLY in the build container (it needs `/root/reference`, which never travels):

    python tests/golden/make_golden.py

(1) SAM half: imports the reference's own `model/segment_anything/modeling` package by
    path (the parent package needs torchvision), builds small / real-shaped
    modules, loads seeded weights produced by `anyref_amd.synth.synth_state_dict`, runs the
    REFERENCE code and stores inputs + outputs.
(2) LLaMA / CLIP half: the reference's `model/llava` layer is absent; the stand-in is HF
    transformers (5.15 here, eager attention) `LlamaForCausalLM` / `CLIPVisionModel` with the
    same seeded weights.

Weights are not stored (they are regenerated from the seed); each fixture carries a
checksum of the weights it was made with so generator drift is detected.
Outputs are small `.npz` files next to this script.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from anyref_amd.config import AnyRefConfig, ClipConfig, LlmConfig, SamConfig  # noqa: E402
from anyref_amd.synth import synth_state_dict, SAM_PREFIX, CLIP_PREFIX  # noqa: E402

REF_SAM = "/root/reference/model/segment_anything"


def checksum(sd, prefix=""):
    tot = 0.0
    for k in sorted(sd):
        if k.startswith(prefix):
            tot += float(sd[k].double().abs().sum())
    return np.float64(tot)


def golden_cfgs():
    """name -> config.  Chosen to hit: window padding (14 % 4 != 0), head_dim 80 (SAM-H's),
    non-square crops in postprocess, a global + a windowed block."""
    base = dict(clip=ClipConfig(image_size=224, patch=14, dim=64, heads=2, layers=2, mlp=128),
                llm=LlmConfig(vocab=200, dim=64, heads=2, layers=1, mlp=96, max_seq=512))
    return {
        "sam_w14": AnyRefConfig(sam=SamConfig(img_size=224, patch=16, dim=192, depth=2, heads=3,
                                              window=14, global_idx=(1,)), **base),
        "sam_w4_hd80": AnyRefConfig(sam=SamConfig(img_size=224, patch=16, dim=160, depth=3, heads=2,
                                                  window=4, global_idx=(2,)), **base),
    }


def golden_inputs(cfg, seed):
    g = torch.Generator().manual_seed(seed + 1)
    img = torch.randn(2, 3, cfg.sam.img_size, cfg.sam.img_size, generator=g)
    text = torch.randn(3, 1, cfg.sam.out_chans, generator=g) * 0.5
    return img, text


def llm_clip_cfg():
    return AnyRefConfig(
        clip=ClipConfig(image_size=224, patch=14, dim=128, heads=2, layers=3, mlp=256),
        llm=LlmConfig(vocab=500, dim=128, heads=4, layers=2, mlp=344, max_seq=512),
        sam=SamConfig(img_size=224, patch=16, dim=64, depth=1, heads=1, window=14, global_idx=(0,)))


def llm_clip_inputs(cfg, seed):
    g = torch.Generator().manual_seed(seed + 1)
    images = torch.randn(2, 3, 224, 224, generator=g)
    embeds = torch.randn(1, 40, cfg.llm.dim, generator=g) * 0.5
    return images, embeds


def make_sam(name, cfg, seed):
    sys.path.insert(0, REF_SAM)
    import modeling as ref  # the reference's package, imported by path
    from functools import partial
    s = cfg.sam
    sd = synth_state_dict(cfg, seed=seed, scale=0.05)
    enc = ref.ImageEncoderViT(
        depth=s.depth, embed_dim=s.dim, img_size=s.img_size, mlp_ratio=s.mlp_ratio,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=s.heads, patch_size=s.patch,
        qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(s.global_idx),
        window_size=s.window, out_chans=s.out_chans)
    pe = ref.PromptEncoder(embed_dim=s.out_chans, image_embedding_size=(s.grid, s.grid),
                           input_image_size=(s.img_size, s.img_size), mask_in_chans=16)
    dec = ref.MaskDecoder(num_multimask_outputs=3,
                          transformer=ref.TwoWayTransformer(depth=s.dec_depth, embedding_dim=s.out_chans,
                                                            mlp_dim=s.dec_mlp, num_heads=s.dec_heads),
                          transformer_dim=s.out_chans, iou_head_depth=3, iou_head_hidden_dim=256)
    sam = ref.Sam(enc, pe, dec).eval()
    sub = {k[len(SAM_PREFIX):]: v for k, v in sd.items() if k.startswith(SAM_PREFIX)}
    missing, unexpected = sam.load_state_dict(sub, strict=False)
    assert not unexpected, unexpected
    # everything we did not set must be off the text-prompt path
    for m in missing:
        assert m.startswith(("prompt_encoder.point_embeddings", "prompt_encoder.not_a_point",
                             "prompt_encoder.mask_downscaling")), m

    img, text = golden_inputs(cfg, seed)
    with torch.no_grad():
        emb = sam.image_encoder(img)
        sparse, dense = sam.prompt_encoder(points=None, boxes=None, masks=None, text_embeds=text)
        dpe = sam.prompt_encoder.get_dense_pe()
        masks4, iou4 = sam.mask_decoder.predict_masks(emb[0:1], dpe, sparse, dense)
        low, iou = sam.mask_decoder(emb[0:1], dpe, sparse, dense, multimask_output=False)
        hs, keys = sam.mask_decoder.transformer(
            torch.repeat_interleave(emb[0:1], 3, 0) + dense, torch.repeat_interleave(dpe, 3, 0),
            torch.cat([torch.cat([dec.iou_token.weight, dec.mask_tokens.weight], 0)[None].expand(3, -1, -1),
                       sparse], 1))
        post_a = sam.postprocess_masks(low, (s.img_size, s.img_size), (s.img_size, s.img_size))
        post_b = sam.postprocess_masks(low, (150, 224), (301, 437))
    # inputs are regenerated from the seed by the test (`golden_inputs`); large outputs are
    # stored strided to keep the fixtures small.
    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"), seed=seed, wsum=checksum(sd, SAM_PREFIX),
        insum=np.float64(img.double().abs().sum() + text.double().abs().sum()),
        emb=emb.numpy()[:, ::2], dense_pe=dpe.numpy()[:, ::4],
        masks4=masks4.numpy()[:, :, ::2, ::2], iou4=iou4.numpy(), low=low.numpy()[:, :, ::2, ::2], hs=hs.numpy(),
        keys=keys.numpy()[:, ::16], post_a=post_a.numpy()[:, :, ::8, ::8], post_b=post_b.numpy()[:, :, ::8, ::8])
    print(name, "emb", tuple(emb.shape), "low", tuple(low.shape), "post_b", tuple(post_b.shape))


def sam_h_width_cfg():
    """SAM-H's real shapes (build_sam.py:15-22: 1024^2 image, width 1280, 16 heads of 80, window 14) cut to 4
    blocks (3 windowed + 1 global) so the reference runs in seconds."""
    c = AnyRefConfig(clip=ClipConfig(image_size=224, patch=14, dim=64, heads=2, layers=2, mlp=128),
                     llm=LlmConfig(vocab=200, dim=64, heads=2, layers=1, mlp=96, max_seq=512))
    c.sam = SamConfig(img_size=1024, patch=16, dim=1280, depth=4, heads=16, window=14, global_idx=(3,))
    return c


def sam_h_width_inputs(seed):
    return torch.randn(1, 3, 1024, 1024, generator=torch.Generator().manual_seed(seed + 1))


def make_sam_h_width(seed=11):
    sys.path.insert(0, REF_SAM)
    import modeling as ref
    from functools import partial
    cfg = sam_h_width_cfg()
    s = cfg.sam
    sd = synth_state_dict(cfg, seed=seed, scale=0.02)
    enc = ref.ImageEncoderViT(
        depth=s.depth, embed_dim=s.dim, img_size=s.img_size, mlp_ratio=s.mlp_ratio,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=s.heads, patch_size=s.patch,
        qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(s.global_idx),
        window_size=s.window, out_chans=s.out_chans).eval()
    pre = SAM_PREFIX + "image_encoder."
    enc.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    img = sam_h_width_inputs(seed)
    with torch.no_grad():
        emb = enc(img)
    np.savez_compressed(os.path.join(HERE, "sam_h_width.npz"), seed=seed, wsum=checksum(sd, pre),
                        insum=np.float64(img.double().abs().sum()), emb=emb.numpy()[:, ::4, ::2, ::2],
                        absmax=np.float64(emb.abs().max()))
    print("sam_h_width: emb", tuple(emb.shape), "absmax", float(emb.abs().max()))


def relpos_interp_tables(cfg, seed):
    """rel_pos tables of OTHER lengths than 2*size-1 for every block of `sam_w14`'s encoder (a checkpoint trained
    at another window / image size): block 0 (14-window) gets 13-row tables (up-sampled to 27), block 1 (global,
    grid 14) 39-row tables (down-sampled to 27) -- both directions of image_encoder.py:335-345."""
    g = torch.Generator().manual_seed(seed + 5)
    hd = cfg.sam.dim // cfg.sam.heads
    pre = SAM_PREFIX + "image_encoder.blocks."
    return {pre + "0.attn.rel_pos_h": torch.randn(13, hd, generator=g) * 0.3,
            pre + "0.attn.rel_pos_w": torch.randn(13, hd, generator=g) * 0.3,
            pre + "1.attn.rel_pos_h": torch.randn(39, hd, generator=g) * 0.3,
            pre + "1.attn.rel_pos_w": torch.randn(39, hd, generator=g) * 0.3}


def make_sam_relpos_interp(seed=11):
    """`get_rel_pos` with a mismatched table (image_encoder.py:333-345): the reference's encoder with its rel_pos
    parameters replaced by tables of other lengths; stores the encoder output and the reference's resampled tables."""
    sys.path.insert(0, REF_SAM)
    import modeling as ref
    from modeling import image_encoder as ref_ie
    from functools import partial
    cfg = golden_cfgs()["sam_w14"]
    s = cfg.sam
    sd = synth_state_dict(cfg, seed=seed, scale=0.05)
    enc = ref.ImageEncoderViT(
        depth=s.depth, embed_dim=s.dim, img_size=s.img_size, mlp_ratio=s.mlp_ratio,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=s.heads, patch_size=s.patch,
        qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(s.global_idx),
        window_size=s.window, out_chans=s.out_chans).eval()
    pre = SAM_PREFIX + "image_encoder."
    enc.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    tabs = relpos_interp_tables(cfg, seed)
    for k, v in tabs.items():
        blk, name = int(k.split("blocks.")[1].split(".")[0]), k.rsplit(".", 1)[1]
        setattr(enc.blocks[blk].attn, name, torch.nn.Parameter(v.clone()))
    img, _ = golden_inputs(cfg, seed)
    with torch.no_grad():
        emb = enc(img)
        # the resampled 27-row table as the reference's get_rel_pos sees it: R[q, k] = table[q - k + 13], so
        # R[0, 13..0] are rows 0..13 and R[1..13, 0] rows 14..26
        res_full = {}
        for k, v in tabs.items():
            R = ref_ie.get_rel_pos(14, 14, v)
            res_full[k] = torch.cat([R[0].flip(0), R[1:, 0]], 0)
    np.savez_compressed(os.path.join(HERE, "sam_relpos_interp.npz"), seed=seed, wsum=checksum(sd, pre),
                        insum=np.float64(img.double().abs().sum()), emb=emb.numpy()[:, ::2],
                        **{"tab_" + k.split("blocks.")[1]: v.numpy() for k, v in tabs.items()},
                        **{"res_" + k.split("blocks.")[1]: v.numpy() for k, v in res_full.items()})
    print("sam_relpos_interp: emb", tuple(emb.shape), "resampled", tuple(next(iter(res_full.values())).shape))


def make_llm_clip(seed=7):
    from transformers import LlamaConfig, LlamaForCausalLM, CLIPVisionConfig, CLIPVisionModel
    cfg = llm_clip_cfg()
    sd = synth_state_dict(cfg, seed=seed, scale=0.08)
    c, l = cfg.clip, cfg.llm
    hf_l = LlamaForCausalLM(LlamaConfig(
        vocab_size=l.vocab, hidden_size=l.dim, intermediate_size=l.mlp, num_hidden_layers=l.layers,
        num_attention_heads=l.heads, num_key_value_heads=l.heads, rms_norm_eps=l.rms_eps,
        max_position_embeddings=2048, rope_theta=l.rope_theta, attention_bias=False, tie_word_embeddings=False,
        attn_implementation="eager")).eval()
    llm_sd = {k: v for k, v in sd.items() if k.startswith(("model.layers.", "model.embed_tokens", "model.norm", "lm_head"))}
    missing, unexpected = hf_l.load_state_dict(llm_sd, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    hf_c = CLIPVisionModel(CLIPVisionConfig(
        hidden_size=c.dim, intermediate_size=c.mlp, num_hidden_layers=c.layers, num_attention_heads=c.heads,
        image_size=c.image_size, patch_size=c.patch, hidden_act="quick_gelu", layer_norm_eps=c.eps,
        attn_implementation="eager")).eval()
    hf_keys = list(hf_c.state_dict().keys())
    pre = "vision_model." if hf_keys[0].startswith("vision_model.") else ""   # 4.x has the prefix, 5.x not
    clip_sd = {pre + k[len(CLIP_PREFIX):]: v for k, v in sd.items() if k.startswith(CLIP_PREFIX)}
    missing, unexpected = hf_c.load_state_dict(clip_sd, strict=False)
    assert not unexpected, unexpected
    # layers past select_layer and post_layernorm are off the path (hidden_states[-2])
    for m in missing:
        assert ("post_layernorm" in m or f"layers.{c.layers - 1}." in m or "position_ids" in m), m

    images, embeds = llm_clip_inputs(cfg, seed)
    with torch.no_grad():
        co = hf_c(pixel_values=images, output_hidden_states=True)
        clip_feat = co.hidden_states[c.select_layer][:, 1:]
        lo = hf_l(inputs_embeds=embeds, output_hidden_states=True, output_attentions=True, use_cache=False)
        hidden = lo.hidden_states[-1]
        attn_last = lo.attentions[-1]
        gen = hf_l.generate(inputs_embeds=embeds, do_sample=False, max_new_tokens=12, use_cache=True,
                            eos_token_id=None, pad_token_id=0)
    np.savez_compressed(
        os.path.join(HERE, "llm_clip_hf.npz"), seed=seed, wsum=checksum(sd, "model.layers") + checksum(sd, CLIP_PREFIX),
        insum=np.float64(images.double().abs().sum() + embeds.double().abs().sum()),
        clip_feat=clip_feat.numpy()[:, ::2], hidden=hidden.numpy(),
        logits_last=lo.logits[0, -1].numpy(), attn_last_mean=attn_last[0].mean(0).numpy(), gen_ids=gen.numpy())
    print("llm_clip_hf: clip_feat", tuple(clip_feat.shape), "hidden", tuple(hidden.shape), "gen", gen.tolist())


if __name__ == "__main__":
    torch.manual_seed(0)
    for i, (name, cfg) in enumerate(golden_cfgs().items()):
        make_sam(name, cfg, seed=11 + i)
    make_llm_clip()
    make_sam_h_width()
    make_sam_relpos_interp()
