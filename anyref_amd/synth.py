"""Random-init state dicts under the reference's weight names.

There is no network for checkpoints, so benchmarks and tests run on synthetic weights
(SURVEY.md §8d): Linear/Conv ~ N(0, scale²), norm weights 1 (+ jitter), SAM's
`positional_encoding_gaussian_matrix` ~ N(0,1).  Keys are exactly the reference's
state_dict names (`model/anyref.py:116-127,161`, `build_sam.py:67-99`, HF Llama / CLIP
names under LLaVA's `model.vision_tower.vision_tower.` prefix) so the same loader
serves real checkpoints.
"""
from __future__ import annotations

from typing import Dict, Iterator, Tuple

import torch

from .config import AnyRefConfig

CLIP_PREFIX = "model.vision_tower.vision_tower.vision_model."
SAM_PREFIX = "model.visual_model."


def weight_shapes(cfg: AnyRefConfig, audio: bool = True) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """Yield (name, shape, kind) for every tensor the inference path reads.
    kind: 'w' matrix/conv weight, 'b' bias, 'g' norm gain, 'n1' unit normal, 'e' embedding."""
    c, l, s = cfg.clip, cfg.llm, cfg.sam
    p = CLIP_PREFIX
    yield p + "embeddings.class_embedding", (c.dim,), "w"
    yield p + "embeddings.patch_embedding.weight", (c.dim, 3, c.patch, c.patch), "w"
    yield p + "embeddings.position_embedding.weight", (c.n_patches + 1, c.dim), "w"
    yield p + "pre_layrnorm.weight", (c.dim,), "g"
    yield p + "pre_layrnorm.bias", (c.dim,), "b"
    for i in range(c.layers_run):
        lp = f"{p}encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            yield lp + n + ".weight", (c.dim,), "g"
            yield lp + n + ".bias", (c.dim,), "b"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield lp + f"self_attn.{n}.weight", (c.dim, c.dim), "w"
            yield lp + f"self_attn.{n}.bias", (c.dim,), "b"
        yield lp + "mlp.fc1.weight", (c.mlp, c.dim), "w"
        yield lp + "mlp.fc1.bias", (c.mlp,), "b"
        yield lp + "mlp.fc2.weight", (c.dim, c.mlp), "w"
        yield lp + "mlp.fc2.bias", (c.dim,), "b"
    yield "model.mm_projector.weight", (l.dim, c.dim), "w"
    yield "model.mm_projector.bias", (l.dim,), "b"

    yield "model.embed_tokens.weight", (l.vocab, l.dim), "e"
    for i in range(l.layers):
        lp = f"model.layers.{i}."
        yield lp + "input_layernorm.weight", (l.dim,), "g"
        yield lp + "post_attention_layernorm.weight", (l.dim,), "g"
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            yield lp + f"self_attn.{n}.weight", (l.dim, l.dim), "w"
        yield lp + "mlp.gate_proj.weight", (l.mlp, l.dim), "w"
        yield lp + "mlp.up_proj.weight", (l.mlp, l.dim), "w"
        yield lp + "mlp.down_proj.weight", (l.dim, l.mlp), "w"
    yield "model.norm.weight", (l.dim,), "g"
    yield "lm_head.weight", (l.vocab, l.dim), "w"

    yield "model.text_hidden_fcs.0.0.weight", (l.dim, l.dim), "w"
    yield "model.text_hidden_fcs.0.0.bias", (l.dim,), "b"
    yield "model.text_hidden_fcs.0.2.weight", (cfg.out_dim, l.dim), "w"
    yield "model.text_hidden_fcs.0.2.bias", (cfg.out_dim,), "b"
    if audio:
        yield "model.audio_projector.weight", (l.dim, cfg.audio_dim), "w"
        yield "model.audio_projector.bias", (l.dim,), "b"
    at = getattr(cfg, "audio_trunk", None)
    if audio and at is not None:        # ImageBind audio trunk under the reference's `model.audio_encoder.` names
        ap = "model.audio_encoder."
        pp = ap + "modality_preprocessors.audio."
        yield pp + "cls_token", (1, 1, at.dim), "w"
        yield pp + "rgbt_stem.proj.weight", (at.dim, 1, at.kernel, at.kernel), "w"
        yield pp + "rgbt_stem.norm_layer.weight", (at.dim,), "g"
        yield pp + "rgbt_stem.norm_layer.bias", (at.dim,), "b"
        yield pp + "pos_embedding_helper.pos_embed", (1, at.n_patches + 1, at.dim), "w"
        for i in range(at.blocks):
            bp = f"{ap}modality_trunks.audio.blocks.{i}."
            yield bp + "attn.in_proj_weight", (3 * at.dim, at.dim), "w"
            yield bp + "attn.in_proj_bias", (3 * at.dim,), "b"
            yield bp + "attn.bias_k", (1, 1, at.dim), "b"
            yield bp + "attn.bias_v", (1, 1, at.dim), "b"
            yield bp + "attn.out_proj.weight", (at.dim, at.dim), "w"
            yield bp + "attn.out_proj.bias", (at.dim,), "b"
            for nm in ("norm_1", "norm_2"):
                yield bp + nm + ".weight", (at.dim,), "g"
                yield bp + nm + ".bias", (at.dim,), "b"
            yield bp + "mlp.fc1.weight", (4 * at.dim, at.dim), "w"
            yield bp + "mlp.fc1.bias", (4 * at.dim,), "b"
            yield bp + "mlp.fc2.weight", (at.dim, 4 * at.dim), "w"
            yield bp + "mlp.fc2.bias", (at.dim,), "b"
        yield ap + "modality_heads.audio.0.weight", (at.dim,), "g"
        yield ap + "modality_heads.audio.0.bias", (at.dim,), "b"
        yield ap + "modality_heads.audio.2.weight", (cfg.audio_dim, at.dim), "w"
        yield ap + "modality_postprocessors.audio.1.log_logit_scale", (), "ls"

    p = SAM_PREFIX + "image_encoder."
    g = s.grid
    hd = s.dim // s.heads
    yield p + "patch_embed.proj.weight", (s.dim, 3, s.patch, s.patch), "w"
    yield p + "patch_embed.proj.bias", (s.dim,), "b"
    yield p + "pos_embed", (1, g, g, s.dim), "w"
    for i in range(s.depth):
        bp = f"{p}blocks.{i}."
        sz = g if i in s.global_idx else s.window
        for n in ("norm1", "norm2"):
            yield bp + n + ".weight", (s.dim,), "g"
            yield bp + n + ".bias", (s.dim,), "b"
        yield bp + "attn.qkv.weight", (3 * s.dim, s.dim), "w"
        yield bp + "attn.qkv.bias", (3 * s.dim,), "b"
        yield bp + "attn.proj.weight", (s.dim, s.dim), "w"
        yield bp + "attn.proj.bias", (s.dim,), "b"
        yield bp + "attn.rel_pos_h", (2 * sz - 1, hd), "w"
        yield bp + "attn.rel_pos_w", (2 * sz - 1, hd), "w"
        yield bp + "mlp.lin1.weight", (s.mlp_ratio * s.dim, s.dim), "w"
        yield bp + "mlp.lin1.bias", (s.mlp_ratio * s.dim,), "b"
        yield bp + "mlp.lin2.weight", (s.dim, s.mlp_ratio * s.dim), "w"
        yield bp + "mlp.lin2.bias", (s.dim,), "b"
    C = s.out_chans
    yield p + "neck.0.weight", (C, s.dim, 1, 1), "w"
    yield p + "neck.1.weight", (C,), "g"
    yield p + "neck.1.bias", (C,), "b"
    yield p + "neck.2.weight", (C, C, 3, 3), "w"
    yield p + "neck.3.weight", (C,), "g"
    yield p + "neck.3.bias", (C,), "b"

    p = SAM_PREFIX + "prompt_encoder."
    yield p + "pe_layer.positional_encoding_gaussian_matrix", (2, C // 2), "n1"
    yield p + "no_mask_embed.weight", (1, C), "e"

    p = SAM_PREFIX + "mask_decoder."
    yield p + "iou_token.weight", (1, C), "e"
    yield p + "mask_tokens.weight", (s.num_mask_tokens, C), "e"

    def attn(ap, internal):
        for n in ("q_proj", "k_proj", "v_proj"):
            yield ap + n + ".weight", (internal, C), "w"
            yield ap + n + ".bias", (internal,), "b"
        yield ap + "out_proj.weight", (C, internal), "w"
        yield ap + "out_proj.bias", (C,), "b"

    for i in range(s.dec_depth):
        lp = f"{p}transformer.layers.{i}."
        yield from attn(lp + "self_attn.", C)
        yield from attn(lp + "cross_attn_token_to_image.", C // 2)
        yield from attn(lp + "cross_attn_image_to_token.", C // 2)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            yield lp + n + ".weight", (C,), "g"
            yield lp + n + ".bias", (C,), "b"
        yield lp + "mlp.lin1.weight", (s.dec_mlp, C), "w"
        yield lp + "mlp.lin1.bias", (s.dec_mlp,), "b"
        yield lp + "mlp.lin2.weight", (C, s.dec_mlp), "w"
        yield lp + "mlp.lin2.bias", (C,), "b"
    yield from attn(p + "transformer.final_attn_token_to_image.", C // 2)
    yield p + "transformer.norm_final_attn.weight", (C,), "g"
    yield p + "transformer.norm_final_attn.bias", (C,), "b"
    yield p + "output_upscaling.0.weight", (C, C // 4, 2, 2), "w"
    yield p + "output_upscaling.0.bias", (C // 4,), "b"
    yield p + "output_upscaling.1.weight", (C // 4,), "g"
    yield p + "output_upscaling.1.bias", (C // 4,), "b"
    yield p + "output_upscaling.3.weight", (C // 4, C // 8, 2, 2), "w"
    yield p + "output_upscaling.3.bias", (C // 8,), "b"
    for i in range(s.num_mask_tokens):
        for j, (a, b) in enumerate(((C, C), (C, C), (C // 8, C))):
            yield f"{p}output_hypernetworks_mlps.{i}.layers.{j}.weight", (a, b), "w"
            yield f"{p}output_hypernetworks_mlps.{i}.layers.{j}.bias", (a,), "b"
    for j, (a, b) in enumerate(((C, C), (C, C), (s.num_mask_tokens, C))):
        yield f"{p}iou_prediction_head.layers.{j}.weight", (a, b), "w"
        yield f"{p}iou_prediction_head.layers.{j}.bias", (a,), "b"


def _fan_in(name: str, shape) -> int:
    """Contraction length of the op a weight feeds (what its output variance scales with)."""
    if "output_upscaling" in name and len(shape) == 4:      # ConvTranspose2d k2 s2 [C_in, C_out, 2, 2]: one tap per pixel
        return shape[0]
    n = 1
    for d in shape[1:]:
        n *= d
    return max(n, 1)


def synth_state_dict(cfg: AnyRefConfig, seed: int = 0, scale: float = 0.02, device="cpu",
                     dtype=torch.float32, round_bf16: bool = True, jitter: bool = True,
                     audio: bool = True, init: str = "normal", head_gain: float = 4.0,
                     mask_gain: float = 4.0) -> Dict[str, torch.Tensor]:
    """Seeded random weights.  With `round_bf16` every value is rounded to bf16 once (and then
    stored in `dtype`), so the fp32 CPU oracle and the bf16 GPU path see identical numbers
    (SURVEY.md §8d).  `jitter` makes norm gains / biases non-trivial so tests can see them.

    init="normal": every matrix ~ N(0, scale^2) -- SURVEY.md §8d's throughput workload.  At 7B widths it makes the
    residual stream tiny and the LM / mask logits nearly flat (range +-0.02), so a parity bound on it is close to
    vacuous for bf16.
    init="fan_in": the PARITY workload -- matrices ~ N(0, 1/fan_in) (activations stay O(1) through every layer, as
    in a trained network), `lm_head` x `head_gain` (peaked next-token logits: std ~ head_gain, like a trained LM's),
    the last hypernetwork layer x `mask_gain` (mask logits spanning several units; trained ones reach +-20,
    SURVEY.md §7), embeddings ~ N(0,1), position / rel-pos tables and biases ~ N(0, scale^2), norm gains 1 (+ jitter)."""
    assert init in ("normal", "fan_in")
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape, kind in weight_shapes(cfg, audio):
        t = torch.randn(shape, generator=gen, device=device, dtype=torch.float32)
        if kind in ("w", "e"):
            table = len(shape) < 2 or any(k in name for k in ("pos_embed", "position_embedding", "rel_pos", "class_embedding"))
            if init == "normal" or table:
                t *= scale
            elif kind == "w":
                gain = head_gain if name == "lm_head.weight" else (
                    mask_gain if "output_hypernetworks_mlps" in name and name.endswith("layers.2.weight") else 1.0)
                t *= gain / _fan_in(name, shape) ** 0.5
        elif kind == "b":
            t = t * scale if jitter else torch.zeros_like(t)
        elif kind == "g":
            t = 1.0 + (t * 0.1 if jitter else 0.0 * t)
        elif kind == "ls":
            t = torch.full(shape, 2.995732273553991, device=device)      # log(20), imagebind_model.py:425-428
        if round_bf16 and kind not in ("n1", "ls"):
            t = t.to(torch.bfloat16).to(torch.float32)
        out[name] = t.to(dtype)
    return out
