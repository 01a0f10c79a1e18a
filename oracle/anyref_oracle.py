"""CPU fp32 oracle for the AnyRef refer-segmentation inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under `anyref_amd/` may import this module; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and
only as the checker / the reported CPU baseline.

This is an independent restatement (plain torch CPU ops, fp32) of the arithmetic of

  * `model/anyref.py:647-822`  (`AnyRefForCausalLM.generate`)
  * `model/anyref.py:239-466`  (`model_forward_new`, teacher-forced, + losses `:19-68`)
  * `model/segment_anything/modeling/{image_encoder,prompt_encoder,mask_decoder,transformer,sam,common}.py`
  * the absent `model/llava/**` layer = HF `transformers==4.31.0` (`requirements.txt:29`)
    `LlamaForCausalLM` + `CLIPVisionModel` + upstream LLaVA v1.1 multimodal splice
    (call sites `model/anyref.py:341-354,704-716`).

Pinning: the reference has no tests / golden vectors of its own, so every pin is
an output of the reference's code RUN in the build container by a committed script:
  * SAM half (`sam_image_encoder`, `dense_pe`, `two_way_transformer`, `mask_decoder_predict`,
    `postprocess_masks`): `tests/golden/make_golden.py` -> `sam_*.npz` (small shapes) and
    `sam_h_width.npz` (SAM-H's real width / head_dim / window at 1024^2, 4 blocks).
  * Glue (`generate_tail`, `forward_tail`, `ref_features_*`, `text_hidden_fc`, the two losses):
    `tests/golden/make_golden_glue.py` runs the reference's `model/anyref.py` (:19-68, :96-161,
    :239-466, :647-822) on CANNED LLM outputs -> `glue_anyref.npz`.
  * Metrics (`intersection_and_union`, `avs_*`): `tests/golden/make_golden_metrics.py` runs the
    reference's `utils/utils.py` / `utils/pyutils.py` -> `metrics_ref.npz`.
  * LLaMA / CLIP half (`clip_patch_tokens`, `llama_*`, `greedy_generate`): the reference's
    `model/llava/**` is git-ignored and absent; pinned against the HF transformers 5.15 stand-in
    only (`llm_clip_hf.npz`) -- w.r.t. the reference this half is **parity unpinned**.  So is
    `splice_embeddings` / `splice_ref_rows` (the llava layer's splice, inferred from call sites).

All functions take a flat dict `w` of fp32 tensors keyed by the reference's
state_dict names (SURVEY.md §8b "Weight names").
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]

IMAGE_TOKEN_INDEX = -200
AUDIO_REF_INDEX = -300
IMG_REF_INDEX = -400

CLIP_PREFIX = "model.vision_tower.vision_tower.vision_model."
SAM_PREFIX = "model.visual_model."


def _lin(x, w: W, name: str, bias: bool = True):
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias") if bias else None)


def _ln(x, w: W, name: str, eps: float):
    return F.layer_norm(x, (x.shape[-1],), w[name + ".weight"], w[name + ".bias"], eps)


# ----------------------------------------------------------------------------------------
# CLIP ViT tower + mm_projector  (HF CLIPVisionModel; LLaVA `encode_images`, anyref.py:334)
# ----------------------------------------------------------------------------------------
def clip_patch_tokens(w: W, cfg, images: torch.Tensor) -> torch.Tensor:
    """images [B,3,S,S] -> hidden_states[select_layer][:, 1:]  [B, n_patches, Dc]."""
    c = cfg.clip
    p = CLIP_PREFIX
    B = images.shape[0]
    x = F.conv2d(images, w[p + "embeddings.patch_embedding.weight"], None, stride=c.patch)
    x = x.flatten(2).transpose(1, 2)                                   # [B, n, D]
    cls = w[p + "embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], 1) + w[p + "embeddings.position_embedding.weight"][None]
    x = _ln(x, w, p + "pre_layrnorm", c.eps)
    hd = c.dim // c.heads
    for i in range(c.layers_run):
        lp = f"{p}encoder.layers.{i}."
        h = _ln(x, w, lp + "layer_norm1", c.eps)
        q = _lin(h, w, lp + "self_attn.q_proj") * (hd ** -0.5)
        k = _lin(h, w, lp + "self_attn.k_proj")
        v = _lin(h, w, lp + "self_attn.v_proj")
        q, k, v = (t.view(B, -1, c.heads, hd).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2), -1) @ v
        a = a.transpose(1, 2).reshape(B, -1, c.dim)
        x = x + _lin(a, w, lp + "self_attn.out_proj")
        h = _ln(x, w, lp + "layer_norm2", c.eps)
        h = _lin(h, w, lp + "mlp.fc1")
        h = h * torch.sigmoid(1.702 * h)                              # quick_gelu
        x = x + _lin(h, w, lp + "mlp.fc2")
    return x[:, 1:]


def encode_images(w: W, cfg, images: torch.Tensor) -> torch.Tensor:
    """LLaVA `encode_images`: CLIP patch features -> mm_projector (Linear Dc->H)."""
    return _lin(clip_patch_tokens(w, cfg, images), w, "model.mm_projector")


# ----------------------------------------------------------------------------------------
# LLaMA decoder (HF LlamaModel 4.31 semantics: RMSNorm, rotate_half RoPE, SwiGLU)
# ----------------------------------------------------------------------------------------
def _rms(x, weight, eps):
    v = x.pow(2).mean(-1, keepdim=True)
    return weight * (x * torch.rsqrt(v + eps))


def _rope_cos_sin(cfg, positions: torch.Tensor):
    hd = cfg.llm.head_dim
    inv = 1.0 / (cfg.llm.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = positions.float()[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def splice_embeddings(w: W, cfg, input_ids: torch.Tensor, image_feats: torch.Tensor,
                      audio_feats: Optional[torch.Tensor] = None,
                      ref_feats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One sequence.  ids [L] (with negative placeholders) -> embeds [L+255, H].

    Upstream LLaVA v1.1 `prepare_inputs_labels_for_multimodal` semantics as inferred from
    the reference's call sites: the single IMAGE placeholder expands 1->256;
    audio / image-ref placeholders are replaced 1:1 by the rows of their feature tensor.
    """
    emb = w["model.embed_tokens.weight"]
    out = []
    a_i = r_i = 0
    for t in input_ids.tolist():
        if t == IMAGE_TOKEN_INDEX:
            out.append(image_feats)
        elif t == AUDIO_REF_INDEX:
            out.append(audio_feats[a_i:a_i + 1]); a_i += 1
        elif t == IMG_REF_INDEX:
            out.append(ref_feats[r_i:r_i + 1]); r_i += 1
        else:
            out.append(emb[t][None])
    return torch.cat(out, 0)


def llama_layers(w: W, cfg, x: torch.Tensor, want_last_attn: bool = False):
    """x [S,H] one sequence, full causal forward.  Returns (post-norm hidden [S,H],
    last-layer attention probs [heads,S,S] or None)."""
    c = cfg.llm
    S = x.shape[0]
    cos, sin = _rope_cos_sin(cfg, torch.arange(S))
    mask = torch.full((S, S), float("-inf")).triu(1)
    attn_last = None
    for i in range(c.layers):
        lp = f"model.layers.{i}."
        h = _rms(x, w[lp + "input_layernorm.weight"], c.rms_eps)
        q = F.linear(h, w[lp + "self_attn.q_proj.weight"]).view(S, c.heads, -1).transpose(0, 1)
        k = F.linear(h, w[lp + "self_attn.k_proj.weight"]).view(S, c.heads, -1).transpose(0, 1)
        v = F.linear(h, w[lp + "self_attn.v_proj.weight"]).view(S, c.heads, -1).transpose(0, 1)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(c.head_dim) + mask, -1)
        if want_last_attn and i == c.layers - 1:
            attn_last = a
        o = (a @ v).transpose(0, 1).reshape(S, c.dim)
        x = x + F.linear(o, w[lp + "self_attn.o_proj.weight"])
        h = _rms(x, w[lp + "post_attention_layernorm.weight"], c.rms_eps)
        g = F.linear(h, w[lp + "mlp.gate_proj.weight"])
        u = F.linear(h, w[lp + "mlp.up_proj.weight"])
        x = x + F.linear(F.silu(g) * u, w[lp + "mlp.down_proj.weight"])
    return _rms(x, w["model.norm.weight"], c.rms_eps), attn_last


class _KVState:
    """Per-sequence KV cache for the cached greedy loop (results identical to the
    reference's `use_cache=False` loop, anyref.py:171)."""

    def __init__(self, layers):
        self.k = [None] * layers
        self.v = [None] * layers
        self.n = 0


def llama_step(w: W, cfg, x: torch.Tensor, st: _KVState, want_last_attn: bool = False):
    """Append rows x [s,H] to the cache; returns (post-norm hidden [s,H], last attn [heads,s,n+s])."""
    c = cfg.llm
    s = x.shape[0]
    pos = torch.arange(st.n, st.n + s)
    cos, sin = _rope_cos_sin(cfg, pos)
    tot = st.n + s
    mask = torch.full((s, tot), float("-inf")).triu(st.n + 1)
    attn_last = None
    for i in range(c.layers):
        lp = f"model.layers.{i}."
        h = _rms(x, w[lp + "input_layernorm.weight"], c.rms_eps)
        q = F.linear(h, w[lp + "self_attn.q_proj.weight"]).view(s, c.heads, -1).transpose(0, 1)
        k = F.linear(h, w[lp + "self_attn.k_proj.weight"]).view(s, c.heads, -1).transpose(0, 1)
        v = F.linear(h, w[lp + "self_attn.v_proj.weight"]).view(s, c.heads, -1).transpose(0, 1)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        st.k[i] = k if st.k[i] is None else torch.cat([st.k[i], k], 1)
        st.v[i] = v if st.v[i] is None else torch.cat([st.v[i], v], 1)
        a = torch.softmax(q @ st.k[i].transpose(-1, -2) / math.sqrt(c.head_dim) + mask, -1)
        if want_last_attn and i == c.layers - 1:
            attn_last = a
        o = (a @ st.v[i]).transpose(0, 1).reshape(s, c.dim)
        x = x + F.linear(o, w[lp + "self_attn.o_proj.weight"])
        h = _rms(x, w[lp + "post_attention_layernorm.weight"], c.rms_eps)
        g = F.linear(h, w[lp + "mlp.gate_proj.weight"])
        u = F.linear(h, w[lp + "mlp.up_proj.weight"])
        x = x + F.linear(F.silu(g) * u, w[lp + "mlp.down_proj.weight"])
    st.n = tot
    return _rms(x, w["model.norm.weight"], c.rms_eps), attn_last


def greedy_generate(w: W, cfg, embeds: torch.Tensor, max_new_tokens: int,
                    eos_token_id: Optional[int], use_cache: bool = True,
                    want_attn: bool = False):
    """One sequence.  HF greedy search: argmax until EOS or max_new_tokens.

    Returns (new_ids list[T], hidden [S0+T-1... rows = every position the final forward
    saw], attn rows): `hidden[p]` is the post-norm last-layer state at position p, for
    p in [0, S0+T-1) exactly as `outputs.hidden_states[-1]` in anyref.py:718.
    With `want_attn`, also the head-mean last-layer attention rows [S0+T-1, S0+T-1]
    (lower-triangular, zero-padded) used by the rephrase branch (anyref.py:735-755).
    """
    emb_w = w["model.embed_tokens.weight"]
    lm = w["lm_head.weight"]
    new_ids: List[int] = []
    S0 = embeds.shape[0]
    if use_cache:
        st = _KVState(cfg.llm.layers)
        hid, att = llama_step(w, cfg, embeds, st, want_attn)
        hiddens = [hid]
        attn_rows = [att.mean(0)] if want_attn else None
        for _ in range(max_new_tokens):
            nxt = int(torch.argmax(F.linear(hiddens[-1][-1], lm)))
            new_ids.append(nxt)
            if (eos_token_id is not None and nxt == eos_token_id) or len(new_ids) == max_new_tokens:
                break
            hid, att = llama_step(w, cfg, emb_w[nxt][None], st, want_attn)
            hiddens.append(hid)
            if want_attn:
                attn_rows.append(att.mean(0))
        hidden = torch.cat(hiddens, 0)
        attn = None
        if want_attn:
            n = hidden.shape[0]
            attn = torch.zeros(n, n)
            r = 0
            for a in attn_rows:
                attn[r:r + a.shape[0], :a.shape[1]] = a
                r += a.shape[0]
        return new_ids, hidden, attn
    # reference behaviour: re-run the whole prefix for every token (anyref.py:171)
    x = embeds
    hidden = attn = None
    for _ in range(max_new_tokens):
        hidden, a = llama_layers(w, cfg, x, want_attn)
        attn = a.mean(0) if want_attn else None
        nxt = int(torch.argmax(F.linear(hidden[-1], lm)))
        new_ids.append(nxt)
        if (eos_token_id is not None and nxt == eos_token_id) or len(new_ids) == max_new_tokens:
            break
        x = torch.cat([x, emb_w[nxt][None]], 0)
    return new_ids, hidden, attn


# ----------------------------------------------------------------------------------------
# SAM image encoder (image_encoder.py:17-426)
# ----------------------------------------------------------------------------------------
def resample_rel_pos(rel_pos: torch.Tensor, length: int) -> torch.Tensor:
    """The table-length fix-up of get_rel_pos (image_encoder.py:333-345): a table [L, C] whose L is not the
    2*size-1 the block needs is resampled along L by `F.interpolate(mode="linear")` (align_corners=False): output
    row i sits at src = max((i + 0.5) * L / length - 0.5, 0) and blends rows floor(src) and floor(src) + 1
    (clamped to L - 1) with weights (1 - frac, frac), all in fp32."""
    L = rel_pos.shape[0]
    if L == length:
        return rel_pos
    scale = torch.tensor(L, dtype=torch.float32) / length
    src = (scale * (torch.arange(length, dtype=torch.float32) + 0.5) - 0.5).clamp_min(0.0)
    i0 = src.floor().long().clamp_max(L - 1)
    i1 = (i0 + 1).clamp_max(L - 1)
    lam = (src - i0.float()).clamp(0.0, 1.0)[:, None]
    return (1.0 - lam) * rel_pos[i0] + lam * rel_pos[i1]


def _rel_pos_table(rel_pos: torch.Tensor, size: int) -> torch.Tensor:
    """get_rel_pos (image_encoder.py:321-351) for q_size == k_size == size: R[q, k] = rel_pos[q - k + size - 1],
    on the table resampled to 2*size-1 rows when the checkpoint's has another length."""
    rel_pos = resample_rel_pos(rel_pos, 2 * size - 1)
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + (size - 1)
    return rel_pos[idx]                                                # [size, size, hd]


def _sam_attention(w: W, p: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """Attention.forward (image_encoder.py:231-260).  x [B', h, w, D]."""
    Bp, Hh, Ww, D = x.shape
    hd = D // heads
    qkv = _lin(x, w, p + "qkv").reshape(Bp, Hh * Ww, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, Bp * heads, Hh * Ww, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    Rh = _rel_pos_table(w[p + "rel_pos_h"], Hh)
    Rw = _rel_pos_table(w[p + "rel_pos_w"], Ww)
    rq = q.reshape(Bp * heads, Hh, Ww, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(-1, Hh, Ww, Hh, Ww) + rel_h[..., :, None] + rel_w[..., None, :]).view(
        -1, Hh * Ww, Hh * Ww)
    attn = attn.softmax(-1)
    o = (attn @ v).view(Bp, heads, Hh, Ww, hd).permute(0, 2, 3, 1, 4).reshape(Bp, Hh, Ww, D)
    return _lin(o, w, p + "proj")


def sam_image_encoder(w: W, cfg, images: torch.Tensor) -> torch.Tensor:
    """[B,3,S,S] -> [B,256,g,g] (NCHW), ImageEncoderViT.forward (image_encoder.py:110-125)."""
    s = cfg.sam
    p = SAM_PREFIX + "image_encoder."
    x = F.conv2d(images, w[p + "patch_embed.proj.weight"], w[p + "patch_embed.proj.bias"],
                 stride=s.patch).permute(0, 2, 3, 1)
    x = x + w[p + "pos_embed"]
    B, Hh, Ww, D = x.shape
    for i in range(s.depth):
        bp = f"{p}blocks.{i}."
        ws = 0 if i in s.global_idx else s.window
        h = _ln(x, w, bp + "norm1", 1e-6)
        if ws > 0:
            ph, pw = (ws - Hh % ws) % ws, (ws - Ww % ws) % ws
            h = F.pad(h, (0, 0, 0, pw, 0, ph))                        # zero pad AFTER the norm
            Hp, Wp = Hh + ph, Ww + pw
            h = h.view(B, Hp // ws, ws, Wp // ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, D)
        h = _sam_attention(w, bp + "attn.", h, s.heads)
        if ws > 0:
            h = h.view(B, Hp // ws, Wp // ws, ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)
            h = h[:, :Hh, :Ww]
        x = x + h
        h = _ln(x, w, bp + "norm2", 1e-6)
        h = _lin(F.gelu(_lin(h, w, bp + "mlp.lin1")), w, bp + "mlp.lin2")
        x = x + h
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, w[p + "neck.0.weight"])
    x = _ln2d(x, w[p + "neck.1.weight"], w[p + "neck.1.bias"])
    x = F.conv2d(x, w[p + "neck.2.weight"], padding=1)
    x = _ln2d(x, w[p + "neck.3.weight"], w[p + "neck.3.bias"])
    return x


def _ln2d(x, weight, bias, eps: float = 1e-6):
    """LayerNorm2d (common.py:31-43): normalise over the channel axis of NCHW."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight[:, None, None] * x + bias[:, None, None]


# ----------------------------------------------------------------------------------------
# Prompt encoder text path + dense PE (prompt_encoder.py:67-76,140-229)
# ----------------------------------------------------------------------------------------
def dense_pe(w: W, cfg) -> torch.Tensor:
    """get_dense_pe -> [1, 256, g, g]."""
    g = cfg.sam.grid
    G = w[SAM_PREFIX + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    ax = (torch.arange(g, dtype=torch.float32) + 0.5) / g
    coords = torch.stack([ax[None, :].expand(g, g), ax[:, None].expand(g, g)], -1)   # (x, y)
    c = (2 * coords - 1) @ G
    c = 2 * math.pi * c
    return torch.cat([c.sin(), c.cos()], -1).permute(2, 0, 1)[None]


def prompt_encoder_text(w: W, cfg, text_embeds: torch.Tensor):
    """text_embeds [n,1,256] -> sparse [n,1,256], dense [n,256,g,g] (prompt_encoder.py:162-186)."""
    g = cfg.sam.grid
    n = text_embeds.shape[0]
    dense = w[SAM_PREFIX + "prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(n, -1, g, g)
    return text_embeds, dense


# ----------------------------------------------------------------------------------------
# Mask decoder (mask_decoder.py:116-179, transformer.py:62-242)
# ----------------------------------------------------------------------------------------
def _dec_attn(w: W, p: str, q, k, v, heads: int):
    q, k, v = _lin(q, w, p + "q_proj"), _lin(k, w, p + "k_proj"), _lin(v, w, p + "v_proj")
    b, n, c = q.shape
    sh = lambda t: t.reshape(b, t.shape[1], heads, c // heads).transpose(1, 2)
    q, k, v = sh(q), sh(k), sh(v)
    a = torch.softmax(q @ k.permute(0, 1, 3, 2) / math.sqrt(c // heads), -1)
    o = (a @ v).transpose(1, 2).reshape(b, n, c)
    return _lin(o, w, p + "out_proj")


def two_way_transformer(w: W, cfg, src: torch.Tensor, pos: torch.Tensor, tokens: torch.Tensor):
    """src,pos [n,C,g,g]; tokens [n,N,C] -> (queries [n,N,C], keys [n,g*g,C])."""
    s = cfg.sam
    p = SAM_PREFIX + "mask_decoder.transformer."
    keys = src.flatten(2).permute(0, 2, 1)
    kpe = pos.flatten(2).permute(0, 2, 1)
    queries, qpe = tokens, tokens
    for i in range(s.dec_depth):
        lp = f"{p}layers.{i}."
        if i == 0:
            queries = _dec_attn(w, lp + "self_attn.", queries, queries, queries, s.dec_heads)
        else:
            q = queries + qpe
            queries = queries + _dec_attn(w, lp + "self_attn.", q, q, queries, s.dec_heads)
        queries = _ln(queries, w, lp + "norm1", 1e-5)
        q, k = queries + qpe, keys + kpe
        queries = queries + _dec_attn(w, lp + "cross_attn_token_to_image.", q, k, keys, s.dec_heads)
        queries = _ln(queries, w, lp + "norm2", 1e-5)
        m = _lin(F.relu(_lin(queries, w, lp + "mlp.lin1")), w, lp + "mlp.lin2")
        queries = _ln(queries + m, w, lp + "norm3", 1e-5)
        q, k = queries + qpe, keys + kpe
        keys = keys + _dec_attn(w, lp + "cross_attn_image_to_token.", k, q, queries, s.dec_heads)
        keys = _ln(keys, w, lp + "norm4", 1e-5)
    q, k = queries + qpe, keys + kpe
    queries = queries + _dec_attn(w, p + "final_attn_token_to_image.", q, k, keys, s.dec_heads)
    queries = _ln(queries, w, p + "norm_final_attn", 1e-5)
    return queries, keys


def _mlp3(w: W, p: str, x, n: int = 3):
    for i in range(n):
        x = _lin(x, w, f"{p}layers.{i}")
        if i < n - 1:
            x = F.relu(x)
    return x


def mask_decoder_predict(w: W, cfg, image_embedding: torch.Tensor, image_pe: torch.Tensor,
                         sparse: torch.Tensor, dense: torch.Tensor):
    """MaskDecoder.predict_masks.  image_embedding [1,C,g,g], sparse [n,Ns,C], dense [n,C,g,g]
    -> masks [n,4,4g,4g], iou [n,4]."""
    s = cfg.sam
    p = SAM_PREFIX + "mask_decoder."
    n = sparse.shape[0]
    out_tok = torch.cat([w[p + "iou_token.weight"], w[p + "mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok[None].expand(n, -1, -1), sparse], 1)
    src = image_embedding.expand(n, -1, -1, -1) + dense
    pos = image_pe.expand(n, -1, -1, -1)
    b, c, h, ww = src.shape
    hs, keys = two_way_transformer(w, cfg, src, pos, tokens)
    iou_tok = hs[:, 0]
    mask_tok = hs[:, 1:1 + s.num_mask_tokens]
    x = keys.transpose(1, 2).reshape(b, c, h, ww)
    x = F.conv_transpose2d(x, w[p + "output_upscaling.0.weight"], w[p + "output_upscaling.0.bias"], stride=2)
    x = F.gelu(_ln2d(x, w[p + "output_upscaling.1.weight"], w[p + "output_upscaling.1.bias"]))
    x = F.gelu(F.conv_transpose2d(x, w[p + "output_upscaling.3.weight"], w[p + "output_upscaling.3.bias"], stride=2))
    hyper = torch.stack([_mlp3(w, f"{p}output_hypernetworks_mlps.{i}.", mask_tok[:, i])
                         for i in range(s.num_mask_tokens)], 1)
    b, c2, h2, w2 = x.shape
    masks = (hyper @ x.view(b, c2, h2 * w2)).view(b, s.num_mask_tokens, h2, w2)   # b may be 0 (a row without [SEG])
    iou = _mlp3(w, p + "iou_prediction_head.", iou_tok)
    return masks, iou


def postprocess_masks(cfg, masks: torch.Tensor, input_size: Sequence[int], original_size: Sequence[int]):
    """Sam.postprocess_masks (sam.py:137-172)."""
    S = cfg.sam.img_size
    m = F.interpolate(masks.float(), (S, S), mode="bilinear", align_corners=False)
    m = m[..., : input_size[0], : input_size[1]]
    return F.interpolate(m, tuple(original_size), mode="bilinear", align_corners=False)


def sam_decode(w: W, cfg, image_embedding: torch.Tensor, pred_embeddings: torch.Tensor,
               resized_size, orig_size):
    """Per-image tail of generate (anyref.py:797-819): pred_embeddings [n,256] -> [n,H,W] logits."""
    sparse, dense = prompt_encoder_text(w, cfg, pred_embeddings[:, None])
    low, _ = mask_decoder_predict(w, cfg, image_embedding, dense_pe(w, cfg), sparse, dense)
    low = low[:, 0:1]                                                  # multimask_output=False
    return postprocess_masks(cfg, low, resized_size, orig_size)[:, 0], low


# ----------------------------------------------------------------------------------------
# [SEG] hand-off + top level (anyref.py:647-822, :239-466)
# ----------------------------------------------------------------------------------------
def text_hidden_fc(w: W, h: torch.Tensor) -> torch.Tensor:
    """text_hidden_fcs[0]: Linear(H,H) -> ReLU -> Linear(H,256) (anyref.py:116-124)."""
    return _lin(F.relu(_lin(h, w, "model.text_hidden_fcs.0.0")), w, "model.text_hidden_fcs.0.2")


def _is_seg(cfg, ids: torch.Tensor) -> torch.Tensor:
    lo, hi = cfg.seg_range()
    return (ids >= lo) & (ids <= hi)


def pool_ref_tokens(f: torch.Tensor, n_out: int = 4) -> torch.Tensor:
    """[b, 256, c] -> mean over groups of 16 -> [b, 16, c] -> (if 16 != IMG_REF_NUM) mean over groups of
    IMG_REF_NUM -> [b, IMG_REF_NUM, c]   (anyref.py:335-338, :697-700)."""
    b, ll, c = f.shape
    f = f.reshape(b, ll // 16, 16, c).mean(dim=2)
    if f.shape[1] != n_out:
        f = f.reshape(b, n_out, n_out, c).mean(dim=2)
    return f


def ref_features_generate(encode, ref_images, bs: int, n_out: int = 4):
    """What `generate` hands to the (absent) llava layer as `ref_images=` (anyref.py:681-702).
    `encode(x [n,3,S,S]) -> [n,256,H]` stands for `self.encode_images`.  NOTE the asymmetry, kept as the
    reference has it: LIST items go down UNPOOLED ([256,H] each, :691-692; 1-D RoI coordinates pass
    through, :688-689), a TENSOR batch is pooled 256 -> 16 -> IMG_REF_NUM (:695-700)."""
    if ref_images is None:
        return None
    if isinstance(ref_images, list):
        out = []
        for r in ref_images:
            if r is None:
                out.append(None)
            elif r.dim() == 1:
                out.append(r)
            else:
                out.append(encode(r[None])[0])
        return out
    if ref_images.shape[0] == bs and ref_images.ndim == 4:
        return pool_ref_tokens(encode(ref_images), n_out)
    raise NotImplementedError


def ref_features_forward(encode, ref_images, n_out: int = 4):
    """The teacher-forced twin (anyref.py:319-339): a list only, every image item POOLED to IMG_REF_NUM rows."""
    if ref_images is None:
        return None
    out = []
    for r in ref_images:
        if r is None:
            out.append(None)
        elif r.dim() == 1:
            out.append(r)
        else:
            out.append(pool_ref_tokens(encode(r[None]), n_out)[0])
    return out


def splice_ref_rows(f: Optional[torch.Tensor], n_slots: int):
    """The absent llava layer receives [256,H] (generate, list form) or [IMG_REF_NUM,H] rows for the IMG_REF_NUM
    `<img_ref>` placeholders of a prompt (utils/coco20i.py:319 "put 4 * <ref_img>").  Its source is not in the
    reference; this build's reading (INFERRED, unpinned): unpooled features are pooled exactly as the forward
    path pools them (:335-338) before the 1:1 replacement."""
    if f is None or f.shape[0] == n_slots:
        return f
    return pool_ref_tokens(f[None], n_slots)[0]


def generate_tail(w: W, cfg, output_ids: List[torch.Tensor], prompt_lens: Sequence[int],
                  hiddens: List[torch.Tensor], attns: Optional[List[Optional[torch.Tensor]]],
                  sam_images, sam_resized_sizes, height, width, image_embeddings=None):
    """Everything `generate` does AFTER `super().generate` returns (anyref.py:718-822), for per-sample LLM
    outputs: `output_ids[b]` [L_b+T_b], `hiddens[b]` = `outputs.hidden_states[-1][b]` [L_b+T_b-1+255, H],
    `attns[b]` = `outputs.attentions[-1][b]` [heads,S,S] (or its head mean [S,S]).

    Pinned by `tests/golden/glue_*.npz`: the reference's own lines run on canned LLM outputs.

    Batch convention of this build: every row is a batch of one.  For a batch of one the reference rephrases
    the FIRST [SEG] only (`for i in range(bs)` indexes the flattened [SEG] list, :739-741,:768-769); in a real
    batch the reference pairs the i-th [SEG] of the flattened list with sample i's states, which coincides
    with this whenever every row holds exactly one [SEG]."""
    B = len(output_ids)
    seg_hidden, seg_batch = [], []
    for b in range(B):
        ids, hidden = output_ids[b], hiddens[b]
        pos = torch.where(_is_seg(cfg, ids[1:]))[0]                       # :723-726
        for j, p_ in enumerate(pos.tolist()):
            h = hidden[p_ + 255].clone()                                   # :758
            if cfg.rephrase_weight > 0 and j == 0:
                a = attns[b]
                if a.dim() == 3:
                    a = a.mean(0)                                          # :748
                s0, e0 = prompt_lens[b] - 1 + 255, p_ + 255                # :745, :741
                a = a[e0, s0:e0]
                a = a / a.sum()                                            # :749-750
                h = h + (hidden[s0:e0] * a[:, None]).sum(0) * cfg.rephrase_weight   # :754, :769
            seg_hidden.append(h)
            seg_batch.append(b)
    if not seg_hidden:                                                     # :729-730
        return dict(pred_masks=None, low_res=None, pred_embeddings=None)
    pred = text_hidden_fc(w, torch.stack(seg_hidden))                      # :770
    # :793 (`image_embeddings`: a caller that runs several prompts over one image may pass the encoder output in)
    img_emb = sam_image_encoder(w, cfg, sam_images) if image_embeddings is None else image_embeddings
    seg_batch_t = torch.tensor(seg_batch)
    masks, lows = [], []
    for b in range(B):                                                     # :797-819
        pe = pred[seg_batch_t == b]
        m, low = sam_decode(w, cfg, img_emb[b:b + 1], pe, sam_resized_sizes[b], (height[b], width[b]))
        masks.append(m); lows.append(low)
    return dict(pred_masks=masks, low_res=lows, pred_embeddings=pred, image_embeddings=img_emb)


def anyref_generate(w: W, cfg, clip_images, input_ids: List[torch.Tensor], sam_images,
                    sam_resized_sizes, height, width, audio_embeds=None, ref_feats=None,
                    max_new_tokens: int = 128, use_cache: bool = True, eos: bool = True):
    """Restatement of AnyRefForCausalLM.generate for a list of per-sample prompts (each run
    exactly as the reference runs a batch of one).  `audio_embeds[b]`: ImageBind embedding
    [3,1024] or None (the encoder itself is outside the path).
    `ref_feats[b]`: what `ref_features_generate` returns for row b (rows [256,H] or [IMG_REF_NUM,H]) or None.

    Returns dict(output_ids=list[Tensor], pred_masks=list[Tensor[n,H,W]] or None, low_res=...,
    hidden=list, pred_embeddings=list)."""
    B = len(input_ids)
    img_feats = encode_images(w, cfg, clip_images)
    out_ids, hiddens, attns = [], [], []
    for b in range(B):
        af = None
        if audio_embeds is not None and audio_embeds[b] is not None:
            af = _lin(audio_embeds[b], w, "model.audio_projector")          # :673
        rf = None
        if ref_feats is not None and ref_feats[b] is not None:
            rf = splice_ref_rows(ref_feats[b], int((input_ids[b] == IMG_REF_INDEX).sum()))
        emb = splice_embeddings(w, cfg, input_ids[b], img_feats[b], af, rf)
        new_ids, hidden, attn = greedy_generate(
            w, cfg, emb, max_new_tokens, cfg.eos_token_id if eos else None, use_cache,
            want_attn=cfg.rephrase_weight > 0)
        out_ids.append(torch.cat([input_ids[b], torch.tensor(new_ids, dtype=torch.long)]))
        hiddens.append(hidden)
        attns.append(attn)
    tail = generate_tail(w, cfg, out_ids, [len(r) for r in input_ids], hiddens, attns, sam_images,
                         sam_resized_sizes, height, width)
    return dict(output_ids=out_ids, hidden=hiddens, **tail)


def dice_loss(inputs, targets, num_masks, scale=1000, eps=1e-6):
    """anyref.py:19-47 -- the LIVE body (:35-37,:43-47); `scale` and `eps` are dead arguments there (the
    scaled form is commented out, :38-42)."""
    inputs = inputs.sigmoid().flatten(1, 2)
    targets = targets.flatten(1, 2)
    num = 2 * (inputs * targets).sum(-1)
    den = inputs.sum(-1) + targets.sum(-1)
    loss = 1 - (num + 1) / (den + 1)
    return loss.sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks):
    """anyref.py:51-68."""
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.flatten(1, 2).mean(1).sum() / (num_masks + 1e-8)


def forward_tail(w: W, cfg, input_ids: List[torch.Tensor], labels: List[torch.Tensor],
                 hiddens: List[torch.Tensor], attns: Optional[List[Optional[torch.Tensor]]], lm_loss,
                 sam_images, sam_resized_sizes, gt_masks, height, width,
                 dice_loss_weight=0.5, bce_loss_weight=2.0):
    """Everything `model_forward_new` does around `super().forward` (anyref.py:273-282 [SEG] search with the
    `pos - 1 + 255` offset, :356-466 hand-off, rephrase, mask decode, BCE + Dice), for per-sample LLM outputs
    (`hiddens[b]` = `output.hidden_states[-1][b]`, `attns[b]` = `output.attentions[-1][b]`, `lm_loss` =
    `output.loss`).  Pinned by `tests/golden/glue_*.npz`.  Same batch-of-one convention as `generate_tail`."""
    B = len(input_ids)
    seg_hidden, seg_batch = [], []
    for b in range(B):
        hidden = hiddens[b]
        pos = torch.where(_is_seg(cfg, input_ids[b]))[0]                  # :273-276
        for j, p_ in enumerate(pos.tolist()):
            e0 = p_ - 1 + 255                                              # :282
            h = hidden[e0].clone()
            if cfg.rephrase_weight > 0 and j == 0:
                s0 = int(torch.where(labels[b] > 0)[0][0]) - 1 + 255       # :378
                a = attns[b]
                if a.dim() == 3:
                    a = a.mean(0)
                a = a[e0, s0:e0]
                a = a / a.sum()
                h = h + (hidden[s0:e0] * a[:, None]).sum(0) * cfg.rephrase_weight
            seg_hidden.append(h); seg_batch.append(b)
    if not seg_hidden:                                                     # :356-365
        return dict(loss=lm_loss, lm_loss=lm_loss)
    pred = text_hidden_fc(w, torch.stack(seg_hidden))
    img_emb = sam_image_encoder(w, cfg, sam_images)
    seg_batch_t = torch.tensor(seg_batch)
    masks = []
    ce = dice = 0.0
    nm = 0
    for b in range(B):
        m, _ = sam_decode(w, cfg, img_emb[b:b + 1], pred[seg_batch_t == b], sam_resized_sizes[b],
                          (height[b], width[b]))
        masks.append(m)
        if gt_masks is not None:                                           # :432-446
            gt = gt_masks[b].to(m)
            pm = m
            if pm.shape[-2:] != gt.shape[-2:]:
                pm = F.interpolate(pm[None], size=gt.shape[-2:], mode="bilinear", align_corners=False)[0]
            ce = ce + sigmoid_ce_loss(pm, gt, gt.shape[0]) * gt.shape[0]
            dice = dice + dice_loss(pm, gt, gt.shape[0]) * gt.shape[0]
            nm += gt.shape[0]
    out = dict(lm_loss=lm_loss, pred_masks=masks, pred_embeddings=pred)
    if gt_masks is not None:                                               # :448-466
        ce = bce_loss_weight * ce / (nm + 1e-8)
        dice = dice_loss_weight * dice / (nm + 1e-8)
        out.update(ce_loss=ce, dice_loss=dice, mask_loss=ce + dice, loss=lm_loss + ce + dice)
    return out


def anyref_forward(w: W, cfg, clip_images, sam_images, input_ids: List[torch.Tensor],
                   labels: List[torch.Tensor], sam_resized_sizes, gt_masks, height, width,
                   audio_embeds=None, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0):
    """Teacher-forced `model_forward_new` (anyref.py:239-466) for unpadded per-sample prompts."""
    B = len(input_ids)
    img_feats = encode_images(w, cfg, clip_images)
    lm_num, lm_den = 0.0, 0
    hiddens, attns = [], []
    for b in range(B):
        af = None
        if audio_embeds is not None and audio_embeds[b] is not None:
            af = _lin(audio_embeds[b], w, "model.audio_projector")
        emb = splice_embeddings(w, cfg, input_ids[b], img_feats[b], af, None)
        hidden, attn = llama_layers(w, cfg, emb, cfg.rephrase_weight > 0)
        hiddens.append(hidden)
        attns.append(attn)
        # HF causal-LM loss with the image span labelled IGNORE (LLaVA splice semantics)
        ids = input_ids[b]
        ip = int(torch.where(ids == IMAGE_TOKEN_INDEX)[0][0])
        lab = torch.cat([labels[b][:ip], torch.full((256,), -100, dtype=torch.long), labels[b][ip + 1:]])
        logits = F.linear(hidden, w["lm_head.weight"])
        valid = lab[1:] != -100
        if valid.any():
            lm_num = lm_num + F.cross_entropy(logits[:-1][valid], lab[1:][valid], reduction="sum")
            lm_den += int(valid.sum())
    lm_loss = lm_num / max(lm_den, 1)
    out = forward_tail(w, cfg, input_ids, labels, hiddens, attns, lm_loss, sam_images, sam_resized_sizes, gt_masks,
                       height, width, dice_loss_weight, bce_loss_weight)
    out["hidden"] = hiddens
    return out


# ---------------------------------------------------------------------------------------------
# the steps right before / after the path (checkers for anyref_amd/evalops.py)
# ---------------------------------------------------------------------------------------------
def intersection_and_union(output: torch.Tensor, target: torch.Tensor, K: int, ignore_index: int = 255):
    """utils/utils.py:79-91 intersectionAndUnionGPU, restated on CPU (integer label maps in, float counts out)."""
    assert output.dim() in (1, 2, 3) and output.shape == target.shape
    output = output.reshape(-1).clone().to(torch.int64)
    target = target.reshape(-1).to(torch.int64)
    output[target == ignore_index] = ignore_index
    intersection = output[output == target]
    area_intersection = torch.histc(intersection.float(), bins=K, min=0, max=K - 1)
    area_output = torch.histc(output.float(), bins=K, min=0, max=K - 1)
    area_target = torch.histc(target.float(), bins=K, min=0, max=K - 1)
    return area_intersection, area_output + area_target - area_intersection, area_target


def eval_mask_counts(pred_logits: torch.Tensor, gt_mask: torch.Tensor):
    """eval_referseg.py:189-208: threshold the logits, then intersectionAndUnionGPU(pred, gt, 2, 255)."""
    pred = (torch.sigmoid(pred_logits.float()) > 0.5).int()
    return intersection_and_union(pred, gt_mask.int(), 2, ignore_index=255)


def sam_preprocess(image_hwc_u8: torch.Tensor, sam_image_size: int = 1024,
                   pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)):
    """utils/refer_seg.py:560-570 on `torch.from_numpy(img).permute(2, 0, 1)` (`:588-590`)."""
    x = image_hwc_u8.permute(2, 0, 1).contiguous().float()
    x = (x - torch.tensor(pixel_mean).view(-1, 1, 1)) / torch.tensor(pixel_std).view(-1, 1, 1)
    h, w = x.shape[-2:]
    return F.pad(x, (0, sam_image_size - w, 0, sam_image_size - h))


def avs_mask_iou(pred_logits: torch.Tensor, target: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """utils/pyutils.py:163-190 mask_iou (size_average form; caller eval_avs_object.py:168): per mask,
    |P & G| / |P | G| with P = sigmoid > 0.5; a mask whose ground truth is empty scores the fraction of
    pixels it (correctly) leaves empty; mean over the N masks."""
    assert pred_logits.dim() == 3 and pred_logits.shape == target.shape
    n, npix = pred_logits.shape[0], pred_logits.shape[-1] * pred_logits.shape[-2]
    empty_gt = target.sum(2).sum(1) == 0
    p = (torch.sigmoid(pred_logits) > 0.5).int()
    inter = (p * target).sum(2).sum(1)
    union = torch.max(p, target).sum(2).sum(1)
    both_empty = ((1 - target) * (1 - p)).sum(2).sum(1)
    inter[empty_gt] = both_empty[empty_gt]
    union[empty_gt] = npix
    return torch.sum(inter / (union + eps)) / n


def avs_pr_curve(prob: torch.Tensor, gt: torch.Tensor, num: int):
    """utils/pyutils.py:223-236 _eval_pr on CPU tensors: precision / recall at `num` thresholds."""
    prec, recall = torch.zeros(num), torch.zeros(num)
    th = torch.linspace(0, 1 - 1e-10, num)
    for i in range(num):
        passed = (prob >= th[i]).float()
        tp = (passed * gt).sum()
        prec[i], recall[i] = tp / (passed.sum() + 1e-20), tp / (gt.sum() + 1e-20)
    return prec, recall


def avs_fmeasure(pred_logits: torch.Tensor, gt: torch.Tensor, pr_num: int = 255) -> float:
    """utils/pyutils.py:193-220 Eval_Fmeasure without its (empty) log file: max over thresholds of the
    F_beta curve (beta^2 = 0.3) averaged over the masks whose ground truth is not empty."""
    prob = torch.sigmoid(pred_logits)
    beta2 = 0.3
    total, used = 0.0, 0
    score = torch.zeros(pr_num)
    for i in range(prob.shape[0]):
        if torch.mean(gt[i]) == 0.0:
            continue
        prec, recall = avs_pr_curve(prob[i], gt[i], pr_num)
        f = (1 + beta2) * prec * recall / (beta2 * prec + recall)
        f[f != f] = 0
        total = total + f
        used += 1
        score = total / used
    return score.max().item()
