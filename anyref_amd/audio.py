"""ImageBind audio trunk as a plain PyTorch(-ROCm) module.

north_star keeps the audio/image binding a PyTorch-ROCm preprocessing step: this module produces
the `[1, 3, 1024]` embedding that crosses into the HIP path as a device pointer
(`anyref_project_audio`).  It is a build-owned restatement of the part of
ImageBind that AnyRef keeps (`model/anyref.py:140-161` deletes every other modality):

  stem   Conv2d(1, 768, k=16, s=10, bias=False) on mel [*,1,128,204] -> 12x19 = 228 patches,
         LayerNorm(768)                                     (imagebind_model.py:175-192,
                                                             multimodal_preprocessors.py:121-157)
  tokens [CLS] + learnable pos_embed [1,229,768]             (multimodal_preprocessors.py:255-271)
  trunk  12 pre-LN blocks, nn.MultiheadAttention(768, 12, add_bias_kv=True), GELU MLP x4,
         LayerNorm eps 1e-6                                  (imagebind_model.py:331-338,
                                                             transformer.py:94-170)
  head   LayerNorm(eps 1e-6) -> CLS -> Linear(768, 1024, bias=False) -> L2-normalise -> x20
                                                             (imagebind_model.py:391-395,425-428)

Parameter names match the reference's `audio_encoder` state_dict, so
`load_state_dict({k[len("model.audio_encoder."):]: v ...})` of an AnyRef checkpoint works.
Parity: pinned at the real audio size (768 / 12 blocks / 12 heads / 1024-d head, 3 clips of 128 x 204) against
outputs of the reference's own `ImageBindModel.get_audio_feature`, run in the build container with inert
stand-ins for the three imports this image lacks (`tests/golden/make_golden_audio.py` -> `imagebind_audio.npz`).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Block(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(dim, heads, bias=True, add_bias_kv=True)
        self.norm_1 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, 4 * dim)
        self.mlp.fc2 = nn.Linear(4 * dim, dim)
        self.norm_2 = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, x):                               # [L, B, D] (sequence first, as the reference)
        h = self.norm_1(x)
        x = x + self.attn(h, h, h, need_weights=False)[0]
        h = self.norm_2(x)
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(h)))


class _Stem(nn.Module):
    def __init__(self, dim: int, kernel: int, stride: int):
        super().__init__()
        self.proj = nn.Conv2d(1, dim, kernel_size=kernel, stride=stride, bias=False)
        self.norm_layer = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm_layer(self.proj(x).flatten(2).transpose(1, 2))


class ImageBindAudio(nn.Module):
    """`get_audio_feature(mel)` with the reference's return convention
    (`imagebind_model.py:477-511`): (LN'd CLS feature [B,S,768], embedding [B,S,1024])."""

    def __init__(self, dim=768, blocks=12, heads=12, out_dim=1024, mel_bins=128, target_len=204, kernel=16,
                 stride=10, logit_scale=20.0):
        super().__init__()
        gh = (mel_bins - kernel) // stride + 1
        gw = (target_len - kernel) // stride + 1
        pre = nn.Module()
        pre.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        pre.rgbt_stem = _Stem(dim, kernel, stride)
        pre.pos_embedding_helper = nn.Module()
        pre.pos_embedding_helper.pos_embed = nn.Parameter(torch.zeros(1, gh * gw + 1, dim))
        self.modality_preprocessors = nn.ModuleDict({"audio": pre})
        trunk = nn.Module()
        trunk.blocks = nn.Sequential(*[_Block(dim, heads) for _ in range(blocks)])
        self.modality_trunks = nn.ModuleDict({"audio": trunk})
        self.modality_heads = nn.ModuleDict({"audio": nn.Sequential(
            nn.LayerNorm(dim, eps=1e-6), nn.Identity(), nn.Linear(dim, out_dim, bias=False))})
        post = nn.Sequential(nn.Identity(), nn.Module())
        post[1].register_buffer("log_logit_scale", torch.tensor(math.log(logit_scale)))
        self.modality_postprocessors = nn.ModuleDict({"audio": post})
        self.out_dim = out_dim

    def load_reference_state_dict(self, sd):
        """Load `imagebind_huge.pth` (full ImageBind: the other five modalities are skipped, anyref.py:142-147) or an
        AnyRef checkpoint's `model.audio_encoder.*` slice."""
        pre = "model.audio_encoder."
        own = set(self.state_dict().keys())
        pick = {}
        for k, v in sd.items():
            k = k[len(pre):] if k.startswith(pre) else k
            if k in own:
                pick[k] = v
        missing = own - set(pick)
        if missing:
            raise KeyError(f"audio trunk weights missing: {sorted(missing)[:4]} ...")
        self.load_state_dict(pick, strict=True)
        return self

    @torch.no_grad()
    def get_audio_feature(self, inputs: torch.Tensor, modality_type=None):
        x = inputs
        reduce_list = x.ndim >= 5                      # [B, S clips, 1, mel, T]
        if reduce_list:
            B, S = x.shape[:2]
            x = x.reshape(B * S, *x.shape[2:])
        pre = self.modality_preprocessors["audio"]
        tok = pre.rgbt_stem(x)
        tok = torch.cat([pre.cls_token.expand(tok.shape[0], -1, -1), tok], 1) + pre.pos_embedding_helper.pos_embed
        h = self.modality_trunks["audio"].blocks(tok.transpose(0, 1)).transpose(0, 1)
        head = self.modality_heads["audio"]
        feat = head[0](h)[:, 0]                         # LayerNorm -> SelectElement(0)
        emb = head[2](feat)
        scale = torch.clip(self.modality_postprocessors["audio"][1].log_logit_scale.exp(), max=100.0)
        emb = F.normalize(emb, dim=-1) * scale
        if reduce_list:
            feat, emb = feat.reshape(B, S, -1), emb.reshape(B, S, -1)
        return feat, emb
