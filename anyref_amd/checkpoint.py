"""Checkpoint readers for the construction path the reference's callers take
(`eval_referseg.py:62-88`, `eval_avs_object.py:56-82`, `merge_lora.py:38-62`):

    AnyRefForCausalLM.from_pretrained(model_version, torch_dtype=..., **model_args)   HF LLaMA/LLaVA directory
    model.get_model().initialize_vision_modules(cfg)                                   CLIP tower directory
    model.get_model().initialize_anyref_modules(cfg)                                   SAM `.pth` (+ ImageBind)
    model.resize_token_embeddings(len(tokenizer))
    PeftModel.from_pretrained(model, lora_name).merge_and_unload()                     LoRA adapter directory

Everything here is host-side file parsing into a flat {reference state_dict name: tensor} dict; the
arithmetic (the LoRA merge W += (alpha / r) * B @ A, `train.py:371-396`) is plain torch on the host, done
once at load.  No HIP, no oracle.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Optional

import torch

from .config import AnyRefConfig, ClipConfig, LlmConfig, SamConfig
from .synth import CLIP_PREFIX, SAM_PREFIX

# build_sam.py:15-60: (embed_dim, depth, heads, global attention blocks) by the substring `vision_pretrained` carries
# (anyref.py:98-105)
SAM_VARIANTS = {
    "vit_b": (768, 12, 12, (2, 5, 8, 11)),
    "vit_l": (1024, 24, 16, (5, 11, 17, 23)),
    "vit_h": (1280, 32, 16, (7, 15, 23, 31)),
}


def read_tensors(path: str) -> Dict[str, torch.Tensor]:
    """One weight file: `.safetensors`, or a torch pickle (`.bin` / `.pth` / `.pt`) holding a state dict."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict) and all(
            isinstance(v, torch.Tensor) for v in sd["model"].values()):
        sd = sd["model"]
    return sd


def read_hf_dir(path: str) -> Dict[str, torch.Tensor]:
    """Every tensor of an HF `save_pretrained` directory: sharded (`*.index.json`) or single-file, safetensors
    preferred over `.bin` when both exist (HF's own order)."""
    if os.path.isfile(path):
        return read_tensors(path)
    for index, single in (("model.safetensors.index.json", "model.safetensors"),
                          ("pytorch_model.bin.index.json", "pytorch_model.bin")):
        ip = os.path.join(path, index)
        if os.path.exists(ip):
            files = sorted(set(json.load(open(ip))["weight_map"].values()))
            out: Dict[str, torch.Tensor] = {}
            for f in files:
                out.update(read_tensors(os.path.join(path, f)))
            return out
        sp = os.path.join(path, single)
        if os.path.exists(sp):
            return read_tensors(sp)
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin (or their .index.json) under {path}")


def llm_config_from_hf(hf: dict, max_seq: int = 1024) -> LlmConfig:
    """HF `LlamaConfig` fields -> LlmConfig (the reference's LLM is whatever `config.json` of `model_version` says)."""
    heads = int(hf["num_attention_heads"])
    if int(hf.get("num_key_value_heads", heads)) != heads:
        raise ValueError("grouped-query attention checkpoints are outside the reference's LLaMA-1/LLaVA-v1.1 scope")
    return LlmConfig(vocab=int(hf["vocab_size"]), dim=int(hf["hidden_size"]), heads=heads,
                     layers=int(hf["num_hidden_layers"]), mlp=int(hf["intermediate_size"]),
                     rms_eps=float(hf.get("rms_norm_eps", 1e-6)), rope_theta=float(hf.get("rope_theta", 10000.0)),
                     max_seq=max_seq)


def clip_config_from_hf(hf: Optional[dict]) -> ClipConfig:
    """`CLIPVisionConfig` (or the `vision_config` of a full `CLIPConfig`); default = ViT-L/14 (anyref.py:190-192)."""
    if not hf:
        return ClipConfig()
    v = hf.get("vision_config", hf)
    return ClipConfig(image_size=int(v.get("image_size", 224)), patch=int(v.get("patch_size", 14)),
                      dim=int(v.get("hidden_size", 1024)), heads=int(v.get("num_attention_heads", 16)),
                      layers=int(v.get("num_hidden_layers", 24)), mlp=int(v.get("intermediate_size", 4096)),
                      eps=float(v.get("layer_norm_eps", 1e-5)))


def load_clip_tower(path: str):
    """CLIP directory (`openai/clip-vit-large-patch14` layout) -> (ClipConfig, {reference name: tensor}).
    Keys `vision_model.*` (CLIPModel / CLIPVisionModel, transformers 4.x) or un-prefixed (5.x) are renamed under
    LLaVA's `model.vision_tower.vision_tower.vision_model.` prefix; the text tower is dropped."""
    cfgp = os.path.join(path, "config.json")
    cfg = clip_config_from_hf(json.load(open(cfgp)) if os.path.exists(cfgp) else None)
    raw = read_hf_dir(path)
    out = {}
    for k, v in raw.items():
        if k.startswith("vision_model."):
            out[CLIP_PREFIX + k[len("vision_model."):]] = v
        elif k.startswith(("embeddings.", "pre_layrnorm.", "encoder.", "post_layernorm.")):
            out[CLIP_PREFIX + k] = v
    if not out:
        raise ValueError(f"{path}: no CLIP vision tower weights found")
    return cfg, out


def sam_config_for(vision_pretrained: str, **over) -> SamConfig:
    for tag, (dim, depth, heads, gidx) in SAM_VARIANTS.items():
        if tag in vision_pretrained:                         # anyref.py:98-105
            return SamConfig(dim=dim, depth=depth, heads=heads, global_idx=gidx, **over)
    raise NotImplementedError(f"vision_pretrained={vision_pretrained!r} names none of vit_b / vit_l / vit_h")


def load_sam(path: str) -> Dict[str, torch.Tensor]:
    """SAM checkpoint (`build_sam.py:104-107`: a bare state dict) -> names under `model.visual_model.`."""
    return {SAM_PREFIX + k: v for k, v in read_tensors(path).items()}


def resize_token_rows(sd: Dict[str, torch.Tensor], n: int, std: float = 0.02, seed: Optional[int] = None):
    """`resize_token_embeddings` (eval_referseg.py:79): grow (or cut) `embed_tokens` and `lm_head` to n rows.  New
    rows are N(0, std^2) as HF's `_init_weights` makes them (their trained values arrive with the adapter's
    `modules_to_save`, train.py:374-381); HF draws them from the global RNG, so they are not reproducible across
    libraries -- pass `seed` for a deterministic fill."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    for name in ("model.embed_tokens.weight", "lm_head.weight"):
        w = sd[name]
        if w.shape[0] == n:
            continue
        if w.shape[0] > n:
            sd[name] = w[:n].contiguous()
            continue
        extra = torch.randn(n - w.shape[0], w.shape[1], generator=g, dtype=torch.float32) * std
        sd[name] = torch.cat([w, extra.to(w.dtype)], 0)


def _strip_peft_key(k: str) -> str:
    for pre in ("base_model.model.",):
        if k.startswith(pre):
            k = k[len(pre):]
    return k.replace(".modules_to_save.default", "").replace(".modules_to_save", "").replace(".default", "")


def merge_lora(sd: Dict[str, torch.Tensor], adapter_dir: str) -> Dict[str, int]:
    """`PeftModel.from_pretrained(model, dir).merge_and_unload()` on a flat state dict (peft 0.4.0 file layout:
    `adapter_config.json` + `adapter_model.bin|safetensors`).  LoRA pairs: W += (lora_alpha / r) * B @ A (transposed
    when `fan_in_fan_out`); `modules_to_save` tensors replace the base ones.  Returns counts for the log."""
    ac = json.load(open(os.path.join(adapter_dir, "adapter_config.json")))
    r, alpha = int(ac["r"]), float(ac["lora_alpha"])
    scaling = alpha / r
    fifo = bool(ac.get("fan_in_fan_out", False))
    for f in ("adapter_model.safetensors", "adapter_model.bin"):
        p = os.path.join(adapter_dir, f)
        if os.path.exists(p):
            ad = read_tensors(p)
            break
    else:
        raise FileNotFoundError(f"no adapter_model.safetensors / adapter_model.bin under {adapter_dir}")
    A, B, saved = {}, {}, {}
    for k, v in ad.items():
        k = _strip_peft_key(k)
        if ".original_module." in k:
            continue
        if k.endswith(".lora_A.weight"):
            A[k[: -len(".lora_A.weight")]] = v
        elif k.endswith(".lora_B.weight"):
            B[k[: -len(".lora_B.weight")]] = v
        elif ".lora_" in k:
            raise ValueError(f"unsupported adapter tensor {k} (only Linear LoRA pairs, bias='none')")
        else:
            saved[k] = v
    if set(A) != set(B):
        raise ValueError("adapter has unpaired lora_A / lora_B tensors")
    for mod in A:
        name = mod + ".weight"
        if name not in sd:
            raise KeyError(f"adapter targets {name}, which the base checkpoint does not have")
        a, b = A[mod].float(), B[mod].float()
        if a.shape[0] != r or b.shape[1] != r:
            raise ValueError(f"{mod}: LoRA rank {a.shape[0]} / {b.shape[1]} != r = {r} of adapter_config.json")
        delta = (b @ a) * scaling
        if fifo:
            delta = delta.t()
        w = sd[name]
        sd[name] = (w.float() + delta).to(w.dtype)
    for k, v in saved.items():
        sd[k] = v
    return {"lora_pairs": len(A), "modules_to_save": len(saved)}


def missing_for(cfg: AnyRefConfig, names: Iterable[str], audio: bool) -> list:
    """Reference names the inference path reads (synth.weight_shapes) that `names` lacks."""
    from .synth import weight_shapes
    have = set(names)
    return [n for n, _, _ in weight_shapes(cfg, audio) if n not in have]
