"""The construction path the reference's callers take (eval_referseg.py:62-88, merge_lora.py:38-62) on the mirror,
with NO GPU: a tiny HF-layout checkpoint (sharded safetensors + index), a CLIP directory, a SAM `.pth`, and a
peft-0.4.0-layout LoRA adapter are written to a temp dir; the caller's own sequence of calls is replayed; the
weights that would be handed to `anyref_set_weight` are compared with an independent computation."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

from anyref_amd.config import config_tiny
from anyref_amd.synth import synth_state_dict, CLIP_PREFIX, SAM_PREFIX


def _write_base(tmp, cfg, sd):
    l = cfg.llm
    base = os.path.join(tmp, "LLaVA-tiny")
    os.makedirs(base)
    json.dump(dict(architectures=["LlavaLlamaForCausalLM"], hidden_size=l.dim, intermediate_size=l.mlp,
                   num_hidden_layers=l.layers, num_attention_heads=l.heads, vocab_size=l.vocab, rms_norm_eps=l.rms_eps,
                   bos_token_id=1, eos_token_id=2, pad_token_id=0, mm_vision_tower=os.path.join(tmp, "clip")),
              open(os.path.join(base, "config.json"), "w"))
    llm = {k: v.half() for k, v in sd.items()
           if k.startswith(("model.layers.", "model.embed_tokens", "model.norm", "lm_head", "model.mm_projector"))}
    names = sorted(llm)
    half = len(names) // 2
    shards = {"model-00001-of-00002.safetensors": names[:half], "model-00002-of-00002.safetensors": names[half:]}
    wm = {}
    for f, ks in shards.items():
        save_file({k: llm[k].contiguous() for k in ks}, os.path.join(base, f))
        wm.update({k: f for k in ks})
    json.dump(dict(metadata={}, weight_map=wm), open(os.path.join(base, "model.safetensors.index.json"), "w"))
    return base, llm


def _write_clip(tmp, cfg, sd):
    c = cfg.clip
    d = os.path.join(tmp, "clip")
    os.makedirs(d)
    json.dump(dict(vision_config=dict(hidden_size=c.dim, intermediate_size=c.mlp, num_hidden_layers=c.layers,
                                      num_attention_heads=c.heads, image_size=c.image_size, patch_size=c.patch)),
              open(os.path.join(d, "config.json"), "w"))
    clip = {"vision_model." + k[len(CLIP_PREFIX):]: v for k, v in sd.items() if k.startswith(CLIP_PREFIX)}
    clip["text_model.embeddings.token_embedding.weight"] = torch.zeros(4, 4)      # the text tower is dropped
    torch.save(clip, os.path.join(d, "pytorch_model.bin"))


def _write_adapter(tmp, cfg, sd, g):
    d = os.path.join(tmp, "adapter")
    os.makedirs(d)
    r, alpha = 4, 16
    json.dump(dict(peft_type="LORA", r=r, lora_alpha=alpha, lora_dropout=0.05, bias="none", fan_in_fan_out=False,
                   target_modules=["q_proj", "v_proj"], modules_to_save=["embed_tokens", "lm_head", "text_hidden_fcs"]),
              open(os.path.join(d, "adapter_config.json"), "w"))
    ad, want = {}, {}
    H = cfg.llm.dim
    for i in range(cfg.llm.layers):
        for proj in ("q_proj", "v_proj"):
            A = torch.randn(r, H, generator=g) * 0.1
            Bm = torch.randn(H, r, generator=g) * 0.1
            mod = f"model.layers.{i}.self_attn.{proj}"
            ad[f"base_model.model.{mod}.lora_A.weight"] = A
            ad[f"base_model.model.{mod}.lora_B.weight"] = Bm
            want[mod + ".weight"] = (sd[mod + ".weight"].half().float() + (alpha / r) * (Bm @ A)).half()
    V = cfg.llm.vocab + 7
    for name, shape in (("model.embed_tokens.weight", (V, H)), ("lm_head.weight", (V, H)),
                        ("model.text_hidden_fcs.0.0.weight", (H, H)), ("model.text_hidden_fcs.0.0.bias", (H,)),
                        ("model.text_hidden_fcs.0.2.weight", (cfg.out_dim, H)), ("model.text_hidden_fcs.0.2.bias", (cfg.out_dim,))):
        t = torch.randn(*shape, generator=g) * 0.02
        ad["base_model.model." + name] = t
        want[name] = t
    torch.save(ad, os.path.join(d, "adapter_model.bin"))
    return d, want


def test_callers_construction_path(tmp_path):
    from anyref_amd.model import AnyRefForCausalLM
    from anyref_amd.peft_compat import PeftModel
    tmp = str(tmp_path)
    cfg = config_tiny()
    # SAM ViT-B width so `vision_pretrained` picks a real variant by substring; image kept small via the config
    sd = synth_state_dict(cfg, seed=5, scale=0.05)
    base, llm = _write_base(tmp, cfg, sd)
    _write_clip(tmp, cfg, sd)
    import dataclasses
    from anyref_amd.checkpoint import sam_config_for
    sam_cfg = dataclasses.replace(cfg, sam=sam_config_for("sam_vit_b_01ec64.pth", img_size=64, patch=16, window=14))
    sam_sd = {k[len(SAM_PREFIX):]: v for k, v in synth_state_dict(sam_cfg, seed=6, scale=0.05).items() if k.startswith(SAM_PREFIX)}
    sam_path = os.path.join(tmp, "sam_vit_b_01ec64.pth")
    torch.save(sam_sd, sam_path)
    g = torch.Generator().manual_seed(9)
    adapter, want = _write_adapter(tmp, cfg, sd, g)

    # ---- the caller's lines, eval_referseg.py:62-88 (imports swapped, nothing else) ----
    model_args = {"train_mask_decoder": True, "out_dim": 256, "seg_token_idx": cfg.llm.vocab + 0,
                  "vision_pretrained": sam_path, "add_audio_encoder": False, "rephrase_weight": 0.1}
    model = AnyRefForCausalLM.from_pretrained(base, torch_dtype=torch.float16, **model_args)
    model.config.eos_token_id, model.config.bos_token_id, model.config.pad_token_id = 2, 1, 0
    model.get_model().initialize_vision_modules(model.get_model().config)
    model.get_model().get_vision_tower().to(torch.float16)
    model.get_model().initialize_anyref_modules(model.get_model().config)
    model.resize_token_embeddings(cfg.llm.vocab + 7)
    model = PeftModel.from_pretrained(model, adapter)
    model = model.merge_and_unload()
    model.to(torch.float16)
    model.eval()

    got = model.host_state_dict()
    assert model.adapter_stats == {"lora_pairs": 2 * cfg.llm.layers, "modules_to_save": 6}
    assert model.cfg.llm.vocab == cfg.llm.vocab + 7 and model.cfg.rephrase_weight == 0.1
    assert model.cfg.sam.dim == 768 and model.cfg.sam.depth == 12 and model.cfg.sam.global_idx == (2, 5, 8, 11)
    assert model.cfg.clip.dim == cfg.clip.dim and model.cfg.clip.layers == cfg.clip.layers
    for k, v in want.items():                                  # LoRA-merged and modules_to_save tensors
        assert torch.equal(got[k].float(), v.float()), k
    for k, v in llm.items():                                   # untouched base tensors come through bit for bit
        if k not in want:
            assert torch.equal(got[k], v), k
    for k, v in sd.items():                                    # CLIP by LLaVA's names, SAM under model.visual_model.
        if k.startswith(CLIP_PREFIX):
            assert torch.equal(got[k], v), k
    for k, v in sam_sd.items():
        assert torch.equal(got[SAM_PREFIX + k], v), k
    assert not any(k.startswith("text_model") for k in got)
    from anyref_amd.checkpoint import missing_for
    assert missing_for(model.cfg, got.keys(), audio=False) == []   # every tensor the path reads is there


def test_left_padded_rows_without_masks():
    """eval_referseg.py:124-137 with batch_num > 1: left-padded ids and NO attention_masks."""
    from anyref_amd.model import AnyRefForCausalLM
    m = AnyRefForCausalLM(config_tiny(), defer=True)
    ids = torch.tensor([[0, 0, 0, 1, -200, 5, 6, 7], [1, -200, 9, 8, 7, 6, 5, 4]])
    rows, lens = m._rows(ids, None)
    assert lens.tolist() == [5, 8] and rows[0, :5].tolist() == [1, -200, 5, 6, 7] and rows[1].tolist() == ids[1].tolist()
    mask = ids.ne(0)                                           # the collator's mask gives the same rows
    rows2, lens2 = m._rows(ids, mask)
    assert torch.equal(rows, rows2) and torch.equal(lens, lens2)
    one, l1 = m._rows(ids[:1], None)                           # batch of one: taken as is (eval_referseg.py:124)
    assert l1.tolist() == [8]


def test_adapter_errors(tmp_path):
    from anyref_amd.checkpoint import merge_lora
    d = str(tmp_path)
    json.dump(dict(r=4, lora_alpha=8), open(os.path.join(d, "adapter_config.json"), "w"))
    torch.save({"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": torch.zeros(4, 8)},
               os.path.join(d, "adapter_model.bin"))
    with pytest.raises(ValueError, match="unpaired"):
        merge_lora({}, d)


def test_hf_dir_layouts_and_peft_key_forms(tmp_path):
    """sharded `.bin` with an index, a single safetensors file, a merged checkpoint (merge_lora.py:62) that already
    carries the towers, and the key spellings peft versions write for `modules_to_save` / adapter names"""
    from anyref_amd.checkpoint import read_hf_dir, merge_lora, _strip_peft_key
    d = str(tmp_path / "bin")
    os.makedirs(d)
    a, b = {"x.weight": torch.arange(6.0).reshape(2, 3)}, {"y.weight": torch.ones(4)}
    torch.save(a, os.path.join(d, "pytorch_model-00001-of-00002.bin"))
    torch.save(b, os.path.join(d, "pytorch_model-00002-of-00002.bin"))
    json.dump(dict(weight_map={"x.weight": "pytorch_model-00001-of-00002.bin", "y.weight": "pytorch_model-00002-of-00002.bin"}),
              open(os.path.join(d, "pytorch_model.bin.index.json"), "w"))
    got = read_hf_dir(d)
    assert torch.equal(got["x.weight"], a["x.weight"]) and torch.equal(got["y.weight"], b["y.weight"])
    d2 = str(tmp_path / "st")
    os.makedirs(d2)
    save_file({"z": torch.zeros(3)}, os.path.join(d2, "model.safetensors"))
    assert list(read_hf_dir(d2)) == ["z"]
    with pytest.raises(FileNotFoundError):
        read_hf_dir(str(tmp_path))
    k = _strip_peft_key("base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight")
    assert k == "model.layers.0.self_attn.q_proj.lora_A.weight"
    assert _strip_peft_key("base_model.model.lm_head.modules_to_save.default.weight") == "lm_head.weight"
    assert _strip_peft_key("base_model.model.model.text_hidden_fcs.0.0.weight") == "model.text_hidden_fcs.0.0.weight"
    # fan_in_fan_out adapters store the transpose; original_module copies are ignored
    ad = str(tmp_path / "ad")
    os.makedirs(ad)
    json.dump(dict(r=2, lora_alpha=4, fan_in_fan_out=True), open(os.path.join(ad, "adapter_config.json"), "w"))
    A, Bm = torch.randn(2, 5), torch.randn(3, 2)
    torch.save({"base_model.model.m.lora_A.default.weight": A, "base_model.model.m.lora_B.default.weight": Bm,
                "base_model.model.head.original_module.weight": torch.zeros(1),
                "base_model.model.head.modules_to_save.default.weight": torch.full((2,), 7.0)},
               os.path.join(ad, "adapter_model.bin"))
    sd = {"m.weight": torch.zeros(5, 3), "head.weight": torch.zeros(2)}
    stats = merge_lora(sd, ad)
    assert stats == {"lora_pairs": 1, "modules_to_save": 1}
    assert torch.allclose(sd["m.weight"], ((Bm @ A) * 2.0).t()) and sd["head.weight"].tolist() == [7.0, 7.0]


def test_merged_checkpoint_needs_no_initialize(tmp_path):
    """`merge_lora.py:62` saves the merged model with `save_pretrained`: its directory holds every tensor under the
    reference's names and a config that already has `train_mask_decoder`; `from_pretrained` alone is complete."""
    from anyref_amd.checkpoint import missing_for
    from anyref_amd.model import AnyRefForCausalLM
    cfg = config_tiny()
    sd = synth_state_dict(cfg, seed=5, scale=0.05)
    d = str(tmp_path / "merged")
    os.makedirs(d)
    l = cfg.llm
    json.dump(dict(hidden_size=l.dim, intermediate_size=l.mlp, num_hidden_layers=l.layers, num_attention_heads=l.heads,
                   vocab_size=l.vocab, rms_norm_eps=l.rms_eps, train_mask_decoder=True, out_dim=256),
              open(os.path.join(d, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "model.safetensors"))
    m = AnyRefForCausalLM.from_pretrained(d, torch_dtype=torch.float16, seg_token_idx=999, vision_pretrained="x/sam_vit_h.pth")
    assert m.cfg.sam.dim == 1280 and m.cfg.sam.depth == 32          # sized from the name, weights from the checkpoint
    m.get_model().initialize_vision_modules(m.get_model().config)    # both no-ops on a merged checkpoint
    got = m.host_state_dict()
    assert all(torch.equal(got[k], v) for k, v in sd.items())
    tiny = config_tiny()
    assert missing_for(tiny, got.keys(), audio=True) == []
