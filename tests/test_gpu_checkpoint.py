"""The callers' construction path (eval_referseg.py:62-88) END TO END on the GPU: a tiny HF-layout checkpoint, CLIP
directory, SAM `.pth` and peft-layout LoRA adapter on disk -> `from_pretrained` -> `initialize_*` ->
`resize_token_embeddings` -> `PeftModel.from_pretrained(...).merge_and_unload()` -> `.cuda()` -> `generate`, against
the CPU oracle run on the independently merged weights."""
import dataclasses
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from anyref_amd.config import config_tiny, IMAGE_TOKEN_INDEX  # noqa: E402
from anyref_amd.synth import synth_state_dict, SAM_PREFIX  # noqa: E402
from oracle import anyref_oracle as O  # noqa: E402
from test_cpu_checkpoint import _write_base, _write_clip, _write_adapter  # noqa: E402


def test_from_pretrained_to_generate(tmp_path):
    from anyref_amd.checkpoint import sam_config_for
    from anyref_amd.model import AnyRefForCausalLM
    from anyref_amd.peft_compat import PeftModel
    tmp = str(tmp_path)
    cfg = config_tiny()
    sd = synth_state_dict(cfg, seed=5, scale=0.05)
    base, _ = _write_base(tmp, cfg, sd)
    _write_clip(tmp, cfg, sd)
    # a SAM ViT-B-shaped checkpoint at a 64 x 64 input (4 x 4 tokens) so that the variant-by-substring logic is exercised
    # and the CPU oracle stays cheap; window 14 > grid 4 means every block pads its single window 4 -> 14
    sam_shape = sam_config_for("sam_vit_b_01ec64.pth", img_size=64, patch=16, window=14)
    sam_cfg = dataclasses.replace(cfg, sam=sam_shape)
    sam_sd = {k[len(SAM_PREFIX):]: v for k, v in synth_state_dict(sam_cfg, seed=6, scale=0.05).items() if k.startswith(SAM_PREFIX)}
    sam_path = os.path.join(tmp, "sam_vit_b_01ec64.pth")
    torch.save(sam_sd, sam_path)
    adapter, want = _write_adapter(tmp, cfg, sd, torch.Generator().manual_seed(9))

    model = AnyRefForCausalLM.from_pretrained(base, torch_dtype=torch.float16, mode="parity", max_seg=4, max_seq=512,
                                              train_mask_decoder=True, out_dim=256, seg_token_idx=cfg.llm.vocab,
                                              vision_pretrained=sam_path, add_audio_encoder=False, rephrase_weight=0.0)
    model.cfg.sam = dataclasses.replace(model.cfg.sam, img_size=64)      # (the reference hard-codes 1024; tiny here)
    model.config.eos_token_id, model.config.bos_token_id, model.config.pad_token_id = None, 1, 0
    model.get_model().initialize_vision_modules(model.get_model().config)
    model.get_model().get_vision_tower().to(torch.float16)
    model.get_model().initialize_anyref_modules(model.get_model().config)
    model.cfg.sam = dataclasses.replace(model.cfg.sam, img_size=64)
    model.resize_token_embeddings(cfg.llm.vocab + 7)
    model = PeftModel.from_pretrained(model, adapter).merge_and_unload()
    model.to(torch.float16)
    model.eval()
    # the oracle's weights: what the host state dict holds before the build (merged, fp16-rounded base tensors)
    w = {k: v.float() for k, v in model.host_state_dict().items()}
    ocfg = dataclasses.replace(model.cfg)
    model = model.cuda()                                                  # builds the handle, weights to HBM
    assert model.host_state_dict() is None and model.device_bytes > 0

    g = torch.Generator().manual_seed(11)
    clip = torch.randn(1, 3, 224, 224, generator=g)
    sam = torch.randn(1, 3, 64, 64, generator=g)
    ids = torch.cat([torch.tensor([1, IMAGE_TOKEN_INDEX]), torch.randint(3, 990, (12,), generator=g)])
    sizes, H, W = [(64, 48)], [90], [70]
    with torch.no_grad():
        r0 = O.anyref_generate(w, ocfg, clip, [ids], sam, sizes, H, W, max_new_tokens=4, eos=False)
        seg = int(r0["output_ids"][0][-2])
        ocfg.seg_token_idx = seg
        ref = O.anyref_generate(w, ocfg, clip, [ids], sam, sizes, H, W, max_new_tokens=5, eos=False)
    model.set_seg_token_idx(seg)
    out_ids, masks, rest = model.generate(clip, ids[None], sam, sizes, H, W, max_new_tokens=5)
    assert rest == (None, None, None)
    assert out_ids[0].cpu().tolist() == ref["output_ids"][0].tolist()
    assert ref["pred_masks"] is not None and masks[0].shape == ref["pred_masks"][0].shape
    assert (masks[0].cpu() - ref["pred_masks"][0]).abs().max().item() <= 1e-3
    with pytest.raises(RuntimeError, match="before .cuda"):
        model.merge_adapter(adapter)
