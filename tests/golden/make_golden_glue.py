"""Golden vectors that pin the GLUE of the oracle (`oracle/anyref_oracle.py::generate_tail`,
`forward_tail`, `ref_features_*`, the losses) to the reference's own code.

Run ONLY in the build container (needs `/root/reference`, which never travels):

    python tests/golden/make_golden_glue.py

What runs here is the reference's `model/anyref.py` itself -- `AnyRefForCausalLM.generate`
(:647-822), `model_forward_new` (:239-466), `initialize_anyref_modules` (:96-161), `dice_loss` /
`sigmoid_ce_loss` (:19-68) -- imported from `/root/reference` and executed line by line.  Its parent
class lives in `model/llava/**`, which the reference git-ignores and does not ship,
and its imports of ImageBind / wandb / torchvision need packages this image lacks.  Those are replaced
by the stand-ins below, whose ONLY behaviour is to hand back CANNED tensors:

  * `LlavaLlamaForCausalLM.generate / forward`  -> canned `sequences`, `hidden_states[-1]`,
    `attentions[-1]`, `loss`, and a record of the kwargs the glue passed down;
  * `encode_images`                              -> a fixed seeded linear map of the 14x14 patches;
  * `audio_encoder.get_audio_feature`            -> a canned [1,3,1024] tensor;
  * `build_sam_vit_b`                            -> a SMALL `Sam` built from the reference's own
    `segment_anything/modeling` classes (the real builders hard-code 1024^2 ViT-B/L/H).

So these fixtures pin the glue (the [SEG] index arithmetic with its +255 / -1+255 offsets, rephrase,
`text_hidden_fcs`, per-sample prompt-encoder -> mask-decoder -> postprocess loop, return conventions,
what is handed to the llava layer for audio / reference images, BCE + Dice) -- NOT the LLM / CLIP
arithmetic, which stays pinned to the HF stand-in only (`llm_clip_hf.npz`).

No reference source is stored: inputs are regenerated from seeds by `case_inputs`, outputs are arrays.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from anyref_amd.config import AnyRefConfig, ClipConfig, LlmConfig, SamConfig  # noqa: E402
from anyref_amd.synth import synth_state_dict, SAM_PREFIX  # noqa: E402

REF = "/root/reference"
SEED = 31
H_LLM, HEADS, VOCAB = 64, 2, 300
SEG, SEG_LIST = 290, [291, 292, 293]
IMG_REF_NUM = 4


def glue_cfg():
    return AnyRefConfig(
        clip=ClipConfig(image_size=224, patch=14, dim=32, heads=2, layers=2, mlp=64),
        llm=LlmConfig(vocab=VOCAB, dim=H_LLM, heads=HEADS, layers=1, mlp=96, max_seq=512),
        sam=SamConfig(img_size=224, patch=16, dim=128, depth=2, heads=2, window=14, global_idx=(1,)),
        seg_token_idx=SEG)


# ---------------------------------------------------------------------------------------------
# cases: (name, dict).  ids are int lists; `seg_at` = positions (in the full sequence) that hold a [SEG] id.
# ---------------------------------------------------------------------------------------------
GEN_CASES = {
    "g_one_seg": dict(bs=1, L=12, T=6, segs=[[15]], rephrase=0.0, sizes=[(224, 224)], hw=[(224, 224)]),
    "g_two_seg_crop": dict(bs=1, L=10, T=8, segs=[[12, 16]], rephrase=0.0, sizes=[(150, 224)], hw=[(301, 437)]),
    "g_rephrase_two_seg": dict(bs=1, L=11, T=9, segs=[[15, 18]], rephrase=0.5, sizes=[(224, 200)], hw=[(180, 160)]),
    "g_no_seg": dict(bs=1, L=9, T=5, segs=[[]], rephrase=0.0, sizes=[(224, 224)], hw=[(224, 224)]),
    "g_seg_list": dict(bs=1, L=9, T=7, segs=[[11, 14]], rephrase=0.0, sizes=[(224, 224)], hw=[(100, 120)],
                       seg_list=True),
    "g_batch2_rephrase": dict(bs=2, L=10, T=6, segs=[[13], [14]], rephrase=0.3, sizes=[(224, 224), (200, 224)],
                              hw=[(224, 224), (120, 333)]),
    "g_batch2_row_without_seg": dict(bs=2, L=10, T=6, segs=[[12, 14], []], rephrase=0.0,
                                     sizes=[(224, 224), (224, 224)], hw=[(224, 224), (224, 224)]),
}
FWD_CASES = {
    "f_one_seg": dict(bs=1, L=20, segs=[[16]], ans=[12], rephrase=0.0, sizes=[(224, 224)], hw=[(224, 224)],
                      gt_hw=[(224, 224)]),
    "f_batch2_rephrase_avs": dict(bs=2, L=22, segs=[[17], [19]], ans=[11, 13], rephrase=0.4,
                                  sizes=[(224, 224), (180, 224)], hw=[(224, 224), (200, 260)],
                                  gt_hw=[(112, 112), (200, 260)]),
    "f_two_seg": dict(bs=1, L=24, segs=[[15, 21]], ans=[10], rephrase=0.0, sizes=[(224, 224)], hw=[(150, 170)],
                      gt_hw=[(150, 170)]),
    "f_no_seg": dict(bs=1, L=14, segs=[[]], ans=[8], rephrase=0.0, sizes=[(224, 224)], hw=[(224, 224)],
                     gt_hw=[(224, 224)]),
}


def encode_stub_weight():
    g = torch.Generator().manual_seed(SEED + 5)
    return torch.randn(3 * 14 * 14, H_LLM, generator=g) * 0.05


def encode_stub(x: torch.Tensor) -> torch.Tensor:
    """Stand-in for the absent `encode_images`: [n,3,224,224] -> [n,256,H] (a fixed linear map of each patch)."""
    n = x.shape[0]
    p = x.float().unfold(2, 14, 14).unfold(3, 14, 14)                  # [n,3,16,16,14,14]
    p = p.permute(0, 2, 3, 1, 4, 5).reshape(n, 256, 3 * 14 * 14)
    return p @ encode_stub_weight()


def case_inputs(name: str):
    """Seeded canned LLM outputs + model inputs of one case (shared by the generator and the test)."""
    gen = name in GEN_CASES
    c = (GEN_CASES if gen else FWD_CASES)[name]
    g = torch.Generator().manual_seed(SEED + sum(map(ord, name)))
    bs, L = c["bs"], c["L"]
    n = L + c["T"] if gen else L
    seq = torch.randint(3, 280, (bs, n), generator=g)
    seq[:, 0] = 1
    for b, pos in enumerate(c["segs"]):
        for j, p in enumerate(pos):
            seq[b, p] = (SEG_LIST[j % len(SEG_LIST)] if c.get("seg_list") else SEG)
    S = (n - 1 if gen else n) + 255                                   # rows of hidden_states[-1]
    hidden = torch.randn(bs, S, H_LLM, generator=g)
    attn = torch.rand(bs, HEADS, S, S, generator=g).tril()
    attn = attn / attn.sum(-1, keepdim=True)
    sam = torch.randn(bs, 3, 224, 224, generator=g)
    clip = torch.randn(bs, 3, 224, 224, generator=g)
    out = dict(c=c, seq=seq, hidden=hidden, attn=attn, sam=sam, clip=clip)
    if not gen:
        labels = seq.clone()
        for b in range(bs):
            labels[b, : c["ans"][b]] = -100
        out["labels"] = labels
        out["lm_loss"] = torch.tensor(1.2345)
        out["gt"] = [(torch.rand(len(c["segs"][b]), *c["gt_hw"][b], generator=g) > 0.5).float() for b in range(bs)]
    return out


def handdown_inputs():
    """Inputs of the audio / reference-image hand-down cases."""
    g = torch.Generator().manual_seed(SEED + 77)
    return dict(ref_a=torch.randn(3, 224, 224, generator=g), ref_b=torch.randn(3, 224, 224, generator=g),
                roi=torch.tensor([0.1, 0.2, 0.6, 0.7]), audio_emb=torch.randn(1, 3, 1024, generator=g),
                audio_emb2=torch.randn(1, 3, 1024, generator=g))


# ---------------------------------------------------------------------------------------------
# everything below touches /root/reference and runs only in the build container
# ---------------------------------------------------------------------------------------------
class _Recorder:
    audio_next = []


def _install_standins():
    def pkg(name, path=None):
        m = types.ModuleType(name)
        m.__path__ = [path] if path else []
        sys.modules[name] = m
        return m

    pkg("model", REF + "/model")
    # segment_anything: the REAL sub-package directory, minus its __init__ (which imports torchvision)
    sa = pkg("model.segment_anything", REF + "/model/segment_anything")
    import importlib
    bsam = importlib.import_module("model.segment_anything.build_sam")
    for n in ("build_sam_vit_h", "build_sam_vit_l", "build_sam_vit_b"):
        setattr(sa, n, getattr(bsam, n))

    class LlavaLlamaModel(nn.Module):
        def __init__(self, config):
            super().__init__()

    class LlavaLlamaForCausalLM(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config

        def post_init(self):
            pass

        def generate(self, **kw):
            self.passed_down = kw
            return self.canned

        def forward(self, **kw):
            self.passed_down = kw
            return self.canned

        def encode_images(self, x):
            return encode_stub(x)

    pkg("model.llava"); pkg("model.llava.model"); pkg("model.llava.model.language_model")
    ll = pkg("model.llava.model.language_model.llava_llama")
    ll.LlavaLlamaForCausalLM, ll.LlavaLlamaModel = LlavaLlamaForCausalLM, LlavaLlamaModel
    pkg("model.llava.constants").IMG_REF_NUM = IMG_REF_NUM

    class AudioStub(nn.Module):
        def __init__(self):
            super().__init__()
            names = ["vision", "text", "depth", "thermal", "imu", "audio"]
            for a in ("modality_preprocessors", "modality_trunks", "modality_postprocessors", "modality_heads"):
                setattr(self, a, nn.ModuleDict({n: nn.Identity() for n in names}))

        def get_audio_feature(self, audio, modality):
            assert modality == "audio"
            return None, _Recorder.audio_next.pop(0)

    pkg("model.ImageBind")
    ibm = pkg("model.ImageBind.models")
    ib = pkg("model.ImageBind.models.imagebind_model")
    ib.ModalityType = types.SimpleNamespace(AUDIO="audio")
    ib.imagebind_huge = lambda: (AudioStub(), 1024)
    ibm.imagebind_model = ib
    wb = types.ModuleType("wandb")
    wb.run = None
    sys.modules["wandb"] = wb


def _small_sam(cfg):
    from functools import partial
    import importlib
    ref = importlib.import_module("model.segment_anything.modeling")
    s = cfg.sam
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
    return ref.Sam(enc, pe, dec)


def build_reference_model(cfg, sd, rephrase, seg_idx):
    import importlib
    anyref = importlib.import_module("model.anyref")
    anyref.build_sam_vit_b = lambda ckpt: _small_sam(cfg)          # the real builder hard-codes 1024^2 ViT-B
    hf_cfg = types.SimpleNamespace(hidden_size=cfg.llm.dim, vocab_size=cfg.llm.vocab)
    m = anyref.AnyRefForCausalLM(hf_cfg, train_mask_decoder=True, out_dim=cfg.out_dim, seg_token_idx=seg_idx,
                                 vision_pretrained="tiny_sam_vit_b", add_audio_encoder=True,
                                 imagebind_ckpt="/nonexistent/imagebind.pth", rephrase_weight=rephrase)
    # `model_forward_new` reads `self.loc_token_idx` (:285,:403) but the shipped `__init__` only has it commented
    # out (:202-207); with the attribute absent the forward raises, so the (absent) parent class or the training
    # script must define it.  None = "no [LOC] tokens", the only setting whose code path is complete.
    m.loc_token_idx = None
    m.get_model = lambda: m.model
    m.model.initialize_anyref_modules(m.model.config)              # anyref.py:96-161 (eval_referseg.py:81)
    sub = {k: v for k, v in sd.items()
           if k.startswith(("model.visual_model.", "model.text_hidden_fcs.", "model.audio_projector.", "lm_head."))}
    missing, unexpected = m.load_state_dict(sub, strict=False)
    assert not unexpected, unexpected
    for k in missing:
        assert k.startswith(("model.visual_model.prompt_encoder.point_embeddings", "model.visual_model.prompt_encoder.not_a_point",
                             "model.visual_model.prompt_encoder.mask_downscaling", "model.audio_encoder.")), k
    return m.eval()


def main():
    _install_standins()
    cfg = glue_cfg()
    sd = synth_state_dict(cfg, seed=SEED, scale=0.05)
    out = {}
    ns = types.SimpleNamespace

    for name, c in GEN_CASES.items():
        x = case_inputs(name)
        m = build_reference_model(cfg, sd, c["rephrase"], SEG_LIST if c.get("seg_list") else SEG)
        m.canned = ns(sequences=x["seq"], hidden_states=(x["hidden"],), attentions=(x["attn"],))
        r = m.generate(x["clip"], x["seq"][:, : c["L"]], x["sam"], c["sizes"], [h for h, _ in c["hw"]],
                       [w for _, w in c["hw"]], max_new_tokens=c["T"])
        out[name + ".arity"] = np.int64(len(r))
        out[name + ".masks_none"] = np.int64(r[1] is None)
        assert torch.equal(r[0], x["seq"])
        assert m.passed_down["output_attentions"] == (c["rephrase"] > 0)
        if r[1] is not None:
            for b, pm in enumerate(r[1]):
                out[f"{name}.mask{b}"] = pm.numpy()[:, ::3, ::3]
                out[f"{name}.shape{b}"] = np.array(pm.shape)
        print(name, "arity", len(r), None if r[1] is None else [tuple(t.shape) for t in r[1]])

    for name, c in FWD_CASES.items():
        x = case_inputs(name)
        m = build_reference_model(cfg, sd, c["rephrase"], SEG)
        m.canned = ns(loss=x["lm_loss"], hidden_states=(x["hidden"],), attentions=(x["attn"],))
        bs = c["bs"]
        r = m.model_forward_new(x["clip"], x["sam"], x["seq"], x["labels"], torch.ones_like(x["seq"]).bool(),
                                c["sizes"], x["gt"], [h for h, _ in c["hw"]], [w for _, w in c["hw"]],
                                audios=[None] * bs, ref_images=[None] * bs)
        out[name + ".keys"] = np.array(sorted(r.keys()))
        for k, v in r.items():
            out[f"{name}.{k}"] = np.float64(float(v))
        print(name, {k: round(float(v), 6) for k, v in r.items()})

    # what the glue hands down to the llava layer for audio / reference images
    hd = handdown_inputs()
    x = case_inputs("g_no_seg")
    m = build_reference_model(cfg, sd, 0.0, SEG)
    m.canned = ns(sequences=x["seq"], hidden_states=(x["hidden"],), attentions=None)
    base = (x["clip"], x["seq"][:, :9], x["sam"], [(224, 224)], [224], [224])
    _Recorder.audio_next = [hd["audio_emb"]]
    m.generate(*base, audios=[torch.zeros(1, 3, 1, 128, 204)], ref_images=[hd["ref_a"]])
    out["hand.gen_list.audio0"] = m.passed_down["audios"][0].detach().numpy()
    out["hand.gen_list.ref0"] = m.passed_down["ref_images"][0].detach().numpy()
    m.generate(*base, audios=[None], ref_images=[hd["roi"]])
    assert m.passed_down["audios"][0] is None
    out["hand.gen_list.roi0"] = m.passed_down["ref_images"][0].numpy()
    _Recorder.audio_next = [hd["audio_emb"]]
    m.generate(*base, audios=torch.zeros(1, 3, 1, 128, 204), ref_images=hd["ref_a"][None])
    out["hand.gen_tensor.audio"] = m.passed_down["audios"].detach().numpy()
    out["hand.gen_tensor.ref"] = m.passed_down["ref_images"].detach().numpy()
    xf = case_inputs("f_no_seg")
    m.canned = ns(loss=xf["lm_loss"], hidden_states=(xf["hidden"],), attentions=(xf["attn"],))
    _Recorder.audio_next = [hd["audio_emb2"]]
    m.model_forward_new(xf["clip"], xf["sam"], xf["seq"], xf["labels"], None, [(224, 224)], xf["gt"], [224], [224],
                        audios=[torch.zeros(1, 3, 1, 128, 204)], ref_images=[hd["ref_b"]])
    out["hand.fwd_list.audio0"] = m.passed_down["audios"][0].detach().numpy()
    out["hand.fwd_list.ref0"] = m.passed_down["ref_images"][0].detach().numpy()
    for k in ("hand.gen_list.ref0", "hand.gen_tensor.ref", "hand.fwd_list.ref0"):
        print(k, out[k].shape)

    # the two mask losses on their own (anyref.py:19-68)
    import importlib
    anyref = importlib.import_module("model.anyref")
    g = torch.Generator().manual_seed(SEED + 9)
    lg = torch.randn(3, 40, 50, generator=g) * 3
    tg = (torch.rand(3, 40, 50, generator=g) > 0.6).float()
    out["loss.dice"] = np.float64(float(anyref.dice_loss(lg, tg, 3)))
    out["loss.bce"] = np.float64(float(anyref.sigmoid_ce_loss(lg, tg, 3)))
    np.savez_compressed(os.path.join(HERE, "glue_anyref.npz"), **out)
    print("wrote glue_anyref.npz with", len(out), "arrays")


if __name__ == "__main__":
    with torch.no_grad():
        main()
