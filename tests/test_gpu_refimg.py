"""the reference-image (`ref_images`) branch of generate / forward -- a second CLIP pass per
reference image, 256 -> 16 -> IMG_REF_NUM token pooling (model/anyref.py:319-339, :681-702), 1:1 replacement of the
prompt's `<img_ref>` placeholders -- through the C-ABI against the oracle restatement, tensor and list forms."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from anyref_amd.config import config_tiny, IMAGE_TOKEN_INDEX, IMG_REF_INDEX, IMG_REF_NUM  # noqa: E402
from anyref_amd.synth import synth_state_dict  # noqa: E402
from oracle import anyref_oracle as O  # noqa: E402


def _setup(B, seed):
    cfg = config_tiny()
    sd = synth_state_dict(cfg, seed=seed, scale=0.05)
    g = torch.Generator().manual_seed(seed + 1)
    clip = torch.randn(B, 3, 224, 224, generator=g)
    sam = torch.randn(B, 3, 224, 224, generator=g)
    refs = torch.randn(B, 3, 224, 224, generator=g)
    ids = []
    for b in range(B):
        body = torch.randint(3, 980, (10 - b,), generator=g)
        ids.append(torch.cat([torch.tensor([1, IMAGE_TOKEN_INDEX]), body[:4], torch.full((IMG_REF_NUM,), IMG_REF_INDEX), body[4:]]))
    return cfg, sd, clip, sam, refs, ids


def test_pool_kernel_matches_reference_expression():
    """anyref.py:335-338 / :697-700 as torch writes it, and the 16-row case that needs no second mean"""
    import ctypes as C
    from anyref_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for L, n_out in ((256, 4), (64, 4), (256, 16)):
        f = torch.randn(3, L, 96, generator=g).cuda()
        out = torch.empty(3, n_out, 96, device="cuda")
        rc = lib.anyref_op_pool_ref_tokens(None, C.c_void_p(f.data_ptr()), 3, L, 96, n_out, C.c_void_p(out.data_ptr()))
        assert rc == 0, lib.anyref_op_last_error()
        torch.cuda.synchronize()
        want = O.pool_ref_tokens(f.cpu(), n_out)
        assert (out.cpu() - want).abs().max().item() < 1e-6
    bad = torch.zeros(1, 100, 8).cuda()
    assert lib.anyref_op_pool_ref_tokens(None, C.c_void_p(bad.data_ptr()), 1, 100, 8, 4, C.c_void_p(bad.data_ptr())) != 0


@pytest.mark.parametrize("form", ["tensor", "list", "list_with_none"])
def test_generate_with_reference_images(form):
    from anyref_amd.model import AnyRefForCausalLM
    B = 2
    cfg, sd, clip, sam, refs, ids = _setup(B, seed=31)
    sizes, H, W = [(224, 224), (224, 200)], [224, 150], [224, 170]
    if form == "list_with_none":                      # row 1 has no reference image: no placeholders in its prompt
        ids[1] = ids[1][ids[1] != IMG_REF_INDEX]
    enc = lambda x: O.encode_images(sd, cfg, x)
    if form == "tensor":
        ref_arg = refs
        feats = list(O.ref_features_generate(enc, refs, B))                 # pooled [B, 4, H] (:695-700)
    else:
        ref_arg = [refs[0], refs[1] if form == "list" else None]
        feats = O.ref_features_generate(enc, ref_arg, B)                    # UNPOOLED [256, H] items (:691-692) ...
        assert feats[0].shape[0] == 256
    with torch.no_grad():                                                   # ... which the splice pools (documented reading)
        r0 = O.anyref_generate(sd, cfg, clip, ids, sam, sizes, H, W, ref_feats=feats, max_new_tokens=4, eos=False)
        cfg.seg_token_idx = int(r0["output_ids"][0][-2])
        ref = O.anyref_generate(sd, cfg, clip, ids, sam, sizes, H, W, ref_feats=feats, max_new_tokens=5, eos=False)
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity", max_batch=B, max_seg=4)
    m.config.eos_token_id = None
    L = max(len(r) for r in ids)
    padded = torch.zeros(B, L, dtype=torch.long)
    mask = torch.zeros(B, L, dtype=torch.bool)
    for b, r in enumerate(ids):
        padded[b, : len(r)] = r
        mask[b, : len(r)] = True
    (out_ids, masks, _), ex = m.generate(clip, padded, sam, sizes, H, W, ref_images=ref_arg, max_new_tokens=5,
                                         attention_masks=mask, _return_extras=True)
    assert ref["pred_masks"] is not None
    for b in range(B):
        want = ref["output_ids"][b]
        assert out_ids[b, : len(want)].cpu().tolist() == want.tolist(), f"row {b}: ids differ"
        n = ref["hidden"][b].shape[0]
        assert (ex["hidden"][b, :n].cpu() - ref["hidden"][b]).abs().max().item() < 2e-4
        if ref["pred_masks"][b].shape[0]:
            assert (masks[b].cpu() - ref["pred_masks"][b]).abs().max().item() <= 1e-3
    # the reference features really entered the sequence: a different reference image changes the hidden states
    other = refs.flip(0) if form == "tensor" else [refs[1], ref_arg[1]]
    (_, _, _), ex2 = m.generate(clip, padded, sam, sizes, H, W, ref_images=other, max_new_tokens=5, attention_masks=mask,
                                _return_extras=True)
    assert (ex2["hidden"][0, :n] - ex["hidden"][0, :n]).abs().max().item() > 1e-3


def test_reference_image_refusals():
    from anyref_amd.model import AnyRefForCausalLM
    cfg, sd, clip, sam, refs, ids = _setup(1, seed=33)
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity", max_batch=1)
    m.config.eos_token_id = None
    args = (clip, ids[0][None], sam, [(224, 224)], [224], [224])
    with pytest.raises(NotImplementedError):
        m.generate(*args, ref_images=[torch.tensor([0.1, 0.2, 0.6, 0.7])], max_new_tokens=2)     # RoI coordinates (:688-689)
    with pytest.raises(NotImplementedError):
        m.generate(*args, ref_images=refs[0], max_new_tokens=2)                                  # 3-D tensor (:701-702)
    with pytest.raises(ValueError, match="placeholders"):
        m.generate(clip, ids[0][ids[0] != IMG_REF_INDEX][None], sam, [(224, 224)], [224], [224], ref_images=refs,
                   max_new_tokens=2)
