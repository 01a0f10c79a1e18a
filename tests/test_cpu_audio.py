"""ImageBind audio trunk restatement: shape/contract checks and an independent functional
re-computation (parity w.r.t. the reference itself is unpinned: its module needs timm)."""
import torch

from anyref_amd.audio import ImageBindAudio


def test_audio_feature_contract_and_math():
    torch.manual_seed(0)
    m = ImageBindAudio(dim=64, blocks=2, heads=4, out_dim=32).eval()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.05)
    keys = set(m.state_dict().keys())
    for k in ("modality_preprocessors.audio.cls_token", "modality_preprocessors.audio.rgbt_stem.proj.weight",
              "modality_preprocessors.audio.rgbt_stem.norm_layer.weight",
              "modality_preprocessors.audio.pos_embedding_helper.pos_embed",
              "modality_trunks.audio.blocks.0.attn.in_proj_weight", "modality_trunks.audio.blocks.0.attn.bias_k",
              "modality_trunks.audio.blocks.1.mlp.fc2.bias", "modality_heads.audio.0.weight",
              "modality_heads.audio.2.weight", "modality_postprocessors.audio.1.log_logit_scale"):
        assert k in keys, k
    x = torch.randn(1, 3, 1, 128, 204)
    feat, emb = m.get_audio_feature(x)
    assert feat.shape == (1, 3, 64) and emb.shape == (1, 3, 32)
    assert torch.allclose(emb.norm(dim=-1), torch.full((1, 3), 20.0), atol=1e-4)     # L2-normalised x 20
    # independent recomputation of one block's attention with explicit bias_k / bias_v rows
    blk = m.modality_trunks["audio"].blocks[0]
    t = torch.randn(5, 2, 64)
    h = blk.norm_1(t)
    W, b = blk.attn.in_proj_weight, blk.attn.in_proj_bias
    q, k, v = (h @ W[i * 64:(i + 1) * 64].t() + b[i * 64:(i + 1) * 64] for i in range(3))
    k = torch.cat([k, blk.attn.bias_k.expand(1, 2, 64)], 0)
    v = torch.cat([v, blk.attn.bias_v.expand(1, 2, 64)], 0)
    sh = lambda z: z.reshape(z.shape[0], 2, 4, 16).permute(1, 2, 0, 3)
    a = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / 4.0, -1) @ sh(v)
    a = a.permute(2, 0, 1, 3).reshape(5, 2, 64) @ blk.attn.out_proj.weight.t() + blk.attn.out_proj.bias
    ref = t + a
    ref = ref + blk.mlp.fc2(torch.nn.functional.gelu(blk.mlp.fc1(blk.norm_2(ref))))
    assert torch.allclose(blk(t), ref, atol=1e-5)


def test_audio_trunk_against_reference_fixture():
    """The restatement at ImageBind's real audio size against outputs of the reference's own
    `ImageBindModel.get_audio_feature` (imagebind_model.py:477-511) run in the build container
    (tests/golden/make_golden_audio.py)."""
    import os
    import sys
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden_audio as ga
    fx = np.load(os.path.join(here, "golden", "imagebind_audio.npz"))
    m = ga.seeded_audio_module()
    wsum = sum(float(v.double().abs().sum()) for v in m.state_dict().values())
    assert abs(wsum - float(fx["wsum"])) < 1e-6 * float(fx["wsum"]), "seeded weights drifted"
    feat, emb = m.get_audio_feature(ga.audio_inputs())
    assert np.abs(feat.numpy() - fx["feat"]).max() < 2e-5 * max(1.0, np.abs(fx["feat"]).max())
    assert np.abs(emb.numpy() - fx["emb"]).max() < 2e-5 * np.abs(fx["emb"]).max()
