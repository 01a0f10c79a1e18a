"""ANYREF_MODE_PARITY16 (Python mode="parity16"): the tolerance-meeting arithmetic at 16-bit MFMA rate.

Weights stay in their exact bf16 storage; every activation that feeds a matrix product is f32 carried as a pair of bf16
terms (hi + lo, |a - hi - lo| <= 2^-18 |a|), one MFMA pass per term; decode GEMVs multiply the f32 row with the bf16 weights.
north_star's bar applies to it as to the pure-f32 `parity` mode: identical greedy ids, mask logits within 1e-3
(model/anyref.py:704-716,793-819 against the CPU fp32 forward).

Kernel level (through include/anyref_hip_ops.h, t = 3): against float64 torch on the SAME bf16 weights; the bound is a
small multiple of the pair's 2^-18 -- a single-term bf16 activation would miss it by two orders of magnitude.
End to end: the tiny plumbing config (its 688-wide MLP is not a multiple of the 64-column pair blocks: padded rows),
two LLaMA-7B-wide layers, teacher-forced forward, ragged batches, launch modes bit-identical."""
import ctypes as C
import dataclasses
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from anyref_amd.config import config_tiny, LlmConfig  # noqa: E402
from anyref_amd.synth import synth_state_dict  # noqa: E402
from oracle import anyref_oracle as O  # noqa: E402
from oracle.check import compare_generate  # noqa: E402
from test_gpu_e2e import make_inputs, pad, rig_seg  # noqa: E402

SP = 3              # storage type id of the split-pair arithmetic (include/anyref_hip_ops.h)
MASK_TOL = 1e-3     # north_star
PAIR_REL = 3e-5     # kernel-level bound relative to sum |a||w| / sqrt(K)-ish scale: ~8 x 2^-18


@pytest.fixture(scope="module")
def lib():
    from anyref_amd import _lib
    return _lib.load()


_KEEP = []


def P(t):
    if t is None:
        return None
    _KEEP.append(t)
    return C.c_void_p(t.data_ptr())


def check(lib, rc):
    assert rc == 0, lib.anyref_op_last_error().decode()
    torch.cuda.synchronize()
    _KEEP.clear()


def bf(t):
    return t.to(torch.bfloat16)


def close(got, ref64, tol, what=""):
    got, ref64 = got.double().cpu(), ref64.double().cpu()
    err = (got - ref64).abs().max().item()
    scale = max(1.0, ref64.abs().max().item())
    assert math.isfinite(err) and err <= tol * scale, f"{what} max abs err {err:.3e} vs scale {scale:.3e} (tol {tol})"
    return err / scale


@pytest.mark.parametrize("M,N,K,act,c_f32", [
    (4096, 3840, 1280, 0, 1),     # SAM qkv: 256^2 tile
    (4096, 5120, 1280, 2, 0),     # SAM fc1: 256 x 320 tile, GELU, pair-typed output (the next GEMM's A operand)
    (4096, 1280, 5120, 0, 1),     # SAM fc2: 128 x 160 tile
    (320, 12288, 4096, 0, 1),     # prefill qkv: 64 x 256 tile
    (320, 22016, 4096, 0, 1),     # prefill gate / up: whole-M 320 x 96 tile
    (320, 4096, 4096, 0, 1),      # prefill o_proj: split-K slabs
    (320, 4096, 11008, 0, 1),     # prefill down_proj: split-K slabs, K not a power of two
    (257, 4096, 1024, 3, 0),      # CLIP fc1: quick-GELU, pair-typed output
    (257, 1024, 4096, 0, 1),      # CLIP fc2: split-K
    (6, 256, 256, 1, 1), (70, 130, 64, 4, 0), (1000, 64, 192, 0, 0)])
def test_split_pair_gemm_vs_float64(lib, M, N, K, act, c_f32):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + act)
    A = torch.randn(M, K, generator=g) * (1 + torch.rand(M, 1, generator=g) * 4)   # full f32 mantissas, rows of mixed scale
    W = bf(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g) if c_f32 else None
    z = A.double() @ W.double().t() + bias.double()
    ref = [z, torch.relu(z), torch.nn.functional.gelu(z), z * torch.sigmoid(1.702 * z), torch.nn.functional.silu(z)][act]
    if resid is not None:
        ref = ref + resid.double()
    out = torch.empty(M, N, device="cuda")
    check(lib, lib.anyref_op_gemm(SP, None, P(A.cuda()), P(W.cuda()), P(bias.cuda()), P(out),
                                  P(resid.cuda()) if resid is not None else None, None, M, N, K, act, c_f32))
    rel = close(out, ref, PAIR_REL, f"gemm {M}x{N}x{K}")
    # the same product with the activation rounded to ONE bf16 term is ~2^-9 off: the pair must be far inside that
    one = (bf(A).double() @ W.double().t() + bias.double() - z).abs().max().item() / max(1.0, z.abs().max().item())
    print(f"split-pair gemm {M}x{N}x{K}: rel err {rel:.2e} (single bf16 term: {one:.2e})")
    assert rel < one / 20


def test_split_pair_gemm_row_map(lib):
    M, N, K = 200, 128, 128
    g = torch.Generator().manual_seed(5)
    A, W = torch.randn(M, K, generator=g), bf(torch.randn(N, K, generator=g) * 0.1)
    perm = torch.randperm(M, generator=g).to(torch.int32)
    perm[::7] = -1
    z = A.double() @ W.double().t()
    ref = torch.zeros(M, N, dtype=torch.float64)
    for m in range(M):
        if perm[m] >= 0:
            ref[perm[m]] = z[m]
    out = torch.zeros(M, N, device="cuda")
    check(lib, lib.anyref_op_gemm(SP, None, P(A.cuda()), P(W.cuda()), None, P(out), None, P(perm.cuda()), M, N, K, 0, 0))
    keep = torch.zeros(M, dtype=torch.bool)
    keep[perm[perm >= 0].long()] = True
    close(out[keep.cuda()], ref[keep], PAIR_REL, "row-mapped pair output")


@pytest.mark.parametrize("B,N,K,dual,norm", [(1, 512, 256, 0, 1), (2, 1000, 688, 1, 1), (1, 12288, 4096, 0, 1), (1, 11008, 4096, 1, 1),
                                             (1, 4096, 11008, 0, 0), (2, 4096, 4096, 0, 0), (4, 300, 1024, 0, 0), (1, 32007, 4096, 0, 1)])
def test_split_pair_gemv_vs_float64(lib, B, N, K, dual, norm):
    """bf16 weights exactly as stored x the f32 activation row (never rounded to 16 bits): the decode step of parity16"""
    g = torch.Generator().manual_seed(B + N + K)
    x = torch.randn(B, K, generator=g)
    W, W2 = bf(torch.randn(N, K, generator=g) * 0.05), bf(torch.randn(N, K, generator=g) * 0.05)
    gain = 1 + 0.1 * torch.randn(K, generator=g)
    resid = torch.randn(B, N, generator=g)
    xn = x.double()
    if norm:
        xn = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * gain).double()   # the kernel's f32 statistics
    z = xn @ W.double().t()
    if dual:
        z = torch.nn.functional.silu(z) * (xn @ W2.double().t())
    ref = z + resid.double()
    y = torch.empty(B, N, device="cuda")
    check(lib, lib.anyref_op_gemv(SP, None, P(x.cuda()), P(gain.cuda()) if norm else None, 1e-6, P(W.cuda()),
                                  P(W2.cuda()) if dual else None, None, P(y), P(resid.cuda()), B, N, K, 0))
    close(y, ref, 1e-5, f"gemv {B}x{N}x{K}")


@pytest.mark.parametrize("rms", [0, 1])
@pytest.mark.parametrize("M,D", [(5, 64), (300, 192), (257, 1024), (33, 1280), (9, 4096), (7, 688)])
def test_split_pair_norm_output(lib, rms, M, D):
    """the norm kernels write the pair (rows padded to whole 64-column blocks); read back as hi + lo"""
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g) * 3 + 1
    gain, bias = torch.randn(D, generator=g), torch.randn(D, generator=g)
    if rms:
        ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * gain
    else:
        ref = torch.nn.functional.layer_norm(x, (D,), gain, bias, 1e-6)
    y = torch.empty(M, D, device="cuda")
    check(lib, lib.anyref_op_norm(SP, None, P(x.cuda()), P(gain.cuda()), None if rms else P(bias.cuda()), P(y), M, D, 1e-6, rms))
    close(y, ref, 2e-5, "norm")
    # and the pair itself loses at most 2^-17 of each value
    y0 = torch.empty(M, D, device="cuda")
    check(lib, lib.anyref_op_norm(0, None, P(x.cuda()), P(gain.cuda()), None if rms else P(bias.cuda()), P(y0), M, D, 1e-6, rms))
    assert ((y - y0).abs() <= y0.abs() * 2.0 ** -17 + 1e-30).all()


@pytest.mark.parametrize("B,H,Sq,Sk,hd,causal", [(2, 4, 257, 257, 64, 0), (1, 4, 320, 320, 128, 1), (2, 4, 196, 196, 80, 0)])
def test_split_pair_attention_output(lib, B, H, Sq, Sk, hd, causal):
    """f32 attention whose output rows are written as pairs (the proj / o_proj GEMM's A operand)"""
    from test_gpu_ops import ref_attention
    g = torch.Generator().manual_seed(B + H + Sq + hd)
    q, k, v = (torch.randn(B, s, H, hd, generator=g) for s in (Sq, Sk, Sk))
    scale = 1.0 / math.sqrt(hd)
    ref = ref_attention(q.double(), k.double(), v.double(), scale, causal, None, None, None, 0)
    o = torch.empty(B, Sq, H, hd, device="cuda")
    check(lib, lib.anyref_op_attention(SP, None, P(q.cuda()), P(k.cuda()), P(v.cuda()), P(o), B, H, Sq, Sk, hd, scale, causal,
                                       None, None, None, 0, 0))
    close(o, ref, 2e-5, "attention")


@pytest.mark.parametrize("B,H,size", [(1, 4, 64), (2, 4, 32), (3, 4, 14)])
def test_split_pair_attention_with_rel_pos_bias(lib, B, H, size):
    """SAM attention as parity16 runs it (image_encoder.py:231-260, 354-392): f32 q / k / v multiplied as bf16 pairs (three
    16-bit MFMA passes per product), the decomposed rel-pos bias from the P buffer of the f32 rel-pos GEMM; size 64 = the
    global layers (key tile = one bias row), 14 = the windows (196 tokens, two query blocks), 32 = the general path."""
    from test_gpu_ops import ref_attention
    hd = 80
    g = torch.Generator().manual_seed(B * 11 + H + size)
    S = size * size
    q, k, v = (torch.randn(B, S, H, hd, generator=g) for _ in range(3))
    th, tw = torch.randn(2 * size - 1, hd, generator=g) * 0.3, torch.randn(2 * size - 1, hd, generator=g) * 0.3
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + size - 1
    rq = q.double().permute(0, 2, 1, 3).reshape(B, H, size, size, hd)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, th.double()[idx]).reshape(B, H, S, size)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, tw.double()[idx]).reshape(B, H, S, size)
    scale = hd ** -0.5
    ref = ref_attention(q.double(), k.double(), v.double(), scale, False, None, rel_h, rel_w, size)
    npad = 2 * size
    p = torch.zeros(H, B * S, 2 * npad, dtype=torch.float64)
    qh = q.double().permute(2, 0, 1, 3).reshape(H, B * S, hd)
    p[:, :, : 2 * size - 1] = qh @ th.double().t()
    p[:, :, npad: npad + 2 * size - 1] = qh @ tw.double().t()
    o = torch.empty(B, S, H, hd, device="cuda")
    check(lib, lib.anyref_op_attention_relp(SP, None, P(q.cuda()), P(k.cuda()), P(v.cuda()), P(o), B, H, S, hd, scale,
                                            P(p.float().cuda().contiguous()), 2 * npad, size, size))
    close(o, ref, 2e-5, f"rel-pos attention size {size}")


@pytest.mark.parametrize("B,H,size,hd", [(1, 4, 14, 80), (5, 4, 14, 80), (2, 2, 4, 64), (3, 1, 16, 128)])
def test_split_pair_window_attention_bias_from_tables(lib, B, H, size, hd):
    """SAM windows as parity16 runs them since the P buffer went away: the split-pair kernel computes q . R^T itself from the
    f32 rel-pos tables (both sides as bf16 pairs) and applies the get_rel_pos shift as a scatter (image_encoder.py:321-392);
    14 x 14 = two query blocks of 112, 4 x 4 / 16 x 16 = the tiny configs' windows.  Also: the same call through the P-buffer
    path of the same kernel (only the bias differs in its summation), and the refusal of a non-window shape."""
    from test_gpu_ops import ref_attention
    ld = 128
    g = torch.Generator().manual_seed(B * 13 + H + size)
    S = size * size
    q, k, v = (torch.randn(B, S, H, hd, generator=g) for _ in range(3))
    th, tw = torch.randn(2 * size - 1, hd, generator=g) * 0.3, torch.randn(2 * size - 1, hd, generator=g) * 0.3
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + size - 1
    rq = q.double().permute(0, 2, 1, 3).reshape(B, H, size, size, hd)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, th.double()[idx]).reshape(B, H, S, size)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, tw.double()[idx]).reshape(B, H, S, size)
    scale = hd ** -0.5
    ref = ref_attention(q.double(), k.double(), v.double(), scale, False, None, rel_h, rel_w, size)
    tab = torch.zeros(2, 2 * size, ld)                      # padded rows / columns as the model packs them
    tab[0, : 2 * size - 1, :hd], tab[1, : 2 * size - 1, :hd] = th, tw
    tab = tab.cuda()
    o = torch.empty(B, S, H, hd, device="cuda")
    check(lib, lib.anyref_op_attention_tab(SP, None, P(q.cuda()), P(k.cuda()), P(v.cuda()), P(o), B, H, S, hd, scale,
                                           P(tab[0]), P(tab[1]), ld, size, size))
    close(o, ref, 2e-5, f"window attention from tables, size {size} hd {hd}")
    npad = 2 * size
    p = torch.zeros(H, B * S, 2 * npad, dtype=torch.float64)
    qh = q.double().permute(2, 0, 1, 3).reshape(H, B * S, hd)
    p[:, :, : 2 * size - 1] = qh @ th.double().t()
    p[:, :, npad: npad + 2 * size - 1] = qh @ tw.double().t()
    o2 = torch.empty_like(o)
    check(lib, lib.anyref_op_attention_relp(SP, None, P(q.cuda()), P(k.cuda()), P(v.cuda()), P(o2), B, H, S, hd, scale,
                                            P(p.float().cuda().contiguous()), 2 * npad, size, size))
    close(o, o2.double().cpu(), 2e-5, "tables vs P buffer")
    assert lib.anyref_op_attention_tab(SP, None, P(q.cuda()), P(k.cuda()), P(v.cuda()), P(o), B, H, S - 4, hd, scale,
                                       P(tab[0]), P(tab[1]), ld, size, size) != 0      # Sq != kh * kw


def test_split_pair_attention_ragged_kv_len(lib):
    from test_gpu_ops import ref_attention
    B, H, Sq, Sk, hd = 3, 4, 100, 300, 128
    g = torch.Generator().manual_seed(77)
    q, k, v = (torch.randn(B, s, H, hd, generator=g) for s in (Sq, Sk, Sk))
    kv_len = torch.tensor([300, 131, 64], dtype=torch.int32)
    ref = ref_attention(q.double(), k.double(), v.double(), hd ** -0.5, False, kv_len, None, None, 0)
    o = torch.empty(B, Sq, H, hd, device="cuda")
    check(lib, lib.anyref_op_attention(SP, None, P(q.cuda()), P(k.cuda()), P(v.cuda()), P(o), B, H, Sq, Sk, hd, hd ** -0.5, 0,
                                       P(kv_len.cuda()), None, None, 0, 0))
    close(o, ref, 2e-5, "ragged kv_len")


# ---------------------------------------------------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("window,sam_dim,sam_heads", [(14, 192, 3), (4, 128, 2), (14, 320, 4)])  # last: hd 80, 196-token windows
def test_generate_matches_oracle(window, sam_dim, sam_heads):
    from anyref_amd.model import AnyRefForCausalLM
    cfg = config_tiny(window=window, sam_dim=sam_dim, sam_heads=sam_heads)
    sd = synth_state_dict(cfg, seed=3, scale=0.05)
    clip, sam, ids = make_inputs(cfg, 1, seed=4)
    sizes, H, W = [(224, 180)], [300], [241]
    rig_seg(cfg, sd, clip, sam, ids, sizes, (H, W))
    with torch.no_grad():
        ref = O.anyref_generate(sd, cfg, clip, ids, sam, sizes, H, W, max_new_tokens=6, eos=False)
    assert ref["pred_masks"] is not None
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity16", max_batch=1, max_seg=4)
    m.config.eos_token_id = None
    (out_ids, masks, rest), ex = m.generate(clip, ids[0][None], sam, sizes, H, W, max_new_tokens=6, _return_extras=True)
    assert out_ids[0].cpu().tolist() == ref["output_ids"][0].tolist(), "greedy ids differ"
    n = ref["hidden"][0].shape[0]
    herr = (ex["hidden"][0, :n].cpu() - ref["hidden"][0]).abs().max().item()
    print(f"[parity16] hidden max-abs-err {herr:.3e} (scale {ref['hidden'][0].abs().max().item():.2f})")
    assert herr < 2e-4
    r = compare_generate(m, ref, clip, ids[0], sam, sizes, H, W, 6, sd["lm_head.weight"], cfg.clip.n_patches)
    print("[parity16] " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in r.items()))
    assert r["greedy_ids_identical"] and r["mask_logit_max_abs_err"] <= MASK_TOL, r


def test_ragged_batch_and_teacher_forward():
    from anyref_amd.model import AnyRefForCausalLM
    cfg = config_tiny()
    sd = synth_state_dict(cfg, seed=7, scale=0.05)
    clip, sam, ids = make_inputs(cfg, 2, seed=8)
    sizes, H, W = [(224, 224), (200, 224)], [224, 260], [224, 300]
    rig_seg(cfg, sd, clip, sam, ids, sizes, (H, W))
    with torch.no_grad():
        ref = O.anyref_generate(sd, cfg, clip, ids, sam, sizes, H, W, max_new_tokens=5, eos=False)
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity16", max_batch=2, max_seg=4)
    m.config.eos_token_id = None
    padded, mask = pad(ids)
    out_ids, masks, _ = m.generate(clip, padded, sam, sizes, H, W, max_new_tokens=5, attention_masks=mask)
    for b in range(2):
        want = ref["output_ids"][b]
        assert out_ids[b, : len(want)].cpu().tolist() == want.tolist(), f"row {b}: greedy ids differ"
        if ref["pred_masks"][b] is not None and ref["pred_masks"][b].numel():
            err = (masks[b].cpu() - ref["pred_masks"][b]).abs().max().item()
            print(f"[parity16] batch row {b}: mask max-abs-err {err:.3e}")
            assert err <= MASK_TOL
    # teacher-forced twin (anyref.py:239-466) on the oracle's own ids
    full = ref["output_ids"][0]
    labels = full.clone()
    labels[: len(ids[0])] = -100
    nseg = ref["pred_masks"][0].shape[0]
    gt = [(torch.rand(nseg, H[0], W[0]) > 0.5).float()]
    with torch.no_grad():
        fr = O.anyref_forward(sd, cfg, clip[:1], sam[:1], [full], [labels], sizes[:1], gt, H[:1], W[:1])
    out = m.model_forward_new(clip[:1], sam[:1], full[None], labels[None], None, sizes[:1], gt, H[:1], W[:1], _return_extras=True)
    assert abs(float(out["lm_loss"]) - float(fr["lm_loss"])) < 1e-3
    perr = (out["pred_masks"][0].cpu() - fr["pred_masks"][0]).abs().max().item()
    print(f"[parity16] teacher-forced mask max-abs-err {perr:.3e}")
    assert perr <= MASK_TOL, f"forward mask err {perr}"


@pytest.mark.parametrize("B", [1, 2])
def test_decode_launch_modes_bit_identical(B):
    from anyref_amd.model import AnyRefForCausalLM
    cfg = config_tiny()
    sd = synth_state_dict(cfg, seed=5, scale=0.05)
    clip, sam, ids = make_inputs(cfg, B, seed=6, L=16)
    ids_p, _ = pad(ids)
    sizes, H, W = [(224, 224)] * B, [224] * B, [224] * B
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity16", max_batch=B, max_seg=4)
    m.config.eos_token_id = None
    m.set_graphs(False)
    m.set_early_tail(False)
    out0, _, _ = m.generate(clip, ids_p, sam, sizes, H, W, max_new_tokens=5)
    m.set_seg_token_idx(int(out0[0, ids_p.shape[1] + 2]))
    ref = None
    for overlap in (False, True):
        for graphs in (False, True):
            m.set_overlap(overlap); m.set_graphs(graphs)
            (o_ids, masks, _), ex = m.generate(clip, ids_p, sam, sizes, H, W, max_new_tokens=12, _return_extras=True)
            cur = (o_ids.cpu(), ex["hidden"].cpu(), [None if t is None else t.cpu() for t in masks])
            if ref is None:
                ref = cur
                assert ref[2][0] is not None
                continue
            tag = f"overlap={overlap} graphs={graphs}"
            assert torch.equal(cur[0], ref[0]), f"ids differ ({tag})"
            assert torch.equal(cur[1], ref[1]), f"hidden states differ ({tag})"
            for a, b in zip(cur[2], ref[2]):
                assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), f"masks differ ({tag})"


@pytest.mark.parametrize("B", [1, 4, 6])
def test_generate_llama7b_shaped_layers_vs_oracle(B):
    """Two decoder layers at LLaMA-7B's real widths behind the tiny vision towers (the tile / GEMV variants the headline shapes
    select): every hidden state against the CPU fp32 oracle at the f32 bound.  B = 4: two passes of the two-row GEMV; B = 6:
    the MFMA decode path on pairs."""
    from anyref_amd.model import AnyRefForCausalLM
    cfg = config_tiny()
    cfg = dataclasses.replace(cfg, llm=LlmConfig(vocab=1000, dim=4096, heads=32, layers=2, mlp=11008, max_seq=512))
    sd = synth_state_dict(cfg, seed=21, scale=0.02)
    clip, sam, ids = make_inputs(cfg, B, seed=22, L=65)
    sizes, H, W = [(224, 224)] * B, [224] * B, [224] * B
    rig_seg(cfg, sd, clip, sam, ids, sizes, (H, W))
    n_ref = min(B, 2)
    with torch.no_grad():
        ref = O.anyref_generate(sd, cfg, clip[:n_ref], ids[:n_ref], sam[:n_ref], sizes[:n_ref], H[:n_ref], W[:n_ref],
                                max_new_tokens=6, eos=False)
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity16", max_batch=B, max_seg=4)
    m.config.eos_token_id = None
    padded, mask = pad(ids)
    (out_ids, masks, _), ex = m.generate(clip, padded, sam, sizes, H, W, max_new_tokens=6, attention_masks=mask,
                                         _return_extras=True)
    for b in range(n_ref):
        want_ids = ref["output_ids"][b]
        assert out_ids[b, : len(want_ids)].cpu().tolist() == want_ids.tolist(), f"row {b}: greedy ids differ"
        n = ref["hidden"][b].shape[0]
        got, want = ex["hidden"][b, :n].cpu(), ref["hidden"][b]
        scale = want.abs().max().item()
        err = (got - want).abs().max().item()
        print(f"[parity16 B={B}] row {b}: hidden max-abs-err {err:.3e} (scale {scale:.2f})")
        assert err < 2e-4 * max(1.0, scale)
