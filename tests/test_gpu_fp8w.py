"""BASELINE config 5's arithmetic (fp8 weight-only LLM path) at test size: the quantiser is bit-exact against
its torch statement, the fp8 GEMV matches a torch reference on the dequantised weights, and the whole
generate() agrees with the oracle run on those same dequantised weights."""
import ctypes as C
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from anyref_amd import _lib  # noqa: E402
from anyref_amd.config import config_tiny  # noqa: E402
from anyref_amd.quant import dequantize_rows_fp8, dequantized_state_dict, is_fp8_weight, quantize_rows_fp8  # noqa: E402
from anyref_amd.synth import synth_state_dict  # noqa: E402
from oracle import anyref_oracle as O  # noqa: E402
from test_gpu_e2e import make_inputs, rig_seg  # noqa: E402

pytestmark = pytest.mark.gpu
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


@pytest.mark.parametrize("N,K", [(64, 256), (33, 688), (7, 16), (128, 4096)])
def test_quantiser_bit_exact(N, K):
    lib = _lib.load()
    g = torch.Generator().manual_seed(N * K)
    w = torch.randn(N, K, generator=g) * 0.02
    w[0, :] = 0                      # all-zero row -> scale 1
    w[1, 0] = 3.0                    # a row dominated by one outlier: most values land in the subnormals
    w[2, :8] = torch.tensor([1e-9, -1e-9, 5e-5, -5e-5, 0.02, -0.02, 1e-3, 7e-4])
    wd = w.cuda()
    q = torch.empty(N, K, dtype=torch.uint8, device="cuda")
    s = torch.empty(N, dtype=torch.float32, device="cuda")
    assert lib.anyref_op_quant_fp8(None, P(wd), N, K, P(q), P(s)) == 0, lib.anyref_op_last_error()
    torch.cuda.synchronize()
    q_ref, s_ref = quantize_rows_fp8(w)
    assert torch.equal(s.cpu(), s_ref)
    # -0 and +0 are the same value; compare the decoded numbers and the bytes away from zero
    assert torch.equal(dequantize_rows_fp8(q.cpu(), s.cpu()), dequantize_rows_fp8(q_ref, s_ref))
    nz = (q_ref & 0x7F) != 0
    assert torch.equal(q.cpu()[nz], q_ref[nz])


@pytest.mark.parametrize("B,N,K,dual,norm", [(1, 512, 256, False, True), (2, 96, 688, False, False),
                                              (1, 688, 256, True, True), (4, 40, 4096, False, True),
                                              # 5 .. 8 rows: one pass on the 4 x 4 x 4 MFMA form (gemv_rows8_kernel, fp8 pairs widened
                                              # to packed bf16); K = 13824 / 11008: down_proj, two K halves inside the launch
                                              (8, 1000, 5120, False, True), (8, 688, 5120, True, True), (5, 100, 256, False, False),
                                              (8, 5120, 13824, False, False), (6, 130, 11008, False, False)])
def test_gemv_fp8_vs_torch(B, N, K, dual, norm):
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * N + K)
    w = torch.randn(N, K, generator=g) * 0.03
    w2 = torch.randn(N, K, generator=g) * 0.03
    x = torch.randn(B, K, generator=g)
    gain = torch.rand(K, generator=g) + 0.5
    q, s = quantize_rows_fp8(w)
    q2, s2 = quantize_rows_fp8(w2)
    xn = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * gain if norm else x
    xb = xn.bfloat16().float()                                     # activations are staged as bf16
    ref = xb @ dequantize_rows_fp8(q, s).T
    if dual:
        ref = torch.nn.functional.silu(ref) * (xb @ dequantize_rows_fp8(q2, s2).T)
    y = torch.empty(B, N, device="cuda")
    keep = [x.cuda(), gain.cuda(), q.cuda(), q2.cuda(), s.cuda(), s2.cuda()]
    rc = lib.anyref_op_gemv_fp8(None, P(keep[0]), P(keep[1]) if norm else None, 1e-6, P(keep[2]),
                                P(keep[3]) if dual else None, P(keep[4]), P(keep[5]) if dual else None, P(y), None, B, N, K)
    assert rc == 0, lib.anyref_op_last_error()
    torch.cuda.synchronize()
    err = (y.cpu() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K", [(320, 512, 256),        # 128^2 tile, 3 stages
                                    (2048, 2048, 512),     # 256^2 tile
                                    (300, 8200, 128),      # 64 x 256 tile, ragged edges
                                    (320, 1024, 4096),     # split-K slices + reduction with the scale
                                    (1500, 700, 192)])     # many 128^2 tiles, ragged N
def test_gemm_fp8_operand_vs_torch(M, N, K):
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g)).bfloat16()
    w = torch.randn(N, K, generator=g) * 0.03
    bias = torch.randn(N, generator=g) * 0.1
    resid = torch.randn(M, N, generator=g)
    q, s = quantize_rows_fp8(w)
    ref = A.float() @ dequantize_rows_fp8(q, s).T + bias + resid
    keep = [A.cuda(), q.cuda(), s.cuda(), bias.cuda(), resid.cuda()]
    C32 = torch.empty(M, N, device="cuda")
    rc = lib.anyref_op_gemm_fp8(None, P(keep[0]), P(keep[1]), P(keep[2]), P(keep[3]), P(C32), P(keep[4]), M, N, K, 0, 1)
    assert rc == 0, lib.anyref_op_last_error()
    torch.cuda.synchronize()
    err = (C32.cpu() - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err


# ---- BASELINE configs[4] widths (13B: H = 5120, 40 heads of 128, MLP 13824): the shapes `bench.py --config c5` runs ----
@pytest.mark.parametrize("B", [1, 4])
@pytest.mark.parametrize("N,K,dual", [(15360, 5120, False),     # fused q/k/v projection (24 x-values per thread)
                                      (5120, 13824, False),    # down_proj (K = 13824: the 32-value staging)
                                      (13824, 5120, True),     # gate/up as 27648 interleaved rows -> SwiGLU pairs
                                      (5120, 5120, False)])    # o_proj
def test_gemv_fp8_13b_widths_vs_torch(B, N, K, dual):
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + N + K)
    w = torch.randn(N, K, generator=g) * 0.02
    w2 = torch.randn(N, K, generator=g) * 0.02
    x = torch.randn(B, K, generator=g)
    gain = torch.rand(K, generator=g) + 0.5
    q, s = quantize_rows_fp8(w)
    q2, s2 = quantize_rows_fp8(w2)
    xn = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * gain
    xb = xn.bfloat16().double()
    ref = xb @ dequantize_rows_fp8(q, s).double().T
    if dual:
        ref = torch.nn.functional.silu(ref) * (xb @ dequantize_rows_fp8(q2, s2).double().T)
    y = torch.empty(B, N, device="cuda")
    keep = [x.cuda(), gain.cuda(), q.cuda(), q2.cuda(), s.cuda(), s2.cuda()]
    rc = lib.anyref_op_gemv_fp8(None, P(keep[0]), P(keep[1]), 1e-6, P(keep[2]), P(keep[3]) if dual else None, P(keep[4]),
                                P(keep[5]) if dual else None, P(y), None, B, N, K)
    assert rc == 0, lib.anyref_op_last_error()
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item()
    print(f"fp8 GEMV B={B} N={N} K={K} dual={dual}: max-abs-err {err:.3e} (scale {ref.abs().max().item():.2f})")
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M", [320, 2560])                      # one prompt; batch 8 x S = 320 (configs[4])
@pytest.mark.parametrize("N,K", [(15360, 5120), (5120, 13824), (5120, 5120)])
def test_gemm_fp8_operand_13b_widths_vs_torch(M, N, K):
    """fp8-operand MFMA GEMM at the 13B shapes: whole-M / 64 x 256 / 256^2 tiles, split-K with the row scale applied
    in the reduction (M = 320, K >= 2048), the scale in the epilogue otherwise.  Reference: fp32 matmul of the same
    bf16 activations with the DEQUANTISED weights (torch on the GPU; fp64 spot-check of a row block on the host)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16()
    w = torch.randn(N, K, generator=g) * 0.02
    resid = torch.randn(M, N, generator=g)
    q, s = quantize_rows_fp8(w)
    wd = dequantize_rows_fp8(q, s)
    keep = [A.cuda(), q.cuda(), s.cuda(), resid.cuda()]
    ref = keep[0].float() @ wd.cuda().T + keep[3]
    C32 = torch.empty(M, N, device="cuda")
    rc = lib.anyref_op_gemm_fp8(None, P(keep[0]), P(keep[1]), P(keep[2]), None, P(C32), P(keep[3]), M, N, K, 0, 1)
    assert rc == 0, lib.anyref_op_last_error()
    torch.cuda.synchronize()
    scale = max(1.0, ref.abs().max().item())
    err = (C32 - ref).abs().max().item()
    ref64 = A[:16].double() @ wd.double().T + resid[:16].double()
    err64 = (C32[:16].cpu().double() - ref64).abs().max().item()
    print(f"fp8 GEMM M={M} N={N} K={K}: max-abs-err {err:.3e} vs torch fp32, {err64:.3e} vs fp64 rows (scale {scale:.2f})")
    assert err < 2e-3 * scale and err64 < 2e-3 * scale, (err, err64)


def test_generate_13b_shaped_layers_fp8w_batch8_vs_oracle():
    """BASELINE configs[4] at reduced depth: two decoder layers at LLaMA-13B's real widths (5120 / 40 heads of 128 / MLP
    13824; vocab 1000) behind the tiny vision towers, `perf_fp8w`, batch 8 -- the fp8-operand prefill GEMMs at M = 8 x S
    rows, the MFMA decode path (B > 4: fp8 weights through the GEMM, split-K + scale in the reduction) and the padded
    fp8 row stride -- every hidden state held against the CPU fp32 oracle on the DEQUANTISED weights.  Clone of
    `test_generate_llama7b_shaped_layers_vs_oracle` (anyref.py:663-679 / :704-716 call sites)."""
    import dataclasses
    from anyref_amd.config import LlmConfig
    from anyref_amd.model import AnyRefForCausalLM
    from test_gpu_e2e import pad, PERF_HIDDEN_REL_7B
    B = 8
    cfg = config_tiny()
    cfg = dataclasses.replace(cfg, llm=LlmConfig(vocab=1000, dim=5120, heads=40, layers=2, mlp=13824, max_seq=512))
    sd = synth_state_dict(cfg, seed=31, scale=0.02)
    sd_dq = dequantized_state_dict(sd)
    clip, sam, ids = make_inputs(cfg, B, seed=32, L=65)
    sizes, H, W = [(224, 224)] * B, [224] * B, [224] * B
    rig_seg(cfg, sd_dq, clip, sam, ids, sizes, (H, W))
    n_ref = 2
    with torch.no_grad():
        ref = O.anyref_generate(sd_dq, cfg, clip[:n_ref], ids[:n_ref], sam[:n_ref], sizes[:n_ref], H[:n_ref], W[:n_ref],
                                max_new_tokens=6, eos=False)
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="perf_fp8w", max_batch=B, max_seg=4)
    m.config.eos_token_id = None
    padded, mask = pad(ids)
    (out_ids, masks, _), ex = m.generate(clip, padded, sam, sizes, H, W, max_new_tokens=6, attention_masks=mask,
                                         _return_extras=True)
    # B = 1 on the same handle: the fp8 decode GEMVs over the padded rows (K = 5120: 24 values / thread, K = 13824: 32)
    (out1, _, _), ex1 = m.generate(clip[:1], ids[0][None], sam[:1], sizes[:1], H[:1], W[:1], max_new_tokens=6,
                                   _return_extras=True)
    for b in range(n_ref):
        want_ids = ref["output_ids"][b]
        n = ref["hidden"][b].shape[0]
        Sp = len(ids[b]) + 255
        want = ref["hidden"][b]
        scale = want.abs().max().item()
        runs = [("B=8", out_ids[b], ex["hidden"][b])] + ([("B=1", out1[0], ex1["hidden"][0])] if b == 0 else [])
        for tag, oi, hid in runs:
            same = oi[: len(want_ids)].cpu().tolist() == want_ids.tolist()
            got = hid[:n].cpu()
            perr = (got[:Sp] - want[:Sp]).abs().max().item()
            print(f"[perf_fp8w 13B-shaped {tag}] row {b}: prefill hidden max-abs-err {perr:.3e} (scale {scale:.2f}), ids identical: {same}")
            assert perr < PERF_HIDDEN_REL_7B * max(1.0, scale), f"prefill hidden err {perr} (scale {scale})"
            if same:
                derr = (got[Sp:] - want[Sp:]).abs().max().item()
                print(f"[perf_fp8w 13B-shaped {tag}] row {b}: decode hidden max-abs-err {derr:.3e}")
                assert derr < PERF_HIDDEN_REL_7B * max(1.0, scale), f"decode hidden err {derr} (scale {scale})"


def test_generate_fp8w_matches_oracle_on_dequantised_weights():
    from anyref_amd.model import AnyRefForCausalLM
    cfg = config_tiny()
    sd = synth_state_dict(cfg, seed=3, scale=0.05)
    sd_dq = dequantized_state_dict(sd)
    assert any(is_fp8_weight(k) for k in sd) and not is_fp8_weight("model.embed_tokens.weight")
    clip, sam, ids = make_inputs(cfg, 1, seed=4)
    sizes, H, W = [(224, 180)], [300], [241]
    rig_seg(cfg, sd_dq, clip, sam, ids, sizes, (H, W))
    with torch.no_grad():
        ref = O.anyref_generate(sd_dq, cfg, clip, ids, sam, sizes, H, W, max_new_tokens=6, eos=False)
    assert ref["pred_masks"] is not None
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="perf_fp8w", max_batch=1, max_seg=4)
    m.config.eos_token_id = None
    (out_ids, masks, _), ex = m.generate(clip, ids[0][None], sam, sizes, H, W, max_new_tokens=6, _return_extras=True)
    n = ref["hidden"][0].shape[0]
    hscale = ref["hidden"][0].abs().max().item()
    if out_ids[0].cpu().tolist() == ref["output_ids"][0].tolist():
        herr = (ex["hidden"][0, :n].cpu() - ref["hidden"][0]).abs().max().item()
        print(f"[perf_fp8w] hidden max-abs-err {herr:.3e} (scale {hscale:.2f})")
        assert herr < 0.012 * hscale, f"hidden err {herr}"         # 2 x measured (2.4e-2 on 4.57), = test_gpu_e2e.PERF_HIDDEN_REL
    from oracle.check import compare_generate                      # masks always compared (teacher-forced after a flip)
    r = compare_generate(m, ref, clip, ids[0], sam, sizes, H, W, 6, sd_dq["lm_head.weight"], cfg.clip.n_patches)
    print("[perf_fp8w] " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in r.items()))
    assert r["mask_logit_max_abs_err"] <= 0.009 * r["logit_range"], r    # 2 x measured (3.5e-3 relative)
    # the fp8 model is NOT the bf16 model: same call on the original weights differs
    m2 = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="perf", max_batch=1, max_seg=4)
    m2.config.eos_token_id = None
    (_, _, _), ex2 = m2.generate(clip, ids[0][None], sam, sizes, H, W, max_new_tokens=6, _return_extras=True)
    assert (ex2["hidden"][0, :n] - ex["hidden"][0, :n]).abs().max().item() > 1e-3
    assert m.device_bytes < m2.device_bytes


def test_c5_full_depth_batch8_mfma_decode_vs_batch1_gemv():
    """BASELINE configs[4] at FULL depth (13B: 40 layers x 5120 / 13824, fp8 weights, SAM-H, 1024^2, batch 8).  A 13B fp32
    oracle does not fit the host budget of `-m gpu`, so this is a GPU-side consistency test of the two decode paths the
    shape selects: the eight rows through ONE batched call (prefill GEMMs at M = 8 x 320 rows, 8 SAM-H encodes, the 8-row
    decode GEMV: fp8 pairs widened to packed bf16 for the 4 x 4 x 4 MFMA, every weight read once per step, down_proj as two
    K halves) against the same rows one at a time (B = 1: the fp8 packed-FMA decode GEMVs).  Both round activations to bf16 at
    the same points and multiply the same fp8 bytes, so they differ by summation order only: hidden states within the bf16
    bound, greedy ids identical on the peaked (fan-in) workload.
    (Reduced-depth parity of each path against the CPU oracle: test_generate_13b_shaped_layers_fp8w_batch8_vs_oracle.)"""
    import gc
    from anyref_amd.config import config_13b, IMAGE_TOKEN_INDEX
    from anyref_amd.model import AnyRefForCausalLM
    B, T = 8, 6
    cfg = config_13b()
    cfg.llm.max_seq = 512
    sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16, init="fan_in")
    g = torch.Generator().manual_seed(41)
    clip = torch.randn(B, 3, 224, 224, generator=g)
    sam = torch.randn(2, 3, 1024, 1024, generator=g).repeat(4, 1, 1, 1)          # two distinct images, four prompts each
    ids = torch.stack([torch.cat([torch.tensor([1, IMAGE_TOKEN_INDEX]), torch.randint(3, 32000, (63,), generator=g)]) for _ in range(B)])
    sizes, H, W = [(1024, 1024)] * B, [1024] * B, [1024] * B
    m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="perf_fp8w", max_batch=B, max_seg=4)
    del sd
    gc.collect()
    torch.cuda.empty_cache()
    m.config.eos_token_id = None
    (o8, _, _), ex8 = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T, _return_extras=True)
    m.set_seg_token_idx(int(o8[0, ids.shape[1] + 2]))                              # row 0 emits a [SEG]: masks from both paths
    (o8, masks8, _), ex8 = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T, _return_extras=True)
    Sp = ids.shape[1] + 255
    n = Sp + T - 1
    same_rows, worst_p, worst_d, worst_m = 0, 0.0, 0.0, 0.0
    for b in range(B):
        (o1, masks1, _), ex1 = m.generate(clip[b:b + 1], ids[b:b + 1], sam[b:b + 1], sizes[:1], H[:1], W[:1], max_new_tokens=T,
                                          _return_extras=True)
        same = o1[0].cpu().tolist() == o8[b, : o1.shape[1]].cpu().tolist()
        same_rows += int(same)
        h8, h1 = ex8["hidden"][b, :n], ex1["hidden"][0, :n]
        scale = h1.abs().max().item()
        worst_p = max(worst_p, (h8[:Sp] - h1[:Sp]).abs().max().item() / scale)
        if same:
            worst_d = max(worst_d, (h8[Sp:] - h1[Sp:]).abs().max().item() / scale)
            if masks1 is not None and masks8 is not None and masks8[b].numel() and masks1[0].shape == masks8[b].shape:
                rng = masks1[0].abs().max().item()
                worst_m = max(worst_m, (masks8[b] - masks1[0]).abs().max().item() / max(rng, 1e-30))
    print(f"[C5 full depth] B=8 (MFMA decode) vs B=1 (GEMV): ids identical on {same_rows}/{B} rows; hidden rel diff prefill "
          f"{worst_p:.3e}, decode {worst_d:.3e}; mask-logit rel diff {worst_m:.3e}; {m.device_bytes / 2**30:.1f} GiB")
    assert same_rows >= B - 1, same_rows          # (one near-tie may fall the other way between two summation orders)
    # 2 - 3 x the differences measured on MI355X (hidden 6.1e-3 / 5.7e-3 of the scale, mask logits 1.1e-3 of their range)
    assert worst_p < 0.015 and worst_d < 0.015, (worst_p, worst_d)
    assert worst_m < 0.004, worst_m
