"""Kernel-level parity: every hand-written HIP kernel against a plain torch fp32 reference of
the same op, through the C-ABI test entry points (include/anyref_hip_ops.h).

t=0 is the parity arithmetic (f32 MFMA, must agree to ~1e-5 relative), t=1 the perf arithmetic
(bf16 MFMA operands, fp32 accumulate; tolerance = bf16 rounding of the operands).
"""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {0: 2e-5, 1: 2e-2, 2: 3e-3}


@pytest.fixture(scope="module")
def lib():
    from anyref_amd import _lib
    return _lib.load()


_DT = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}     # ty: storage type id of include/anyref_hip_ops.h


def dev(t, ty):
    return t.cuda().to(_DT[ty]).contiguous()


def rnd(t, ty):
    """what the kernel sees after storage rounding"""
    return t.to(_DT[ty]).float()


_KEEP = []


def P(t):
    """device pointer of t; t is kept alive until check() has synchronised (a temporary's block
    would otherwise go back to the caching allocator before the kernel runs)."""
    if t is None:
        return None
    _KEEP.append(t)
    return C.c_void_p(t.data_ptr())


def check(lib, rc):
    assert rc == 0, lib.anyref_op_last_error().decode()
    torch.cuda.synchronize()
    _KEEP.clear()


def close(got, ref, tol):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert math.isfinite(err) and err <= tol * scale, f"max abs err {err:.3e} vs scale {scale:.3e} (tol {tol})"


@pytest.mark.parametrize("ty", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 192, 192), (320, 384, 1032), (6, 256, 256), (1000, 64, 72),
                                   (70, 130, 24), (4900, 200, 1280), (320, 640, 4096), (64, 128, 128)])
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_gemm(lib, ty, M, N, K, act):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + act)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    z = rnd(A, ty) @ rnd(W, ty).t() + bias
    ref = [z, torch.relu(z), torch.nn.functional.gelu(z), z * torch.sigmoid(1.702 * z), torch.nn.functional.silu(z)][act]
    ref = ref + resid
    Ad, Wd, bd, rd = dev(A, ty), dev(W, ty), bias.cuda(), resid.cuda()
    out = torch.empty(M, N, device="cuda")
    check(lib, lib.anyref_op_gemm(ty, None, P(Ad), P(Wd), P(bd), P(out), P(rd), None, M, N, K, act, 1))
    close(out, ref, TOL[ty] * 4)


@pytest.mark.parametrize("M,N,K", [(4096, 3840, 128), (4096, 5120, 128), (4096, 1280, 192), (4096, 1280, 64),
                                   (4096, 1280, 448), (320, 12288, 128), (320, 22016, 192), (4900, 1280, 128),
                                   # 13B (config 5): prefill qkv / gate-up widths, split-K o / down at full K,
                                   # and the M = 8 rows of the MFMA decode path
                                   (320, 15360, 128), (320, 27648, 192), (320, 5120, 5120), (320, 5120, 13824),
                                   (8, 15360, 5120), (8, 5120, 13824), (2560, 15360, 128)])
def test_gemm_tile_paths_of_the_big_shapes(lib, M, N, K):
    """the tile heuristics of launch_gemm at the SAM-H / LLaMA-7B output shapes (short K): 256^2, 256x320,
    128x160 (ragged LDS-DMA round, 1 .. 7 K tiles through the 3-stage ring), 64x256 with 3 / 2 stages, 128^2;
    the SAM-H shapes (M >= 4096) also in f16, the storage type of that tower in the perf build"""
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g)
    for ty in ((1, 2) if M >= 4096 else (1,)):
        ref = rnd(A, ty) @ rnd(W, ty).t() + bias
        out = torch.empty(M, N, device="cuda")
        check(lib, lib.anyref_op_gemm(ty, None, P(dev(A, ty)), P(dev(W, ty)), P(bias.cuda()), P(out), None, None, M, N, K, 0, 1))
        close(out, ref, 1e-4 if K < 2048 else 4e-4)   # 16-bit products are exact in f32; only the summation order differs


@pytest.mark.parametrize("ty", [0, 1, 2])
def test_gemm_row_map_and_typed_out(lib, ty):
    M, N, K = 200, 96, 64
    g = torch.Generator().manual_seed(5)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    perm = torch.randperm(M, generator=g).to(torch.int32)
    perm[::7] = -1                                  # dropped rows (window padding)
    ref = torch.zeros(M, N)
    z = rnd(A, ty) @ rnd(W, ty).t()
    for m in range(M):
        if perm[m] >= 0:
            ref[perm[m]] = z[m]
    out = torch.zeros(M, N, device="cuda", dtype=_DT[ty])
    check(lib, lib.anyref_op_gemm(ty, None, P(dev(A, ty)), P(dev(W, ty)), None, P(out), None, P(perm.cuda()), M, N,
                                  K, 0, 0))
    close(out, ref, {0: 1e-4, 1: 1e-2, 2: 2e-3}[ty])


@pytest.mark.parametrize("ty", [1, 2])
@pytest.mark.parametrize("M,Msrc,N,K,c_f32", [(4096, 4900, 1280, 1280, 1), (300, 517, 200, 128, 0), (1000, 1000, 384, 256, 1)])
def test_gemm_a_row_gather(lib, ty, M, Msrc, N, K, c_f32):
    """C[m] = A[map[m]] W^T + bias + resid[m]: the SAM window layers' proj over the real tokens of the window layout
    (image_encoder.py:196-229); first shape = SAM-H's (4096 tokens out of 4900 window rows, f32 output + residual)."""
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(Msrc, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    amap = torch.randperm(Msrc, generator=g)[:M].to(torch.int32)
    ref = rnd(A, ty)[amap.long()] @ rnd(W, ty).t() + bias + (resid if c_f32 else 0)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if c_f32 else _DT[ty])
    check(lib, lib.anyref_op_gemm_gather(ty, None, P(dev(A, ty)), P(amap.cuda()), P(dev(W, ty)), P(bias.cuda()), P(out),
                                         P(resid.cuda()) if c_f32 else None, M, N, K, c_f32))
    close(out, ref, {1: 1e-2, 2: 2e-3}[ty])


@pytest.mark.parametrize("ty", [0, 1])
@pytest.mark.parametrize("B,N,K,dual,norm", [(1, 512, 256, 0, 1), (2, 1000, 688, 1, 1), (4, 300, 1024, 0, 0),
                                             (3, 64, 4096, 1, 0), (1, 33, 11008, 0, 1),
                                             # the 13B widths config 5 selects (K = 5120 -> 24 x-values per thread,
                                             # K = 13824 -> 32, neither a multiple of 512 * 8) and the ragged lm_head
                                             (1, 640, 5120, 0, 1), (2, 130, 5120, 1, 1), (1, 64, 13824, 0, 0),
                                             (4, 40, 13824, 0, 1), (2, 96, 13824, 1, 0), (1, 32007, 4096, 0, 1),
                                             # 5 .. 8 rows (bf16: ONE pass on the 4 x 4 x 4 MFMA form, gemv_rows8_kernel; K = 11008 / 13824 as two K halves)
                                             (8, 12288, 4096, 0, 1), (8, 1000, 4096, 1, 1), (5, 100, 256, 0, 0), (7, 4096, 11008, 0, 0),
                                             (8, 5120, 13824, 0, 0), (6, 33, 5120, 1, 1)])
def test_gemv(lib, ty, B, N, K, dual, norm):
    g = torch.Generator().manual_seed(B + N + K)
    x = torch.randn(B, K, generator=g)
    W, W2 = torch.randn(N, K, generator=g) * 0.05, torch.randn(N, K, generator=g) * 0.05
    gain = 1 + 0.1 * torch.randn(K, generator=g)
    resid = torch.randn(B, N, generator=g)
    xn = x
    if norm:
        xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * gain
    xn = rnd(xn, ty)
    z = xn @ rnd(W, ty).t()
    if dual:
        z = torch.nn.functional.silu(z) * (xn @ rnd(W2, ty).t())
    ref = z + resid
    y = torch.empty(B, N, device="cuda")
    check(lib, lib.anyref_op_gemv(ty, None, P(x.cuda()), P(gain.cuda()) if norm else None, 1e-6, P(dev(W, ty)),
                                  P(dev(W2, ty)) if dual else None, None, P(y), P(resid.cuda()), B, N, K, 0))
    close(y, ref, TOL[ty] * 4)


@pytest.mark.parametrize("rms", [0, 1])
@pytest.mark.parametrize("M,D", [(5, 64), (300, 192), (257, 1024), (33, 1280), (9, 4096), (3, 5120)])
def test_norm(lib, rms, M, D):
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g) * 3 + 1
    gain, bias = torch.randn(D, generator=g), torch.randn(D, generator=g)
    if rms:
        ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * gain
    else:
        ref = torch.nn.functional.layer_norm(x, (D,), gain, bias, 1e-6)
    y = torch.empty(M, D, device="cuda")
    check(lib, lib.anyref_op_norm(0, None, P(x.cuda()), P(gain.cuda()), None if rms else P(bias.cuda()), P(y), M, D,
                                  1e-6, rms))
    close(y, ref, 2e-5)


def ref_attention(q, k, v, scale, causal, kv_len, rel_h, rel_w, kw):
    B, Sq, H, hd = q.shape
    Sk = k.shape[1]
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    if rel_h is not None:
        kh = rel_h.shape[-1]
        s = (s.view(B, H, Sq, kh, kw) + rel_h[..., :, None] + rel_w[..., None, :]).view(B, H, Sq, Sk)
    mask = torch.zeros(B, 1, Sq, Sk, dtype=torch.bool)
    if kv_len is not None:
        for b in range(B):
            mask[b, :, :, kv_len[b]:] = True
    if causal:
        mask |= torch.ones(Sq, Sk, dtype=torch.bool).triu(1)
    s = s.masked_fill(mask, float("-inf"))
    return torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v)


@pytest.mark.parametrize("ty", [0, 1, 2])
@pytest.mark.parametrize("B,H,Sq,Sk,hd,causal", [
    (2, 3, 257, 257, 64, 0),      # CLIP
    (1, 4, 320, 320, 128, 1),     # LLaMA prefill
    (2, 2, 1, 200, 128, 0),       # decode step
    (3, 8, 6, 196, 16, 0),        # mask decoder token -> image
    (3, 8, 196, 6, 16, 0),        # image -> token
    (3, 8, 6, 6, 32, 0),          # token self attention
    (2, 2, 130, 130, 80, 1),      # hd 80, causal, ragged tail tiles
    (2, 2, 100, 130, 80, 0),      # hd 80, Sq != Sk
    (1, 8, 7, 4096, 16, 0),       # SAM-H mask decoder token -> image: keys split over workgroups + merge
    (1, 4, 70, 2500, 32, 0),      # two query blocks, ragged key splits
    (1, 5, 196, 196, 80, 0),      # SAM window without bias: keys resident in LDS (13 waves, 5 tiles of 48)
    (1, 3, 205, 205, 80, 0),      # same form, last tile and last query block ragged
    (1, 3, 230, 230, 80, 0),      # 193..240 tokens beyond the resident form: streaming 7 waves x 80 keys
    (2, 2, 200, 200, 80, 0),      # kv_len given: falls back to the streaming form
])
def test_attention(lib, ty, B, H, Sq, Sk, hd, causal):
    g = torch.Generator().manual_seed(Sq * 3 + Sk + hd)
    q, k, v = (torch.randn(B, S, H, hd, generator=g) for S in (Sq, Sk, Sk))
    kv_len = None
    if B > 1 and Sk > 64 and not causal:
        kv_len = torch.tensor([Sk - 7 * b for b in range(B)], dtype=torch.int32)
    scale = hd ** -0.5
    ref = ref_attention(rnd(q, ty), rnd(k, ty), rnd(v, ty), scale, causal, kv_len, None, None, 0)
    o = torch.empty(B, Sq, H, hd, device="cuda", dtype=_DT[ty])
    check(lib, lib.anyref_op_attention(ty, None, P(dev(q, ty)), P(dev(k, ty)), P(dev(v, ty)), P(o), B, H, Sq, Sk, hd,
                                       scale, causal, P(kv_len.cuda()) if kv_len is not None else None, None, None,
                                       0, 0))
    close(o, ref, {0: 3e-5, 1: 3e-2, 2: 4e-3}[ty])


@pytest.mark.parametrize("ty", [0, 1, 2])
@pytest.mark.parametrize("B,H,size,hd", [(3, 2, 14, 80), (1, 2, 16, 64), (2, 3, 4, 64), (1, 2, 64, 80), (1, 1, 32, 80)])
def test_sam_attention_rel_pos(lib, ty, B, H, size, hd):
    """windowed / global SAM attention incl. the decomposed rel-pos bias (image_encoder.py:231-392)."""
    g = torch.Generator().manual_seed(size + hd)
    S = size * size
    q, k, v = (torch.randn(B, S, H, hd, generator=g) for _ in range(3))
    th, tw = torch.randn(2 * size - 1, hd, generator=g) * 0.3, torch.randn(2 * size - 1, hd, generator=g) * 0.3
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + size - 1
    qr = rnd(q, ty)
    rq = qr.permute(0, 2, 1, 3).reshape(B, H, size, size, hd)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, th[idx]).reshape(B, H, S, size)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, tw[idx]).reshape(B, H, S, size)
    rh = torch.empty(B, H, S, size, device="cuda")
    rw = torch.empty(B, H, S, size, device="cuda")
    if ty == 2:      # (the stand-alone rel-pos kernel is a bf16 / f32 test aid: take the bias from the reference)
        rh, rw = rel_h.cuda().contiguous(), rel_w.cuda().contiguous()
    else:
        check(lib, lib.anyref_op_rel_pos(ty, None, P(dev(q, ty)), P(th.cuda()), P(tw.cuda()), B, H, size, hd, P(rh), P(rw)))
        close(rh, rel_h, 1e-4)
        close(rw, rel_w, 1e-4)
    scale = hd ** -0.5
    ref = ref_attention(qr, rnd(k, ty), rnd(v, ty), scale, False, None, rel_h, rel_w, size)
    o = torch.empty(B, S, H, hd, device="cuda", dtype=_DT[ty])
    check(lib, lib.anyref_op_attention(ty, None, P(dev(q, ty)), P(dev(k, ty)), P(dev(v, ty)), P(o), B, H, S, S, hd,
                                       scale, 0, None, P(rh), P(rw), size, size))
    close(o, ref, {0: 5e-5, 1: 3e-2, 2: 4e-3}[ty])


@pytest.mark.parametrize("ty", [1, 2])
@pytest.mark.parametrize("B,H,size", [(1, 2, 64), (2, 3, 64), (1, 2, 32)])
def test_sam_global_attention_bias_from_p(lib, B, H, size, ty):
    """SAM global attention as the encoder runs it (image_encoder.py:231-260, 354-392): the bias comes from the P buffer
    of the batched rel-pos GEMM (q . rel_pos_h / rel_pos_w for every table row), the kernel applies the get_rel_pos shift.
    size 64 = SAM-H's 4096 tokens: the two-query-blocks-per-wave kernel; size 32: the general one."""
    hd = 80
    g = torch.Generator().manual_seed(B * 11 + H + size)
    S = size * size
    q, k, v = (torch.randn(B, S, H, hd, generator=g) for _ in range(3))
    th, tw = torch.randn(2 * size - 1, hd, generator=g) * 0.3, torch.randn(2 * size - 1, hd, generator=g) * 0.3
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + size - 1
    qr = rnd(q, ty)
    rq = qr.permute(0, 2, 1, 3).reshape(B, H, size, size, hd)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, th[idx]).reshape(B, H, S, size)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, tw[idx]).reshape(B, H, S, size)
    scale = hd ** -0.5
    ref = ref_attention(qr, rnd(k, ty), rnd(v, ty), scale, False, None, rel_h, rel_w, size)
    npad = 2 * size
    p = torch.zeros(H, B * S, 2 * npad)
    qh = qr.permute(2, 0, 1, 3).reshape(H, B * S, hd)
    p[:, :, : 2 * size - 1] = qh @ th.t()
    p[:, :, npad: npad + 2 * size - 1] = qh @ tw.t()
    o = torch.empty(B, S, H, hd, device="cuda", dtype=_DT[ty])
    check(lib, lib.anyref_op_attention_relp(ty, None, P(dev(q, ty)), P(dev(k, ty)), P(dev(v, ty)), P(o), B, H, S, hd, scale,
                                            P(p.cuda().contiguous()), 2 * npad, size, size))
    close(o, ref, 3e-2 if ty == 1 else 4e-3)


@pytest.mark.parametrize("ty", [1, 2])
@pytest.mark.parametrize("B,H", [(1, 1), (5, 3)])
def test_sam_window_attention_bias_from_tables(lib, B, H, ty):
    """SAM-H windows (14 x 14, hd 80, bf16 / f16): the kernel computes q . R^T itself from the rel-pos tables and applies
    the get_rel_pos shift as a scatter (image_encoder.py:321-392), keys held in the row-padded order (one bias row per
    16-key block); refused for shapes outside that form."""
    size, hd, ld = 14, 80, 128
    g = torch.Generator().manual_seed(B * 7 + H)
    S = size * size
    q, k, v = (torch.randn(B, S, H, hd, generator=g) for _ in range(3))
    th, tw = (rnd(torch.randn(2 * size - 1, hd, generator=g) * 0.3, ty) for _ in range(2))
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + size - 1
    qr = rnd(q, ty)
    rq = qr.permute(0, 2, 1, 3).reshape(B, H, size, size, hd)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, th[idx]).reshape(B, H, S, size)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, tw[idx]).reshape(B, H, S, size)
    scale = hd ** -0.5
    ref = ref_attention(qr, rnd(k, ty), rnd(v, ty), scale, False, None, rel_h, rel_w, size)
    tab = torch.zeros(2, 2 * size, ld)                      # padded rows / columns as the model packs them
    tab[0, : 2 * size - 1, :hd], tab[1, : 2 * size - 1, :hd] = th, tw
    tab = tab.to(_DT[ty]).cuda()
    o = torch.empty(B, S, H, hd, device="cuda", dtype=_DT[ty])
    check(lib, lib.anyref_op_attention_tab(ty, None, P(dev(q, ty)), P(dev(k, ty)), P(dev(v, ty)), P(o), B, H, S, hd, scale,
                                           P(tab[0]), P(tab[1]), ld, size, size))
    close(o, ref, 3e-2 if ty == 1 else 4e-3)
    # the same call against the precomputed-bias path of the same kernel: only the f32 summation order differs
    o2 = torch.empty_like(o)
    check(lib, lib.anyref_op_attention(ty, None, P(dev(q, ty)), P(dev(k, ty)), P(dev(v, ty)), P(o2), B, H, S, S, hd, scale, 0,
                                       None, P(rel_h.cuda().contiguous()), P(rel_w.cuda().contiguous()), size, size))
    close(o, o2.float(), 1e-2 if ty == 1 else 2e-3)
    assert lib.anyref_op_attention_tab(ty, None, P(dev(q, ty)), P(dev(k, ty)), P(dev(v, ty)), P(o), B, H, 100, hd, scale,
                                       P(tab[0]), P(tab[1]), ld, 10, 10) != 0      # not a resident-form shape


@pytest.mark.parametrize("n,lh,S,rs,os_", [(2, 56, 224, (224, 224), (224, 224)), (3, 56, 224, (150, 224), (301, 437)),
                                           (1, 256, 1024, (683, 1024), (427, 640))])
def test_postprocess(lib, n, lh, S, rs, os_):
    g = torch.Generator().manual_seed(n + lh)
    low = torch.randn(n, 1, lh, lh, generator=g)
    m = torch.nn.functional.interpolate(low, (S, S), mode="bilinear", align_corners=False)[..., : rs[0], : rs[1]]
    ref = torch.nn.functional.interpolate(m, os_, mode="bilinear", align_corners=False)[:, 0]
    out = torch.empty(n, os_[0], os_[1], device="cuda")
    check(lib, lib.anyref_op_postprocess(None, P(low.cuda().contiguous()), n, lh, lh, S, rs[0], rs[1], os_[0], os_[1],
                                         P(out)))
    close(out, ref, 2e-5)
