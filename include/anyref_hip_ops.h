/*
 * anyref_hip_ops.h — kernel-level entry points of libanyref_hip.so used by the parity tests
 * (tests/test_gpu_ops.py) to check every hand-written kernel against the oracle / torch fp32 in
 * isolation.  Not part of the drop-in boundary (that is anyref_hip.h).
 * `t` selects the storage type: 0 = f32 (MFMA 16x16x4 f32), 1 = bf16 (MFMA 16x16x32 bf16), 3 = split pairs (ANYREF_MODE_PARITY16: f32
 * operands at the interface, carried as two bf16 terms inside; weights bf16; gemm / gemv / norm / attention entries), 2 = f16 (MFMA 16x16x32
 * f16: the SAM image encoder of the perf build; GEMM, norm and attention entries).
 * All pointers are device pointers; `stream` is a hipStream_t.
 */
#ifndef ANYREF_HIP_OPS_H
#define ANYREF_HIP_OPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* C[row_map[m]] = act(A W^T + bias) + resid; A,W in type t; C f32 if c_f32 else t */
int anyref_op_gemm(int t, void* stream, const void* A, const void* W, const float* bias, void* C,
                   const float* resid, const int32_t* row_map, int M, int N, int K, int act, int c_f32);
/* y = act(rmsnorm?(x) W^T) [* (x W2^T)] + resid;  x,y f32; W in type t */
int anyref_op_gemv(int t, void* stream, const float* x, const float* gain, float eps, const void* W,
                   const void* W2, const float* bias, float* y, const float* resid, int B, int N, int K, int act);
/* LayerNorm (rms=0) / RMSNorm (rms=1); x f32, y f32 */
int anyref_op_norm(int t, void* stream, const float* x, const float* gain, const float* bias, float* y, int M,
                   int D, float eps, int rms);
/* q,k,v,o [B,S,H,hd] contiguous in type t; rel_h/rel_w f32 [B,H,Sq,kh|kw] or NULL */
int anyref_op_attention(int t, void* stream, const void* q, const void* k, const void* v, void* o, int B, int H,
                        int Sq, int Sk, int hd, float scale, int causal, const int32_t* kv_len,
                        const float* rel_h, const float* rel_w, int kh, int kw);
/* SAM decomposed rel-pos tables from q [B,S=size*size,H,hd] (type t) */
int anyref_op_rel_pos(int t, void* stream, const void* q, const float* tab_h, const float* tab_w, int B, int H,
                      int size, int hd, float* rel_h, float* rel_w);
/* Row-wise fp8 e4m3 quantisation used by ANYREF_MODE_PERF_FP8W: src f32 [N,K] -> q u8 [N,K], scale f32 [N] */
int anyref_op_quant_fp8(void* stream, const float* src, int N, int K, uint8_t* q, float* scale);
/* bf16 GEMM with an fp8 weight operand: C = act((A W8^T) * scale[n] + bias) (+ resid); A bf16 [M,K], W8 u8 [N,K] */
int anyref_op_gemm_fp8(void* stream, const void* A, const uint8_t* W8, const float* scale, const float* bias, void* C,
                       const float* resid, int M, int N, int K, int act, int c_f32);
/* decode GEMV on fp8 weights: y[b,n] = (sum_k bf16(norm(x))[b,k] * q[n,k]) * scale[n] (SwiGLU pair if W2) */
int anyref_op_gemv_fp8(void* stream, const float* x, const float* gain, float eps, const uint8_t* W,
                       const uint8_t* W2, const float* scale, const float* scale2, float* y, const float* resid,
                       int B, int N, int K);
/* SURVEY.md §8 f-2, replaces `(torch.sigmoid(pred) > 0.5).int()` + utils/utils.py:79-91
 * intersectionAndUnionGPU(pred, gt, 2, ignore_index=255) (call site eval_referseg.py:189-208):
 * logits f32 [n, hw] (device), target u8 [n, hw] with values 0 / 1 / 255 (device), counts i64 [n, 6]
 * (device) = {I0, I1, O0, O1, T0, T1}; area_intersection = I, area_union = O + T - I, area_target = T. */
int anyref_op_iou_counts(void* stream, const float* logits, const uint8_t* target, int n, int64_t hw,
                         int64_t* counts);
/* SURVEY.md §8 f-2 (AVS), the per-pixel part of `mask_iou` and `Eval_Fmeasure` / `_eval_pr`
 * (utils/pyutils.py:163-236; caller eval_avs_object.py:168-178) in one pass: logits f32 [n, hw], target u8
 * [n, hw] (0 / non-zero), cuts f32 [nth] ascending (device; cut_i = smallest logit whose sigmoid reaches the
 * i-th P/R threshold), cut_pred = smallest logit with sigmoid > 0.5.  conf i64 [n, 4] = counts by 2 * pred + gt;
 * hist i64 [n, nth + 1, 2] = pixels by (number of thresholds passed, gt). nth <= 255. */
int anyref_op_avs_counts(void* stream, const float* logits, const uint8_t* target, int n, int64_t hw,
                         const float* cuts, int nth, float cut_pred, int64_t* conf, int64_t* hist);
/* SURVEY.md §8 f-1, replaces `sam_preprocess` (utils/refer_seg.py:560-570) after ResizeLongestSide: img u8
 * HWC [h, w, 3] (device) -> out f32 CHW [3, S, S] (device), (x - mean3[c]) / std3[c], zero padded; mean3 /
 * std3 are HOST pointers. */
int anyref_op_sam_preprocess(void* stream, const uint8_t* img, int h, int w, int S, const float* mean3,
                             const float* std3, float* out);
/* SURVEY.md §8 f-1, replaces `ResizeLongestSide.apply_image` (segment_anything/utils/transforms.py:27-34) and the
 * bicubic resize inside `CLIPImageProcessor.preprocess` (utils/refer_seg.py:578-580): Pillow's 8-bit separable
 * fixed-point resampling, bit-exact.  in u8 [H, W, C] dev -> out u8 [oh, ow, C] dev; tmp u8 [H, ow, C] dev scratch
 * (needed when both sizes change).  Coefficient tables (device, int32; built on the host as Pillow's
 * precompute_coeffs + normalize_coeffs_8bpc do): xbounds [ow, 2] = (first input column, tap count), xk [ow, kx]
 * = round(tap * 2^22); ybounds / yk likewise for rows.  A pass whose size does not change is skipped (tables may be
 * NULL), as in Pillow. */
int anyref_op_pil_resample_u8(void* stream, const uint8_t* in, int H, int W, int C, uint8_t* tmp, uint8_t* out,
                              int ow, int oh, const int32_t* xbounds, const int32_t* xk, int kx,
                              const int32_t* ybounds, const int32_t* yk, int ky);
/* SURVEY.md §8 f-3, the token pooling of the `ref_images` branch (model/anyref.py:335-338, :697-700): feats f32
 * [n, L, H] dev (`encode_images` output, L = 256) -> mean over groups of 16 tokens -> (L/16 != n_out) mean over
 * groups of n_out rows -> out f32 [n, n_out, H] dev (n_out = IMG_REF_NUM = 4). */
int anyref_op_pool_ref_tokens(void* stream, const float* feats, int n, int L, int H, int n_out, float* out);
/* SURVEY.md §8 f-1, the rest of the CLIP input path (utils/refer_seg.py:578-587): window [y0, y0+h) x [x0, x0+w) of
 * img u8 [ih, iw, 3] dev -> x * (1/255) (in double, then f32) -> (x - mean3[c]) / std3[c] -> bilinear
 * (align_corners = False) to out f32 CHW [3, S, S] dev.  mean3 / std3 are HOST pointers. */
int anyref_op_clip_finish(void* stream, const uint8_t* img, int ih, int iw, int y0, int x0, int h, int w, int S,
                          const float* mean3, const float* std3, float* out);
/* SURVEY.md §8 f-4, the audio front-end in front of the ImageBind trunk: replaces `waveform2melspec` + `Normalize`
 * (model/ImageBind/data.py:28-64,152-153; torchaudio.compliance.kaldi.fbank with htk_compat, hanning window, 25 ms / 10 ms
 * frames, no dither, 0.97 pre-emphasis, DC removal, power spectrum, log) for ONE clip: wave f32 [C, T] dev (channel 0
 * is analysed, the clip mean is taken over all channels), banks f32 [n_mel, padded / 2 + 1] dev (kaldi mel filters + zero
 * Nyquist column), tw f64 [2 * padded] dev (cos, then sin, of 2 pi i / padded), scratch f64 [1] dev -> out f32
 * [n_mel, target_len] dev = (pad_or_cut(log-mel^T) - mean) / std.  win / padded: 400 / 512 (16 kHz). */
int anyref_op_kaldi_fbank(void* stream, const float* wave, int C, int T, int win, int shift, int padded, float preemph,
                          const float* banks, int n_mel, const double* tw, double* scratch, int target_len, float mean,
                          float stdv, float* out);
/* 16-bit (t = 1 bf16 / 2 f16) window attention with the decomposed rel-pos bias computed inside the kernel from the tables
 * (image_encoder.py:321-392 get_rel_pos / add_decomposed_rel_pos): tab_h [2*kh-1, hd], tab_w [2*kw-1, hd] in type t,
 * rows at stride tab_ld elements; S = kh*kw tokens, [B,S,H,hd] operands.  Only the shapes the resident-key form
 * takes (hd 80, 192 < S <= 208); others are refused. */
int anyref_op_attention_tab(int t, void* stream, const void* q, const void* k, const void* v, void* o, int B, int H, int S,
                            int hd, float scale, const void* tab_h, const void* tab_w, int tab_ld, int kh, int kw);
/* 16-bit GEMM with an A-row gather: C[m, :] = A[a_row_map[m], :] W^T + bias (+ resid[m, :]) -- the SAM window layers'
 * proj over the real tokens of the window-layout attention output (model.hip sam_encoder; image_encoder.py:196-229,
 * window_unpartition).  t = 1 (bf16) / 2 (f16); K % 64 == 0. */
int anyref_op_gemm_gather(int t, void* stream, const void* A, const int32_t* a_row_map, const void* W, const float* bias,
                          void* C, const float* resid, int M, int N, int K, int c_f32);
/* the same attention with the bias taken from the P buffer the model's batched rel-pos GEMM writes (model.hip
 * sam_encoder, image_encoder.py:354-392): P f32 [H][B*S][rel_ld], columns [0, rel_ld/2) = q . rel_pos_h[e],
 * [rel_ld/2, rel_ld) = q . rel_pos_w[e]; the kernel applies the get_rel_pos shift.  S = kh*kw tokens, [B,S,H,hd]
 * operands.  kw == 64 (SAM global attention) takes the two-query-blocks-per-wave kernel. */
int anyref_op_attention_relp(int t, void* stream, const void* q, const void* k, const void* v, void* o, int B, int H, int S,
                             int hd, float scale, const float* rel_p, int rel_ld, int kh, int kw);
/* Sam.postprocess_masks on low [n,lh,lw] f32 */
int anyref_op_postprocess(void* stream, const float* low, int n, int lh, int lw, int S, int rh, int rw, int H,
                          int W, float* out);
const char* anyref_op_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
