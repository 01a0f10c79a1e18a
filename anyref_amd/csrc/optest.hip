// Kernel-level C entry points for the parity tests (include/anyref_hip_ops.h).
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/anyref_hip_ops.h"
#include "kernels.h"

using namespace anyref;

static thread_local std::string g_op_err;

#define OP_GUARD(body)                                   \
  try {                                                  \
    body;                                                \
    hipError_t _e = hipGetLastError();                   \
    if (_e != hipSuccess) {                              \
      g_op_err = std::string("HIP error: ") + hipGetErrorString(_e); \
      return 3;                                          \
    }                                                    \
    return 0;                                            \
  } catch (const std::exception& e) {                    \
    g_op_err = e.what();                                 \
    return 2;                                            \
  }

namespace {
// t = 3 (split pairs, ANYREF_MODE_PARITY16): the entry points keep their f32 interface -- an f32 operand that the mode
// carries as a bf16 pair is split into a temporary here, a pair-typed result is read back as hi + lo
struct TmpBuf {
  void* p = nullptr;
  explicit TmpBuf(size_t bytes) { HIP_TRY(hipMalloc(&p, bytes ? bytes : 16)); HIP_TRY(hipMemset(p, 0, bytes ? bytes : 16)); }
  ~TmpBuf() { (void)hipDeviceSynchronize(); (void)hipFree(p); }
};
inline int pad64(int k) { return (k + 63) / 64 * 64; }
}  // namespace

extern "C" {

const char* anyref_op_last_error(void) { return g_op_err.c_str(); }

int anyref_op_gemm(int t, void* stream, const void* A, const void* W, const float* bias, void* C,
                   const float* resid, const int32_t* row_map, int M, int N, int K, int act, int c_f32) {
  OP_GUARD({
    GemmArgs a;
    a.A = A; a.lda = K; a.W = W; a.ldw = K; a.bias = bias; a.C = C; a.ldc = N; a.resid = resid; a.ldr = N;
    a.row_map = row_map; a.M = M; a.N = N; a.K = K; a.act = act; a.c_f32 = c_f32;
    if (const char* e = getenv("ANYREF_OPTEST_LDW_PAD")) a.ldw = K + atoi(e);  // probe: padded weight rows
    if (t == 3) {  // A f32 [M,K] -> pairs; W bf16 [N,K] (K % 64 == 0); C f32 [*, N] either way (read back from pairs if !c_f32)
      hipStream_t st = (hipStream_t)stream;
      if (K % 64) throw std::runtime_error("op_gemm t=3: K % 64 != 0");
      TmpBuf As((size_t)M * K * 4);
      launch_convert<sp16>(reinterpret_cast<const float*>(A), K, As.p, K, M, K, st);
      a.A = As.p;
      if (c_f32) {
        launch_gemm<sp16>(a, st);
      } else {
        int rows = M;
        if (row_map) {
          std::vector<int> h(M);
          HIP_TRY(hipMemcpy(h.data(), row_map, M * 4, hipMemcpyDeviceToHost));
          for (int v : h) rows = std::max(rows, v + 1);
        }
        TmpBuf Cs((size_t)rows * pad64(N) * 4);
        a.C = Cs.p; a.ldc = pad64(N);
        launch_gemm<sp16>(a, st);
        launch_unsplit(Cs.p, pad64(N), reinterpret_cast<float*>(C), N, rows, N, st);
      }
    }
    else if (t == 0) launch_gemm<float>(a, (hipStream_t)stream);
    else if (t == 2) launch_gemm<f16>(a, (hipStream_t)stream);
    else launch_gemm<bf16>(a, (hipStream_t)stream);
  });
}

int anyref_op_gemv(int t, void* stream, const float* x, const float* gain, float eps, const void* W,
                   const void* W2, const float* bias, float* y, const float* resid, int B, int N, int K, int act) {
  OP_GUARD({
    GemvArgs a;
    a.x = x; a.ldx = K; a.gain = gain; a.eps = eps; a.W = W; a.W2 = W2; a.bias = bias; a.y = y; a.resid = resid;
    a.ldy = N; a.B = B; a.N = N; a.K = K; a.act = act;
    if (const char* e = getenv("ANYREF_OPTEST_LDW_PAD")) a.ldw = K + atoi(e);  // probe: padded weight rows
    if (t == 3) launch_gemv<sp16>(a, (hipStream_t)stream);  // W bf16, x f32 staged as f32
    else if (t == 0) launch_gemv<float>(a, (hipStream_t)stream); else launch_gemv<bf16>(a, (hipStream_t)stream);
  });
}

int anyref_op_norm(int t, void* stream, const float* x, const float* gain, const float* bias, float* y, int M,
                   int D, float eps, int rms) {
  OP_GUARD({
    NormArgs a;
    a.x = x; a.ldx = D; a.gain = gain; a.bias = bias; a.y = y; a.ldy = D; a.M = M; a.D = D; a.eps = eps;
    a.rms = rms; a.y_f32 = 1;
    if (t == 3) {  // the norm writes pairs; y gets hi + lo
      TmpBuf Ys((size_t)M * pad64(D) * 4);
      a.y = Ys.p; a.ldy = pad64(D); a.y_f32 = 0;
      launch_norm<sp16>(a, (hipStream_t)stream);
      launch_unsplit(Ys.p, pad64(D), y, D, M, D, (hipStream_t)stream);
    } else
    if (t == 0) launch_norm<float>(a, (hipStream_t)stream);
    else if (t == 2) launch_norm<f16>(a, (hipStream_t)stream);
    else launch_norm<bf16>(a, (hipStream_t)stream);
  });
}

int anyref_op_attention(int t, void* stream, const void* q, const void* k, const void* v, void* o, int B, int H,
                        int Sq, int Sk, int hd, float scale, int causal, const int32_t* kv_len,
                        const float* rel_h, const float* rel_w, int kh, int kw) {
  OP_GUARD({
    AttnArgs a;
    a.Q = q; a.K = k; a.V = v; a.O = o;
    a.q_bs = (int64_t)Sq * H * hd; a.q_rs = H * hd; a.q_hs = hd;
    a.k_bs = a.v_bs = (int64_t)Sk * H * hd; a.k_rs = a.v_rs = H * hd; a.k_hs = a.v_hs = hd;
    a.o_bs = (int64_t)Sq * H * hd; a.o_rs = H * hd; a.o_hs = hd;
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.hd = hd; a.scale = scale; a.causal = causal; a.kv_len = kv_len;
    a.rel_h = rel_h; a.rel_w = rel_w; a.kh = kh; a.kw = kw;
    if (t == 3) {  // f32 operands, pair-typed output rows [B*Sq, H*hd] (H*hd % 64 == 0); o gets hi + lo
      if ((H * hd) % 64) throw std::runtime_error("op_attention t=3: H * hd % 64 != 0");
      TmpBuf Os((size_t)B * Sq * H * hd * 4);
      a.O = Os.p; a.o_split = 1; a.sp16 = 1;
      launch_attention<float>(a, (hipStream_t)stream);
      launch_unsplit(Os.p, H * hd, reinterpret_cast<float*>(o), H * hd, B * Sq, H * hd, (hipStream_t)stream);
    } else
    if (t == 0) launch_attention<float>(a, (hipStream_t)stream);
    else if (t == 2) launch_attention<f16>(a, (hipStream_t)stream);
    else launch_attention<bf16>(a, (hipStream_t)stream);
  });
}

int anyref_op_attention_tab(int t, void* stream, const void* q, const void* k, const void* v, void* o, int B, int H, int S,
                            int hd, float scale, const void* tab_h, const void* tab_w, int tab_ld, int kh, int kw) {
  OP_GUARD({
    AttnArgs a;
    a.Q = q; a.K = k; a.V = v; a.O = o;
    a.q_bs = a.k_bs = a.v_bs = a.o_bs = (int64_t)S * H * hd;
    a.q_rs = a.k_rs = a.v_rs = a.o_rs = H * hd; a.q_hs = a.k_hs = a.v_hs = a.o_hs = hd;
    a.B = B; a.H = H; a.Sq = S; a.Sk = S; a.hd = hd; a.scale = scale;
    a.rel_tab_h = tab_h; a.rel_tab_w = tab_w; a.rel_tab_ld = tab_ld; a.kh = kh; a.kw = kw;
    if (t == 3) {  // split-pair attention: f32 operands and f32 tables, pair-typed output rows; o gets hi + lo
      if ((H * hd) % 64) throw std::runtime_error("op_attention_tab t=3: H * hd % 64 != 0");
      if (!attention_takes_rel_tables(4, hd, S, S, kh, kw, true)) throw std::runtime_error("op_attention_tab t=3: not a window shape");
      TmpBuf Os((size_t)B * S * H * hd * 4);
      a.O = Os.p; a.o_split = 1; a.sp16 = 1;
      launch_attention<float>(a, (hipStream_t)stream);
      launch_unsplit(Os.p, H * hd, reinterpret_cast<float*>(o), H * hd, B * S, H * hd, (hipStream_t)stream);
    } else
    if (t == 2) launch_attention<f16>(a, (hipStream_t)stream);
    else launch_attention<bf16>(a, (hipStream_t)stream);
  });
}

int anyref_op_gemm_gather(int t, void* stream, const void* A, const int32_t* a_row_map, const void* W, const float* bias,
                          void* C, const float* resid, int M, int N, int K, int c_f32) {
  OP_GUARD({
    GemmArgs a;
    a.A = A; a.lda = K; a.W = W; a.ldw = K; a.bias = bias; a.C = C; a.ldc = N; a.resid = resid; a.ldr = N;
    a.a_row_map = a_row_map; a.M = M; a.N = N; a.K = K; a.c_f32 = c_f32;
    if (t == 2) launch_gemm<f16>(a, (hipStream_t)stream);
    else if (t == 1) launch_gemm<bf16>(a, (hipStream_t)stream);
    else throw std::runtime_error("gemm_gather: 16-bit types only");
  });
}

int anyref_op_attention_relp(int t, void* stream, const void* q, const void* k, const void* v, void* o, int B, int H, int S,
                             int hd, float scale, const float* rel_p, int rel_ld, int kh, int kw) {
  OP_GUARD({
    AttnArgs a;
    a.Q = q; a.K = k; a.V = v; a.O = o;
    a.q_bs = a.k_bs = a.v_bs = a.o_bs = (int64_t)S * H * hd;
    a.q_rs = a.k_rs = a.v_rs = a.o_rs = H * hd; a.q_hs = a.k_hs = a.v_hs = a.o_hs = hd;
    a.B = B; a.H = H; a.Sq = S; a.Sk = S; a.hd = hd; a.scale = scale;
    a.rel_p = rel_p; a.rel_ld = rel_ld; a.rel_hs = (int64_t)B * S * rel_ld; a.kh = kh; a.kw = kw;
    if (t == 3) {  // split-pair attention: f32 operands, pair-typed output rows; o gets hi + lo
      if ((H * hd) % 64) throw std::runtime_error("op_attention_relp t=3: H * hd % 64 != 0");
      TmpBuf Os((size_t)B * S * H * hd * 4);
      a.O = Os.p; a.o_split = 1; a.sp16 = 1;
      launch_attention<float>(a, (hipStream_t)stream);
      launch_unsplit(Os.p, H * hd, reinterpret_cast<float*>(o), H * hd, B * S, H * hd, (hipStream_t)stream);
    } else
    if (t == 0) launch_attention<float>(a, (hipStream_t)stream);
    else if (t == 2) launch_attention<f16>(a, (hipStream_t)stream);
    else launch_attention<bf16>(a, (hipStream_t)stream);
  });
}

int anyref_op_rel_pos(int t, void* stream, const void* q, const float* tab_h, const float* tab_w, int B, int H,
                      int size, int hd, float* rel_h, float* rel_w) {
  OP_GUARD({
    const int64_t S = (int64_t)size * size;
    if (t == 0)
      launch_rel_pos<float>(q, S * H * hd, H * hd, hd, tab_h, tab_w, B, H, size, hd, rel_h, rel_w, (hipStream_t)stream);
    else
      launch_rel_pos<bf16>(q, S * H * hd, H * hd, hd, tab_h, tab_w, B, H, size, hd, rel_h, rel_w, (hipStream_t)stream);
  });
}

int anyref_op_postprocess(void* stream, const float* low, int n, int lh, int lw, int S, int rh, int rw, int H, int W,
                          float* out) {
  OP_GUARD(launch_postprocess(low, (int64_t)lh * lw, n, lh, lw, S, rh, rw, H, W, out, (hipStream_t)stream));
}

int anyref_op_gemm_fp8(void* stream, const void* A, const uint8_t* W8, const float* scale, const float* bias, void* C,
                       const float* resid, int M, int N, int K, int act, int c_f32) {
  OP_GUARD({
    GemmArgs a;
    a.A = A; a.lda = K; a.W = W8; a.ldw = K; a.w_fp8 = 1; a.col_scale = scale; a.bias = bias; a.C = C; a.ldc = N;
    a.resid = resid; a.ldr = N; a.M = M; a.N = N; a.K = K; a.act = act; a.c_f32 = c_f32;
    launch_gemm<bf16>(a, (hipStream_t)stream);
  });
}

int anyref_op_quant_fp8(void* stream, const float* src, int N, int K, uint8_t* q, float* scale) {
  OP_GUARD(launch_quant_fp8_rows(src, K, N, K, q, K, scale, (hipStream_t)stream));
}

int anyref_op_gemv_fp8(void* stream, const float* x, const float* gain, float eps, const uint8_t* W,
                       const uint8_t* W2, const float* scale, const float* scale2, float* y, const float* resid,
                       int B, int N, int K) {
  OP_GUARD({
    GemvArgs a;
    a.x = x; a.ldx = K; a.gain = gain; a.eps = eps; a.W = W; a.W2 = W2; a.wscale = scale; a.wscale2 = scale2;
    a.w_fp8 = 1; a.y = y; a.resid = resid; a.ldy = N; a.B = B; a.N = N; a.K = K;
    launch_gemv<bf16>(a, (hipStream_t)stream);
  });
}

int anyref_op_iou_counts(void* stream, const float* logits, const uint8_t* target, int n, int64_t hw,
                         int64_t* counts) {
  OP_GUARD(launch_iou_counts(logits, target, n, hw, counts, (hipStream_t)stream));
}

int anyref_op_avs_counts(void* stream, const float* logits, const uint8_t* target, int n, int64_t hw,
                         const float* cuts, int nth, float cut_pred, int64_t* conf, int64_t* hist) {
  OP_GUARD(launch_avs_counts(logits, target, n, hw, cuts, nth, cut_pred, conf, hist, (hipStream_t)stream));
}

int anyref_op_sam_preprocess(void* stream, const uint8_t* img, int h, int w, int S, const float* mean3,
                             const float* std3, float* out) {
  OP_GUARD(launch_sam_preprocess(img, h, w, S, mean3, std3, out, (hipStream_t)stream));
}

int anyref_op_pil_resample_u8(void* stream, const uint8_t* in, int H, int W, int C, uint8_t* tmp, uint8_t* out, int ow,
                              int oh, const int32_t* xbounds, const int32_t* xk, int kx, const int32_t* ybounds,
                              const int32_t* yk, int ky) {
  OP_GUARD(launch_pil_resample_u8(in, H, W, C, tmp, out, ow, oh, xbounds, xk, kx, ybounds, yk, ky, (hipStream_t)stream));
}

int anyref_op_pool_ref_tokens(void* stream, const float* feats, int n, int L, int H, int n_out, float* out) {
  OP_GUARD(launch_pool_ref_tokens(feats, n, L, H, n_out, out, (hipStream_t)stream));
}

int anyref_op_kaldi_fbank(void* stream, const float* wave, int C, int T, int win, int shift, int padded, float preemph,
                          const float* banks, int n_mel, const double* tw, double* scratch, int target_len, float mean,
                          float stdv, float* out) {
  OP_GUARD(launch_kaldi_fbank(wave, C, T, win, shift, padded, preemph, banks, n_mel, tw, scratch, target_len, mean, stdv, out,
                              (hipStream_t)stream));
}
int anyref_op_clip_finish(void* stream, const uint8_t* img, int ih, int iw, int y0, int x0, int h, int w, int S,
                          const float* mean3, const float* std3, float* out) {
  OP_GUARD(launch_clip_finish(img, ih, iw, y0, x0, h, w, S, mean3, std3, out, (hipStream_t)stream));
}
}
